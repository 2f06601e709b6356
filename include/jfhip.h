/*
 * libjfhip -- MI355X (gfx950) implementation of the jellyfish PLONK proving hot path:
 * KZG commit / batch_commit / open (Pippenger MSM) and the radix-2 (coset) NTT / iNTT.
 *
 * This is the drop-in boundary: a flat C ABI (plain pointers and sizes) that an FFI crate
 * binds in place of the reference's ICICLE/CUDA seam.  Each entry point cites the reference
 * interface it replaces (paths relative to the jellyfish repository root).
 *
 * Data conventions (identical to the reference's GPU seam, pcs/src/univariate_kzg/mod.rs:
 * 967-998 and the zero-copy reinterpret at :815-819):
 *   - field elements are little-endian u64 limbs in MONTGOMERY form, exactly arkworks'
 *     in-memory `Fp.0.0` (Fr: 4 limbs; BN254 Fq: 4 limbs; BLS12-381 Fq: 6 limbs);
 *   - a packed affine G1 point is x limbs followed by y limbs (64 B BN254, 96 B BLS12-381);
 *     the identity is all-zero limbs (+ the infinity flag where one is carried);
 *   - every output is fully reduced (< p); NTT data is natural order in and out.
 *
 * All functions return JFHIP_OK (0) or a negative status; jfhip_last_error() gives the
 * message for the calling thread.  No function aborts or throws across the boundary.
 * Every entry point is thread-safe (the reference enters commit() concurrently from rayon
 * workers: pcs/src/univariate_kzg/mod.rs:134-136, plonk/src/proof_system/snark.rs:525-534).
 */
#ifndef JFHIP_H
#define JFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JFHIP_OK 0
#define JFHIP_ERR_INVALID (-1)  /* maps to PCSError::InvalidParameters (pcs/src/errors.rs:19-34) */
#define JFHIP_ERR_DEVICE (-2)   /* HIP runtime failure; maps to PCSError::UpstreamError (errors.rs:33-34) */
#define JFHIP_ERR_NO_DEVICE (-3)
#define JFHIP_ERR_OOM (-4)

/* curve_id */
#define JFHIP_BN254 0
#define JFHIP_BLS12_381 1

typedef struct jfhip_srs jfhip_srs;        /* device-resident prover parameters (powers_of_g) */
typedef struct jfhip_stream jfhip_stream;  /* a HIP stream + its MSM/NTT workspace */

/* ---- runtime ------------------------------------------------------------------------------ */

/* Select the HIP device this process uses (one process per GPU).  device < 0: keep the
 * current device.  Replaces nothing in the reference (ICICLE picks device 0 implicitly). */
int32_t jfhip_init(int32_t device);

/* Give back what the library holds on behalf of the calling thread and the process: the calling thread's default stream
 * with its workspace, and the process-wide device-table caches (jfhip_cache_clear).  Handles the caller created
 * (streams, SRS) stay valid and are the caller's to destroy.  The library can be used again afterwards.  No counterpart
 * in the reference (ICICLE's RAII slices free themselves). */
int32_t jfhip_shutdown(void);

/* Free the process-wide caches of device tables (NTT twiddle tables per (domain size, direction, coset offset); the
 * quotient kernel's 1/(n(x_i - 1)) table per (n, m, offset), 32 B x m each) that no call in flight is using, and the
 * FK23 transforms every live SRS handle keeps (jfhip_srs_release_caches).  The caches
 * are bounded anyway (least recently used first, JFHIP_CACHE_BYTES per cache, default 8 GiB); a long-running prover that
 * is done with a circuit size can give the memory back at once. */
int32_t jfhip_cache_clear(void);

/* Copy the calling thread's last error message (NUL-terminated, truncated to len). */
int32_t jfhip_last_error(char *buf, size_t len);

/* Number of visible HIP devices (0 if none); never fails. */
int32_t jfhip_device_count(void);

/* Replaces icicle_deps::warmup_new_stream()  (pcs/src/lib.rs:430-435): create a stream and
 * warm its allocator.  stream == NULL is accepted by every call below and means a
 * per-thread default stream owned by the library. */
int32_t jfhip_stream_create(jfhip_stream **out);
int32_t jfhip_stream_destroy(jfhip_stream *s);
/* Synchronise `s` (NULL: the calling thread's default stream) and free its grow-only MSM / NTT workspace (about 4.9 GiB
 * after a 2^24-point MSM); the stream stays usable and grows a workspace again on demand.  The default stream of a
 * thread other than the one that loaded the library is released with its workspace when that thread exits. */
int32_t jfhip_stream_release_workspace(jfhip_stream *s);
/* Replaces CudaStream::synchronize()  (pcs/src/univariate_kzg/mod.rs:942). */
int32_t jfhip_stream_sync(jfhip_stream *s);
/* The raw hipStream_t of a stream (for callers that enqueue their own HIP work). */
void *jfhip_stream_raw(jfhip_stream *s);
/* Wrap an existing hipStream_t (e.g. torch's current stream); the library does not own it. */
int32_t jfhip_stream_wrap(void *hip_stream, jfhip_stream **out);

/* ---- SRS ---------------------------------------------------------------------------------- */

/* Replaces GPUCommittable::load_prover_param_to_gpu + ark_affine_to_icicle
 * (pcs/src/univariate_kzg/mod.rs:767-800, 975-980).
 * points: n host G1Affine records, `stride_bytes` apart, each starting with x limbs then y
 * limbs (Montgomery).  If stride_bytes > 2*fq_bytes the byte at offset 2*fq_bytes is
 * arkworks' `infinity: bool`.  The library packs the points, uploads them once and
 * precomputes the window multiples 2^(c*w) * P_i it needs (kept in HBM; sized for 288 GB),
 * so later commits need no per-window combine.  window_bits = 0 lets the library choose.
 * An MSM that ends at or below n / 16 (a small circuit against a universal SRS) runs on a second, smaller window table
 * over a prefix of the points, built on the first such call and owned by the handle (< 1/8 of the main table in all;
 * environment JFHIP_SRS_PREFIX=0 disables it).  Results do not depend on which table runs.
 * n == 0 is JFHIP_ERR_INVALID (reference :772-778 rejects supported_degree > len-1). */
int32_t jfhip_srs_load(int32_t curve_id, const void *points, size_t n, size_t stride_bytes, int32_t window_bits,
                       jfhip_srs **out);
/* Lagrange-basis commit key of a radix-2 domain, derived from a loaded (monomial) SRS: a NEW handle of n + blinders points,
 * n = 2^log_n, holding [L_i(beta)]g for the Lagrange basis L_i of the domain {w^i} (the inverse group DFT of the first n
 * points: L_i(X) = (1/n) sum_j w^(-ij) X^j) followed by V_k = [beta^(n+k)]g - [beta^k]g, k < blinders.  jfhip_msm* over
 * (e_0 .. e_(n-1), b_0 .. b_(h)) on that handle is the commitment -- the same group element -- of the polynomial with
 * evaluations e_i on the domain plus the prover's blinding (b_0 + .. + b_h X^h)(X^n - 1) (reference
 * plonk/src/proof_system/prover.rs:457-466; wire polynomials :77-91, Plookup sorted polynomials :99-124): witness values
 * are often a few bits wide where the coefficients after the iFFT (relation/src/constraint_system.rs:1422-1444) never
 * are, and the bucket sort drops zero digits.  No counterpart in the reference (it commits coefficient vectors only).
 * Needs srs to hold n + blinders points; costs one group DFT of size n (seconds at 2^22) and a second window table.
 * Free with jfhip_srs_free. */
int32_t jfhip_srs_lagrange(const jfhip_srs *srs, uint32_t log_n, uint32_t blinders, int32_t window_bits, jfhip_srs **out);
/* Same, from n packed affine points already in device memory. */
int32_t jfhip_srs_load_device(int32_t curve_id, const void *d_points_packed, size_t n, int32_t window_bits,
                              jfhip_srs **out);
/* Replaces the `powers_of_g` half of UnivariateUniversalParams::deserialize_{compressed,uncompressed}[_unchecked]
 * (derive(CanonicalDeserialize) on the struct whose first field is `powers_of_g: Vec<E::G1Affine>`,
 * pcs/src/univariate_kzg/srs.rs:17-31) + load_prover_param_to_gpu: `bytes` are n
 * consecutive ark-serialize G1Affine records (the Vec's 8-byte length prefix already skipped): compressed (32 B BN254,
 * 48 B BLS12-381) or uncompressed (64 / 96 B).  Decoding -- canonical -> Montgomery, y = sqrt(x^3 + b) with the sign
 * the flag bits select, and with validate != 0 ark-serialize's Validate::Yes checks (on the curve; BLS12-381: in the
 * prime-order subgroup) -- runs on the device, one lane per point.  A bad record is JFHIP_ERR_INVALID naming its index
 * (ark's SerializationError::InvalidData).  window_bits as jfhip_srs_load. */
int32_t jfhip_srs_load_serialized(int32_t curve_id, const uint8_t *bytes, size_t len, size_t n, int32_t compressed,
                                  int32_t validate, int32_t window_bits, jfhip_srs **out);
/* Replaces gen_srs_for_testing  (pcs/src/univariate_kzg/srs.rs:175-225) for harnesses:
 * builds [beta^i] g, i < n, on the device (beta: Fr Montgomery limbs, g: packed affine). */
int32_t jfhip_srs_gen_for_testing(int32_t curve_id, const uint64_t *beta_mont, const uint64_t *g_xy, size_t n,
                                  int32_t window_bits, jfhip_srs **out);
int32_t jfhip_srs_free(jfhip_srs *srs);
/* Give back the derived data a handle caches beyond its window tables -- the FK23 transforms of jfhip_kzg_multi_open_rou
 * (at most JFHIP_FK_CACHE_BYTES per handle, default 4 GiB, least recently used out; a transform larger than the budget is
 * never kept) -- without freeing the handle; they are rebuilt on demand.  jfhip_cache_clear() does this for every live
 * handle.  No counterpart in the reference. */
int32_t jfhip_srs_release_caches(jfhip_srs *srs);
size_t jfhip_srs_len(const jfhip_srs *srs);
int32_t jfhip_srs_window_bits(const jfhip_srs *srs);
/* Copy packed affine points [offset, offset+n) of the SRS back to the host. */
int32_t jfhip_srs_read(const jfhip_srs *srs, size_t offset, size_t n, void *out_points_packed);

/* ---- MSM / KZG commit --------------------------------------------------------------------- */

/* Replaces UnivariateKzgPCS::commit -> E::G1::msm_bigint(&powers_of_g[offset..], coeffs)
 * .into_affine()  (pcs/src/univariate_kzg/mod.rs:90-125) and the GPU twin
 * gpu_commit_with_loaded_prover_param (:688-699).
 * scalars_mont: n Fr elements (Montgomery) in HOST memory; result = sum_i s_i * P[offset+i].
 * out_xy: packed affine result (host); *out_infinity = 1 for the identity (then out_xy = 0),
 * which the shim maps to G1Affine::zero() as :989-991 does.
 * n == 0 yields the identity (zero polynomial).  offset + n > srs length is
 * JFHIP_ERR_INVALID ("poly degree {} is larger than allowed {}", :100-106).
 * Synchronous with respect to the host: the result is valid on return.  From 2^20 terms on the scalars are uploaded
 * in 2 (4 from 2^22) ranges whose partial sums run while the next range is in flight (same result; environment
 * JFHIP_MSM_UPLOAD_CHUNKS=1 keeps one upload). */
int32_t jfhip_msm(const jfhip_srs *srs, size_t offset, const uint64_t *scalars_mont, size_t n, void *out_xy,
                  uint8_t *out_infinity, jfhip_stream *stream);

/* Replaces UnivariateKzgPCS::batch_commit (:128-140) / gpu_batch_commit_with_loaded_prover_param
 * (:731-746): `batch` independent commits against the same SRS, ragged lengths allowed
 * (the reference zero-pads to the longest, :844-864).  out_xy: batch packed points,
 * out_infinity: batch flags.  batch == 0 is OK and writes nothing (:706-708). */
int32_t jfhip_msm_batch(const jfhip_srs *srs, const uint64_t *const *scalars_mont, const size_t *lens, size_t batch,
                        void *out_xy, uint8_t *out_infinity, jfhip_stream *stream);

/* batch_commit on DEVICE scalars (the prover keeps its polynomials in HBM): d_scalars is a HOST array
 * of `batch` device pointers; d_out receives `batch` records of ceil16(2*fq_bytes + 4) bytes, each a
 * packed affine point followed by a u32 infinity flag.  Asynchronous on `stream`.  The commits of a batch share ONE
 * pipeline (the bucket id is widened by the polynomial index: one sort, one bucket accumulation, one fix-up / fold /
 * sum tree and one normalisation launch for up to 16 polynomials), so the latency-bound tail of an MSM is paid once per
 * batch; batches too large for that (more than 2^30 digit entries) run one after another on internal auxiliary streams
 * that fork from and join into `stream`.  Environment JFHIP_MSM_FUSED=0 forces the latter (same results). */
int32_t jfhip_msm_batch_device(const jfhip_srs *srs, const void *const *d_scalars_mont, const size_t *lens,
                               size_t batch, void *d_out, jfhip_stream *stream);

/* Replaces commit_on_gpu (:900-930, 1001-1040): scalars already in DEVICE memory
 * (n Montgomery Fr elements), asynchronous on `stream`; the result is left on the device as
 * one packed affine point followed by a u32 infinity flag (d_out: 2*fq_bytes + 4 bytes,
 * 16-byte aligned).  Pair with jfhip_stream_sync + a device->host copy
 * (load_commitments_to_host, :934-964). */
int32_t jfhip_msm_device(const jfhip_srs *srs, size_t offset, const void *d_scalars_mont, size_t n, void *d_out,
                         jfhip_stream *stream);

/* Variable-base MSM on HOST points: sum_i s_i * P_i for n packed affine points (Montgomery) that are NOT an SRS -- the
 * verifier side: ScalarsAndBases::multi_scalar_mul -> VariableBaseMSM::msm_bigint (plonk/src/proof_system/structs.rs:
 * 888-896) as Verifier::batch_verify_opening_proofs calls it (verifier.rs:231,251; ~30-60 commitments per proof) and
 * the combinations of UnivariateKzgPCS::batch_verify (pcs/src/univariate_kzg/mod.rs:244-300).  Builds a throw-away window table for the points and runs the SRS path;
 * out_xy / out_infinity as jfhip_msm.  n == 0 yields the identity. */
int32_t jfhip_msm_small(int32_t curve_id, const uint64_t *points_xy, const uint64_t *scalars_mont, size_t n, void *out_xy,
                        uint8_t *out_infinity, jfhip_stream *stream);

/* Multi-GPU leg (no counterpart in the reference, which is single-device):
 * the un-normalised partial sum of this rank's point range as an XYZZ point
 * (4*fq_bytes, device memory), to be exchanged with RCCL all_gather and folded with
 * jfhip_g1_sum_partials on every rank. */
int32_t jfhip_msm_partial_device(const jfhip_srs *srs, size_t offset, const void *d_scalars_mont, size_t n,
                                 void *d_out_xyzz, jfhip_stream *stream);
/* The same for a batch_commit round (reference pcs/src/univariate_kzg/mod.rs:128-140 has no multi-device form): `batch`
 * partial sums against one SRS, entry b over the points [offsets[b], offsets[b] + lens[b]) (offsets == NULL: all start
 * at 0), written back to back as XYZZ elements in d_out_xyzz.  The entries share one sort / accumulate / fold pipeline
 * like jfhip_msm_batch_device.  d_scalars_mont and offsets / lens are HOST arrays; asynchronous on `stream`. */
int32_t jfhip_msm_batch_partial_device(const jfhip_srs *srs, const size_t *offsets, const void *const *d_scalars_mont,
                                       const size_t *lens, size_t batch, void *d_out_xyzz, jfhip_stream *stream);
/* Sum `count` XYZZ partials (device) and normalise: d_out as in jfhip_msm_device. */
int32_t jfhip_g1_sum_partials(int32_t curve_id, const void *d_partials_xyzz, size_t count, void *d_out,
                              jfhip_stream *stream);

/* Replaces UnivariateKzgPCS::open  (pcs/src/univariate_kzg/mod.rs:144-180): witness
 * polynomial (f(X) - f(z)) / (X - z) by synthetic division on the device, its commitment,
 * and eval = f(z).  coeffs_mont: n Fr coefficients (host), point_mont: z (host).
 * n == 0: proof = identity, eval = 0. */
int32_t jfhip_kzg_open(const jfhip_srs *srs, const uint64_t *coeffs_mont, size_t n, const uint64_t *point_mont,
                       void *out_proof_xy, uint8_t *out_infinity, uint64_t *out_eval_mont, jfhip_stream *stream);

/* Replaces UnivariatePCS::multi_point_open (pcs/src/univariate_kzg/mod.rs:416-468): one proof for the openings of one
 * polynomial at num_points points z_i (host, Montgomery, 4 limbs each): out_evals_mont[i] = f(z_i) (:433-436), and
 * proof = commit(f / prod_i (X - z_i)), the polynomial quotient with the remainder discarded (:443-464), computed on the
 * device as num_points successive synthetic divisions followed by one MSM.  num_points == 0 is JFHIP_ERR_INVALID with
 * the reference's message "no point to evaluate and open" (:421-425).  n <= num_points (and n == 0): the quotient is
 * zero, proof = identity.  num_points == 1 is jfhip_kzg_open.  The reference warns against many points (:415). */
int32_t jfhip_kzg_multi_point_open(const jfhip_srs *srs, const uint64_t *coeffs_mont, size_t n,
                                   const uint64_t *points_mont, size_t num_points, void *out_proof_xy,
                                   uint8_t *out_infinity, uint64_t *out_evals_mont, jfhip_stream *stream);

/* Replaces UnivariatePCS::multi_open_rou_proofs (pcs/src/univariate_kzg/mod.rs:328-395, FK23): the opening proofs
 * of one polynomial at the first num_points points w^j of the domain of 2^log_domain elements, proofs[j] =
 * commit((f(X) - f(w^j)) / (X - w^j)).  The h-polynomial (compute_h_poly_parallel / compute_h_poly_in_fk23, :572-650) is
 * a Toeplitz product over group elements, run as group-element DFTs (pcs/src/toeplitz.rs:43-66, pcs/src/poly.rs:
 * 140-155) on the device, followed by the group DFT over the domain.  coeffs_mont: n Fr coefficients (host).  A
 * polynomial whose h has more coefficients than the domain is JFHIP_ERR_INVALID with poly.rs:145-150's message.  The
 * evaluations f(w^j) (multi_open_rou_evals, :397-417) are jfhip_ntt of the coefficients.  The transform of the SRS prefix
 * depends on the handle and the degree only: it is kept on the handle after the first call of a shape (at most eight
 * shapes, 2^(log of the convolution length) XYZZ elements each) and freed with it. */
int32_t jfhip_kzg_multi_open_rou(const jfhip_srs *srs, const uint64_t *coeffs_mont, size_t n, uint32_t log_domain,
                                 size_t num_points, void *out_proofs_xy, uint8_t *out_infinity, jfhip_stream *stream);

/* ---- Multilinear KZG --------------------------------------------------- */

/* MultilinearKzgPCS::commit / batch_commit (pcs/src/multilinear_kzg/mod.rs:123-174) are
 * E::G1::msm_bigint(&prover_param.0.powers_of_g[ignored].evals, evaluations).into_affine(): load that level's `evals`
 * with jfhip_srs_load and call jfhip_msm / jfhip_msm_device on the evaluation table -- no entry point of their own.
 *
 * Replaces open_internal (pcs/src/multilinear_kzg/mod.rs:288-358): nv rounds over the evaluation table,
 *   q[b] = f[2b + 1] - f[2b],  r[b] = f[2b] + q[b] * point[i],  proofs[i] = msm(levels[i], q),  f = r,
 * on the device.  levels[i]: handle of prover_param.powers_of_g[ignored + i].evals (>= 2^(nv - 1 - i) points;
 * ignored = prover_param.num_vars - nv + 1, :317).  evals_mont: 2^nv Fr values (host, Montgomery), point_mont: nv Fr
 * values.  out_proofs_xy / out_infinity: nv packed affine points and flags; *out_eval_mont = f(point), the value
 * `polynomial.evaluate(point)` of :355.  nv == 0 writes the evaluation only. */
int32_t jfhip_mle_open(const jfhip_srs *const *levels, size_t nv, const uint64_t *evals_mont, const uint64_t *point_mont,
                       void *out_proofs_xy, uint8_t *out_infinity, uint64_t *out_eval_mont, jfhip_stream *stream);
/* One such round on DEVICE memory (mod.rs:330-338), asynchronous on `stream`: d_f holds 2 * half values, d_q and d_r
 * receive half values each (d_r may not alias d_f). */
int32_t jfhip_mle_fold_device(int32_t curve_id, const void *d_f, size_t half, const uint64_t *point_mont, void *d_q,
                              void *d_r, jfhip_stream *stream);

/* ---- NTT ---------------------------------------------------------------------------------- */

/* Replaces Radix2EvaluationDomain::{fft, ifft, ifft_in_place} and get_coset(offset).{fft,ifft}
 * (ark-poly 0.5) at the reference call sites plonk/src/proof_system/prover.rs:534-573,658-660
 * and relation/src/constraint_system.rs:1369,1386,1418,1437,1454,1464,1473,1482,1491,1573,
 * 1621-1622.
 * data: `batch` polynomials, each 2^log_n Fr elements (Montgomery), in HOST memory,
 * transformed in place, natural order in and out.
 *   inverse == 0: out[i] = sum_j in[j] * (h * w^i)^j
 *   inverse != 0: the inverse map (includes the n^{-1} and h^{-j} scaling)
 * w = Fr::TWO_ADIC_ROOT_OF_UNITY^(2^(s - log_n)); h = *coset_offset_mont, or 1 when NULL.
 * in_len <= 2^log_n: only the first in_len elements of each polynomial are read, the rest
 * are taken as zero (ark-poly zero-extends short inputs); in_len == 0 means 2^log_n. */
int32_t jfhip_ntt(int32_t curve_id, uint64_t *data, uint32_t log_n, size_t batch, int32_t inverse,
                  const uint64_t *coset_offset_mont, size_t in_len, jfhip_stream *stream);
/* Same on DEVICE memory, asynchronous on `stream`; d_data holds batch * 2^log_n elements. */
int32_t jfhip_ntt_device(int32_t curve_id, void *d_data, uint32_t log_n, size_t batch, int32_t inverse,
                         const uint64_t *coset_offset_mont, size_t in_len, jfhip_stream *stream);

/* ---- PLONK quotient evaluations (next row of the path) ----------------- */

/* Replaces the per-point loop of Prover::compute_quotient_polynomial
 * (plonk/src/proof_system/prover.rs:587-655, gate part :665-699, copy-constraint part :709-752)
 * for one TurboPlonk instance (5 wire types, 13 selectors, no Plookup):
 *     d_quot[i] += alpha_base * ((t_circ + t_perm_1) * z_h_inv[i mod (m/n)] + t_perm_2)
 * All inputs are DEVICE arrays of coset evaluations over the size-2^log_m quotient domain with
 * offset Fr::GENERATOR (what jfhip_ntt_device leaves), laid out [poly][2^log_m]:
 * d_selectors 13 polys in the reference's order (q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc),
 * d_sigmas 5, d_wires 5, d_prod_perm 1, d_pub_input 1.  k_mont: the 5 coset representatives
 * pk.k(); challenges_mont: alpha, beta, gamma, alpha_base (4 x 4 limbs), all host, Montgomery.
 * d_quot is accumulated into (zero it for the first instance).  Asynchronous on `stream`. */
int32_t jfhip_plonk_quotient_turbo_device(int32_t curve_id, uint32_t log_n, uint32_t log_m, const void *d_selectors,
                                          const void *d_sigmas, const void *d_wires, const void *d_prod_perm,
                                          const void *d_pub_input, const uint64_t *k_mont,
                                          const uint64_t *challenges_mont, void *d_quot, jfhip_stream *stream);

/* UltraPlonk (Plookup) variant of the same loop: 6 wire types, 14 selectors (the reference's order with
 * q_lookup last, relation/src/constraint_system.rs:1091-1107), 6 sigmas, and the Plookup part
 * compute_quotient_plookup_contribution (prover.rs:766-882, eval_merged_table /
 * eval_merged_lookup_witness plonk/src/proof_system/structs.rs:899-931):
 *     d_quot[i] += alpha_base * ((t_circ + t_perm_1 + t_lookup_1) * z_h_inv[i mod (m/n)] + t_perm_2 + t_lookup_2)
 * d_plookup: 7 polys [range_table, key_table, table_dom_sep, q_dom_sep, h_1, h_2, prod_lookup]
 * (the coset FFTs of prover.rs:559-573).  k_mont: 6 coset representatives; challenges_mont: alpha, beta,
 * gamma, tau, alpha_base (5 x 4 limbs).  For batch_prove the caller advances alpha_base by alpha^7 per
 * instance (prover.rs:651-655).  Asynchronous on `stream`. */
int32_t jfhip_plonk_quotient_ultra_device(int32_t curve_id, uint32_t log_n, uint32_t log_m, const void *d_selectors,
                                          const void *d_sigmas, const void *d_wires, const void *d_prod_perm,
                                          const void *d_pub_input, const void *d_plookup, const uint64_t *k_mont,
                                          const uint64_t *challenges_mont, void *d_quot, jfhip_stream *stream);

/* The same loop over ONE SLICE of the quotient coset (SURVEY.md 8e "coset sharding"): with G ranks (G divides
 * m/n), rank g owns the points i = g (mod G) of the reference's coset {h w_m^i} (prover.rs:524-527, 587-642),
 * which form the coset of size 2^log_m = m/G with offset h w_m^g; the "next" point (i + m/n) mod m
 * (prover.rs:594,610) stays inside the slice.  Arguments as the two functions above with that slice's domain:
 * log_m is the SLICE's size, coset_offset_mont its offset (4 limbs, Montgomery; Fr::GENERATOR itself gives
 * the functions above), inputs laid out [poly][2^log_m]; num_wire_types 5 (d_plookup ignored, challenges alpha,
 * beta, gamma, alpha_base) or 6 (challenges alpha, beta, gamma, tau, alpha_base).  No counterpart call in the
 * reference, which evaluates the whole coset in one rayon loop. */
int32_t jfhip_plonk_quotient_coset_device(int32_t curve_id, uint32_t num_wire_types, uint32_t log_n, uint32_t log_m,
                                          const uint64_t *coset_offset_mont, const void *d_selectors,
                                          const void *d_sigmas, const void *d_wires, const void *d_prod_perm,
                                          const void *d_pub_input, const void *d_plookup, const uint64_t *k_mont,
                                          const uint64_t *challenges_mont, void *d_quot, jfhip_stream *stream);

/* jfhip_plonk_quotient_coset_device for circuits that leave gate kinds unused (the reference evaluates every term at
 * every point, prover.rs:665-699): bit j of zero_selector_mask set = selector polynomial j (the reference's order) is
 * identically zero -- its row of d_selectors is never read, so it need not hold coset evaluations, and the products it
 * would scale are not computed; d_pub_input may be NULL for an empty public input (pub_inp_poly = 0).  Same values in
 * d_quot as the unmasked call on zero rows.  coset_offset_mont = Fr::GENERATOR and log_m = the full quotient domain give
 * the reference's coset. */
int32_t jfhip_plonk_quotient_masked_device(int32_t curve_id, uint32_t num_wire_types, uint32_t log_n, uint32_t log_m,
                                           const uint64_t *coset_offset_mont, const void *d_selectors,
                                           const void *d_sigmas, const void *d_wires, const void *d_prod_perm,
                                           const void *d_pub_input, const void *d_plookup, const uint64_t *k_mont,
                                           const uint64_t *challenges_mont, uint32_t zero_selector_mask, void *d_quot,
                                           jfhip_stream *stream);

/* ---- grand-product builders (next row) ------------------------------------ */

/* Replaces the serial loop (one field division per row) of
 * Arithmetization::compute_prod_permutation_polynomial, relation/src/constraint_system.rs:1394-1420,
 * up to (not including) its final ifft_in_place:
 *     d_out[0] = 1,  d_out[j+1] = d_out[j] * prod_i (w[i][j] + gamma + beta * id[i][j])
 *                                          / prod_i (w[i][j] + gamma + beta * sigma[i][j]),  j < n - 1.
 * DEVICE arrays laid out [wire type][n], n = 2^log_n: d_wire_vals = witness[wire_variable(i, j)],
 * d_id_perm = extended_id_permutation[i*n + j], d_sigma_vals = extended_id_permutation at the permuted
 * position (perm_i*n + perm_j).  beta, gamma: host, Montgomery.  Asynchronous on `stream`. */
int32_t jfhip_plonk_perm_product_device(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types,
                                        const void *d_wire_vals, const void *d_id_perm, const void *d_sigma_vals,
                                        const uint64_t *beta_mont, const uint64_t *gamma_mont, void *d_out,
                                        jfhip_stream *stream);

/* Same for Arithmetization::compute_lookup_prod_polynomial, relation/src/constraint_system.rs:1516-1575:
 * d_lookup_wire[n] = merged lookup witness values (merged_lookup_wire_value), d_table[n] = merged lookup
 * table, d_sorted[2n - 1] = the sorted vector; d_out[n] with d_out[0] = d_out[n-1] = 1. */
int32_t jfhip_plonk_lookup_product_device(int32_t curve_id, uint32_t log_n, const void *d_lookup_wire,
                                          const void *d_table, const void *d_sorted, const uint64_t *beta_mont,
                                          const uint64_t *gamma_mont, void *d_out, jfhip_stream *stream);

/* Replaces Arithmetization::compute_merged_lookup_table and the merged_lookup_wire_value calls of
 * compute_lookup_sorted_vec_polynomials / compute_lookup_prod_polynomial (relation/src/constraint_system.rs:1495-1514,
 * 1552-1553, 1601-1602, helpers :1648-1687) on DEVICE data, for every row i of the size-2^log_n domain:
 *     d_merged_table[i]  = range[i] + q_lookup[i] tau (table_dom_sep[i] + tau (key[i] + tau (w3[i] + tau w4[i])))
 *     d_merged_lookup[i] = w5[i]    + q_lookup[i] tau (q_dom_sep[i]     + tau (w0[i]  + tau (w1[i] + tau w2[i])))
 * d_wire_vals: [6][n] witness values per wire type (what jfhip_witness_gather_device leaves); d_table_vals: [5][n] rows
 * range_table, key_table, table_dom_sep, q_dom_sep, q_lookup (values on H, fixed by the circuit); tau: host, Montgomery.
 * Asynchronous on `stream`. */
int32_t jfhip_plonk_lookup_merge_device(int32_t curve_id, uint32_t log_n, const void *d_wire_vals,
                                        const void *d_table_vals, const uint64_t *tau_mont, void *d_merged_table,
                                        void *d_merged_lookup, jfhip_stream *stream);

/* Replaces the HashMap merge of compute_lookup_sorted_vec_polynomials (relation/src/constraint_system.rs:1597-1620)
 * on DEVICE data: d_sorted = every row of d_table (n_table field elements of 32 bytes, the merged lookup table) in
 * table order, each followed by as many copies as d_lookups (n_lookups rows, the first n - 1 merged lookup-wire values)
 * holds of it, counted at the value's FIRST occurrence in the table; n_table + n_lookups rows.  A lookup value that the
 * table does not hold is JFHIP_ERR_INVALID with the reference's message ("The sorted vector has wrong length, some
 * lookup variables might be outside the table", :1613-1615).  Synchronous: the status is read back. */
int32_t jfhip_plonk_lookup_sorted_device(const void *d_table, size_t n_table, const void *d_lookups, size_t n_lookups,
                                         void *d_sorted, jfhip_stream *stream);

/* ---- device-resident polynomial glue (next row) -------------------------- */

/* Replaces the witness-polynomial division of open() / compute_batched_witness_polynomial_commitment
 * (pcs/src/univariate_kzg/mod.rs:153-158, plonk/src/proof_system/prover.rs:483-485) and
 * polynomial.evaluate (mod.rs:174, prover.rs:205-229) on DEVICE data: for n coefficients
 * d_coeffs and the host point z, d_out (n elements) receives  d_out[0] = f(z)  and
 * d_out[1..n) = coefficients of (f(X) - f(z)) / (X - z).  Asynchronous on `stream`. */
int32_t jfhip_poly_divide_linear_device(int32_t curve_id, const void *d_coeffs, size_t n, const uint64_t *point_mont,
                                        void *d_out, jfhip_stream *stream);

/* The entry point that allocates nothing.  It has NO counterpart in the reference; it exists for callers that cannot
 * allocate device memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * arguments and same results as jfhip_poly_divide_linear_device, but the scan's workspace (chunk values, suffix arrays,
 * per-level multipliers) is a static device array of 8.13 MiB that every GPU the library runs on holds until the process
 * ends, the point travels as a kernel argument (no host->device copy), and the call makes no hipMalloc / hipFree or
 * other allocating call, host or device.  Limits: n <= 2^23 coefficients (larger n: JFHIP_ERR_INVALID, use the call
 * above); `stream` must be a handle made at set-up (NULL is JFHIP_ERR_INVALID, because the per-thread default stream is
 * created on first use).  jfhip_init() on the device does the one-time set-up; without it the first call on a device
 * does.  The array used is that of the GPU `stream` belongs to (hipStreamGetDevice), whatever the calling thread's current
 * device.  All calls on one GPU share that GPU's array: the library orders them (a per-device host mutex around the
 * enqueue, and each call's stream waits on an event recorded after the previous call's last kernel -- no allocation
 * either), except for streams under graph capture, whose callers must serialise calls that could overlap on one GPU.
 * The array keeps the previous call's contents; the scan never reads an entry it has not written in the same call, so
 * nothing needs clearing.  Asynchronous on `stream`. */
int32_t jfhip_poly_divide_linear_noalloc_device(int32_t curve_id, const void *d_coeffs, size_t n,
                                                const uint64_t *point_mont, void *d_out, jfhip_stream *stream);

/* Replaces Prover::compute_evaluations / polynomial.evaluate (plonk/src/proof_system/prover.rs:205-229,
 * pcs/src/univariate_kzg/mod.rs:174): d_out[k] = d_polys[k](z) for `count` (<= 32) DEVICE polynomials at
 * one host point; d_out: count field elements on the device.  Asynchronous on `stream`. */
int32_t jfhip_poly_eval_batch_device(int32_t curve_id, const void *const *d_polys, const size_t *lens, size_t count,
                                     const uint64_t *point_mont, void *d_out, jfhip_stream *stream);

/* Replaces mul_poly + the polynomial sums of the linearisation / batched-opening polynomials
 * (plonk/src/proof_system/prover.rs:299-352, 470-488, 1090-1099):
 * d_out[i] = sum_k coeffs[k] * d_polys[k][i], i < n; polynomial k contributes zero beyond lens[k].
 * d_polys: host array of `count` (<= 32) device pointers; coeffs_mont: count host scalars. */
int32_t jfhip_poly_lincomb_device(int32_t curve_id, const void *const *d_polys, const size_t *lens, size_t count,
                                  const uint64_t *coeffs_mont, void *d_out, size_t n, jfhip_stream *stream);

/* Replaces the witness gather of Arithmetization::compute_wire_polynomials
 * (relation/src/constraint_system.rs:1422-1444: `self.witness[var]` for every (wire type, gate) before the ifft) on DEVICE
 * data: d_out[i] = d_witness[d_wire_variables[i]], i < count, rows of 32 bytes (one Fr, Montgomery limbs as everywhere);
 * d_witness has num_vars rows (< 2^32), d_wire_variables is a DEVICE array of uint32 (wire type major: index
 * wire_type * n + gate, the layout jfhip_ntt_device takes as a batch).  check != 0: synchronises and returns
 * JFHIP_ERR_INVALID if any index was >= num_vars (such rows read variable 0); check == 0: asynchronous on `stream`. */
int32_t jfhip_witness_gather_device(const void *d_witness, size_t num_vars, const uint32_t *d_wire_variables,
                                    size_t count, void *d_out, int32_t check, jfhip_stream *stream);

/* ---- field / group vector ops (test and glue surface) -------------------------------------- */

/* field_id: 0 BN254 Fr, 1 BN254 Fq, 2 BLS12-381 Fr, 3 BLS12-381 Fq.
 * op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 inv(a) (inv(0)=0), 5 from_mont(a), 6 to_mont(a);
 *     7 mul, 8 square(a), 9 sub, 10 ((ab - 2a)(a^2 - b))b - ab, 11 ab + (a^2 - b)b computed in the kernels' internal
 *     unsaturated 29-bit-limb form (csrc/fp29.hpp) and converted back: same results as the saturated ops.
 *     12, 13: inv(a) by the kernels' two Euclid-type routines (batched division steps; binary): same results as op 4.
 * Host arrays of n elements; exercises the same device arithmetic the MSM/NTT kernels use. */
int32_t jfhip_field_op(int32_t field_id, int32_t op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);

/* op: 0 out = P + Q (packed affine, host), 1 out = k * P (k: Fr Montgomery, passed as Q);
 *     2, 3: the same two through the MSM tail's group law on lazily reduced 29-bit limbs (csrc/g29.hpp p29_add,
 *     p29_double), 4: k * P by the group-element DFT's non-adjacent-form multiplication (p29_mul_scalar): same results. */
int32_t jfhip_g1_op(int32_t curve_id, int32_t op, const uint64_t *P, const uint64_t *Q, uint64_t *out);

/* Host-only: the Keccak-f[1600] permutation on 25 little-endian u64 lanes, in place.  Used by the host side of the
 * prover for the Merlin transcript behind `StandardTranscript` (plonk/src/transcript/standard.rs:14-39, crate merlin);
 * no device work. */
int32_t jfhip_keccak_f1600(uint64_t *state25);

/* Peak-rate probes used by bench.py for the roofline denominators:
 * kind 0: v_mad_u64_u32 multiply-adds/s (INT32 VALU), kind 1: device copy bytes/s,
 * kind 2: Montgomery multiplications/s of the field given in `arg` (field_id); 3: the same on the kernels' 29-bit limbs;
 * 4 / 5: random 64-byte gathers/s over a table of `arg` GiB (dependent / independent issue);
 * 6: multiplications/s by a CONSTANT through the VALU (fp_mul_const29, the NTT's twiddle product; arg = field_id 0 or 2);
 * 7: the same map through the int8 matrix cores (v_mfma_i32_32x32x32_i8 against the constant's Toeplitz byte matrix, with
 *    the carry normalisations and the Montgomery step) as an instruction-mix lower bound -- an exploratory measurement
 *    (DESIGN.md 3.3), no product path uses the matrix cores. */
int32_t jfhip_probe(int32_t kind, int32_t arg, double *out_rate);

/* Device markers.  With the environment variable JFHIP_MARKERS=1 the library brackets its entry points with roctx ranges
 * named after the reference's own timers ("Committing to polynomial of degree {}", "batch commit {} polynomials",
 * "Opening polynomial of degree {}": pcs/src/univariate_kzg/mod.rs:96-123, 134, 158, 426-430; ark_std start_timer!), so a
 * `rocprofv3 --kernel-trace --marker-trace` run segments by name.  These two let the host side add its own ranges (the
 * prover's rounds) through the same roctx library; no-ops when markers are off or no roctx library is found. */
int32_t jfhip_marker_push(const char *name);
int32_t jfhip_marker_pop(void);

/* Time (ms) the library's kernels of the last MSM / NTT call on this stream took on the
 * device, per phase, measured with HIP events on the stream the kernels ran on.
 * phases: caller array of `cap` doubles; returns the number of phases written.
 * Only recorded while profiling is enabled with jfhip_set_profiling(1). */
int32_t jfhip_set_profiling(int32_t enabled);
int32_t jfhip_last_phase_ms(jfhip_stream *stream, double *phases, int32_t cap);
const char *jfhip_phase_name(int32_t kind /*0 msm, 1 ntt*/, int32_t idx);

#ifdef __cplusplus
}
#endif
#endif /* JFHIP_H */
