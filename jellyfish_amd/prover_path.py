"""Device-side operation sequence of one TurboPlonk proof (host orchestration only).

Mirrors the order in which `PlonkKzgSnark::batch_prove_internal` drives the accelerated path for one
instance (reference plonk/src/proof_system/snark.rs:163-430 -> prover.rs:77-413; call stack in
SURVEY.md section 3.1): per proof 7 size-n iNTTs, 25 forward coset NTTs of size 8n with zero
extension, the quotient evaluation, 1 inverse coset NTT of size 8n, 13 commitments of ~n points,
10 evaluations, 2 linear combinations and 2 divisions by (X - z).

What stays on the host in the reference stays out of this sequence: the transcript (challenges are
inputs here) and the RNG (blinding coefficients are inputs).  The grand products of
compute_prod_permutation_polynomial / compute_lookup_prod_polynomial
(relation/src/constraint_system.rs:1394-1420, 1516-1575) run on the device from the wire values and
the permutation / sorted-vector data.  Every polynomial stays resident in HBM between steps; only 13 group elements and 10
field elements would cross PCIe.  Used by bench.py with synthetic inputs of the reference bench's
shape (plonk/benches/bench.rs:29-46); all arithmetic is in libjfhip.
"""
from __future__ import annotations

import numpy as np
import torch

from . import lib as L

NUM_WIRE_TYPES = 5      # relation/src/constants.rs:18-22 (GATE_WIDTH + 1)
NUM_SELECTORS = 13      # N_TURBO_PLONK_SELECTORS


def domain_size_ratio(n: int, num_wire_types: int) -> int:
    """plonk/src/constants.rs:18-20"""
    return (num_wire_types * (n + 1) + 2) // n + 1


def quot_domain_size(n: int, num_wire_types: int) -> int:
    """Prover::new (prover.rs:57-68): GeneralEvaluationDomain::new(n * ratio) rounds up to 2^k"""
    want = n * domain_size_ratio(n, num_wire_types)
    m = 1
    while m < want:
        m <<= 1
    return m


def _dev(a) -> torch.Tensor:
    """field elements -> flat int64 device tensor (numpy (count, 4) uint64, or already a device tensor)"""
    if isinstance(a, torch.Tensor):
        return a.reshape(-1).cuda()
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1)).cuda()


def _host(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        return a.cpu().numpy().view(np.uint64).reshape(-1, 4)
    return a


class PlonkHotPath:
    """One PLONK instance of n gates on the current device (curve: L.BN254 / L.BLS12_381).

    ultra = False: TurboPlonk, 5 wire types, 13 selectors: 13 commitments.
    ultra = True:  UltraPlonk with Plookup, 6 wire types, 14 selectors, 4 table polynomials in the
    proving key and the online polynomials h_1, h_2, prod_lookup (rounds 1.5 / 2.5, prover.rs:99-178):
    11 iNTTs of size n, 35 coset NTTs of size 8n, 18 commitments."""

    def __init__(self, curve: int, log_n: int, srs: "L.Srs", rand_fr, stream: "L.Stream | None" = None,
                 sharded: bool = False, group=None, ultra: bool = False):
        """rand_fr(count, seed) -> (count, 4) uint64 Montgomery field elements (synthetic inputs).
        sharded: torch.distributed is initialised, `srs` is replicated on every rank, and every
        commitment's coefficient range is partitioned over the ranks (BASELINE config 4: "MSM point
        ranges sharded across 8 MI355X"); one all_gather of partial sums per commit round.  The
        transforms and the quotient are computed redundantly on every rank in this mode."""
        self.curve, self.log_n, self.ultra = curve, log_n, ultra
        self.sharded, self.group = sharded, group
        self.W = W = 6 if ultra else NUM_WIRE_TYPES
        self.NSEL = NSEL = 14 if ultra else NUM_SELECTORS
        self.n = n = 1 << log_n
        self.m = m = quot_domain_size(n, W)
        self.log_m = m.bit_length() - 1
        self.srs = srs
        assert len(srs) >= n + 3, "SRS must cover degree n + 2"
        self.stream = stream or L.Stream(wrap_raw=torch.cuda.current_stream().cuda_stream)
        if sharded:
            from . import dist as jdist

            self._jdist = jdist
            self._engine = jdist.HipMsmEngine(L, srs, self.stream)
        self.fq_bytes = 8 * L.FQ_LIMBS64[curve]
        self.gen = L.field_op(0 if curve == L.BN254 else 2, "to_mont",
                              np.array([5 if curve == L.BN254 else 7, 0, 0, 0], dtype=np.uint64))
        # proving key, coefficient form, resident (preprocess(), snark.rs:492-580)
        self.selectors = _dev(rand_fr(NSEL * n, 11)).view(NSEL, n * 4)
        self.sigmas = _dev(rand_fr(W * n, 12)).view(W, n * 4)
        self.k = _host(rand_fr(W, 13))
        # witness-dependent inputs: wire values and public input values on H; the extended identity permutation and
        # its permuted values (extended_id_permutation / wire_permutation, relation/src/constraint_system.rs:1403-1417)
        self.wire_vals = _dev(rand_fr(W * n, 14)).view(W, n * 4)
        self.id_perm_vals = _dev(rand_fr(W * n, 15)).view(W, n * 4)
        self.sigma_vals = _dev(rand_fr(W * n, 25)).view(W, n * 4)
        self.pub_vals = _dev(rand_fr(n, 16))
        # host-side randomness and transcript outputs, fixed for the synthetic run
        self.blind = _dev(rand_fr(64, 17))
        self.ch = _host(rand_fr(5 if ultra else 4, 18))   # alpha, beta, gamma, [tau,] alpha_base
        self.zeta = _host(rand_fr(1, 19))[0]
        self.zeta_omega = _host(rand_fr(1, 20))[0]
        self.lin_coeffs = _host(rand_fr(32, 21))
        if ultra:
            # plookup_pk: range_table, key_table, table_dom_sep, q_dom_sep polynomials (snark.rs:508-521)
            self.table_polys = _dev(rand_fr(4 * n, 22)).view(4, n * 4)
            # merged lookup witness, merged table and the sorted vector (2n - 1 values; h_1 = sorted[..n],
            # h_2 = sorted[n-1..]) -- the hash-map merge sort is host work in the reference
            # (relation/src/constraint_system.rs:1577-1624) and an input here
            self.lookup_wire_vals = _dev(rand_fr(n, 23))
            self.merged_table_vals = _dev(rand_fr(n, 24))
            self.sorted_vals = _dev(rand_fr(2 * n - 1, 26))
        # working storage
        ln = n + 4                          # room for the n+2 / n+3 coefficients of a blinded polynomial
        self.ln = ln
        self.wire_polys = torch.zeros(W, ln * 4, dtype=torch.int64, device="cuda")
        self.perm_poly = torch.zeros(ln * 4, dtype=torch.int64, device="cuda")
        self.pub_poly = torch.zeros(ln * 4, dtype=torch.int64, device="cuda")
        self.ntt_buf = torch.zeros(W + 2, n * 4, dtype=torch.int64, device="cuda")
        self.npolys = NSEL + 2 * W + 2 + (7 if ultra else 0)
        if ultra:
            self.h_polys = torch.zeros(2, ln * 4, dtype=torch.int64, device="cuda")
            self.plprod_poly = torch.zeros(ln * 4, dtype=torch.int64, device="cuda")
        self.coset = torch.zeros(self.npolys, m * 4, dtype=torch.int64, device="cuda")
        self.quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
        self.lin_poly = torch.zeros(ln * 4, dtype=torch.int64, device="cuda")
        self.open_poly = torch.zeros(ln * 4, dtype=torch.int64, device="cuda")
        self.scratch = torch.zeros(ln * 4, dtype=torch.int64, device="cuda")
        rec = (2 * self.fq_bytes + 4 + 15) // 16 * 16
        self.rec = rec
        self.ncommit = 18 if ultra else 13
        self.commits = torch.zeros(self.ncommit * rec, dtype=torch.uint8, device="cuda")
        self.nevals = 27 if ultra else 10
        self.evals = torch.zeros(self.nevals * 4, dtype=torch.int64, device="cuda")

    # ---- helpers ------------------------------------------------------------------------------
    def _commit_many(self, slot: int, polys):
        """batch_commit of coefficient tensors (int64, 4 words per coefficient) into commitment slots"""
        if not self.sharded:
            if len(polys) == 1:
                L.msm_device(self.srs, polys[0].data_ptr(), polys[0].numel() // 4,
                             self.commits.data_ptr() + slot * self.rec, stream=self.stream)
            else:
                L.msm_batch_device(self.srs, [p.data_ptr() for p in polys], [p.numel() // 4 for p in polys],
                                   self.commits.data_ptr() + slot * self.rec, stream=self.stream)
            return
        outs = self._jdist.sharded_commit_batch(self._engine, polys, group=self.group)
        for j, o in enumerate(outs):
            self.commits[(slot + j) * self.rec:(slot + j) * self.rec + o.numel()] = o

    def _blind(self, poly: torch.Tensor, count: int, which: int):
        """mask_polynomial (prover.rs:457-466) adds (b0 + b1 X ..)(X^n - 1): `count` extra top
        coefficients.  The synthetic run writes the extra coefficients (host RNG output) and leaves
        the two low ones: same shapes and sizes, the values are random either way."""
        n4 = self.n * 4
        poly[n4:n4 + 4 * count] = self.blind[8 * which:8 * which + 4 * count]

    # ---- the proof ----------------------------------------------------------------------------
    def prove(self) -> torch.Tensor:
        c, n, m, st = self.curve, self.n, self.m, self.stream
        n4 = n * 4
        W, NSEL, ultra = self.W, self.NSEL, self.ultra
        slot = 0
        # Round 1 (prover.rs:77-91): wire polynomials = iNTT of wire values, blind, commit; pub input poly
        self.ntt_buf[:W].copy_(self.wire_vals)
        self.ntt_buf[W].copy_(self.pub_vals)
        L.ntt_device(c, self.ntt_buf.data_ptr(), self.log_n, W + 1, inverse=True, stream=st)
        self.wire_polys[:, :n4].copy_(self.ntt_buf[:W])
        for j in range(W):
            self._blind(self.wire_polys[j], 2, j)
        self.pub_poly[:n4].copy_(self.ntt_buf[W])
        self._commit_many(slot, [self.wire_polys[j][:(n + 2) * 4] for j in range(W)])
        slot += W
        if ultra:
            # Round 1.5 (prover.rs:99-124): h_1, h_2 = iNTT of the sorted-vector halves, blind (bound 2), commit
            self.ntt_buf[0].copy_(self.sorted_vals[:n4])
            self.ntt_buf[1].copy_(self.sorted_vals[n4 - 4:2 * n4 - 4])
            L.ntt_device(c, self.ntt_buf.data_ptr(), self.log_n, 2, inverse=True, stream=st)
            self.h_polys[:, :n4].copy_(self.ntt_buf[:2])
            for j in range(2):
                self._blind(self.h_polys[j], 3, 6 + 2 * j)
            self._commit_many(slot, [self.h_polys[j][:(n + 3) * 4] for j in range(2)])
            slot += 2
        # Round 2 (prover.rs:128-142): permutation grand product on H (constraint_system.rs:1394-1420), iNTT, commit
        beta_gamma = self.ch[1:3]
        L.plonk_perm_product_device(c, self.log_n, W, self.wire_vals.data_ptr(), self.id_perm_vals.data_ptr(),
                                    self.sigma_vals.data_ptr(), beta_gamma[0], beta_gamma[1],
                                    self.ntt_buf[W + 1].data_ptr(), stream=st)
        L.ntt_device(c, self.ntt_buf[W + 1].data_ptr(), self.log_n, 1, inverse=True, stream=st)
        self.perm_poly[:n4].copy_(self.ntt_buf[W + 1])
        self._blind(self.perm_poly, 3, 10 if ultra else W)
        self._commit_many(slot, [self.perm_poly[:(n + 3) * 4]])
        slot += 1
        if ultra:
            # Round 2.5 (prover.rs:147-178): lookup grand product on H (constraint_system.rs:1516-1575), iNTT, blind, commit
            L.plonk_lookup_product_device(c, self.log_n, self.lookup_wire_vals.data_ptr(),
                                          self.merged_table_vals.data_ptr(), self.sorted_vals.data_ptr(), beta_gamma[0],
                                          beta_gamma[1], self.ntt_buf[0].data_ptr(), stream=st)
            L.ntt_device(c, self.ntt_buf.data_ptr(), self.log_n, 1, inverse=True, stream=st)
            self.plprod_poly[:n4].copy_(self.ntt_buf[0])
            self._blind(self.plprod_poly, 3, 12)
            self._commit_many(slot, [self.plprod_poly[:(n + 3) * 4]])
            slot += 1
        # Round 3 (prover.rs:183-198, 491-661): coset NTTs (zero extension on device), quotient, split, commit
        ln4 = (n + 3) * 4
        cs = self.coset
        o_sig, o_w, o_z, o_pi, o_lk = NSEL, NSEL + W, NSEL + 2 * W, NSEL + 2 * W + 1, NSEL + 2 * W + 2
        cs[:NSEL, :n4].copy_(self.selectors)
        cs[o_sig:o_w, :n4].copy_(self.sigmas)
        cs[:o_w, n4:ln4].zero_()
        cs[o_w:o_z, :ln4].copy_(self.wire_polys[:, :ln4])
        cs[o_z, :ln4].copy_(self.perm_poly[:ln4])
        cs[o_pi, :ln4].copy_(self.pub_poly[:ln4])
        if ultra:
            # prover.rs:559-573; stored in the quotient kernel's order range, key, table_dom_sep, q_dom_sep, h1, h2, prod
            cs[o_lk:o_lk + 4, :n4].copy_(self.table_polys)
            cs[o_lk:o_lk + 4, n4:ln4].zero_()
            cs[o_lk + 4:o_lk + 6, :ln4].copy_(self.h_polys[:, :ln4])
            cs[o_lk + 6, :ln4].copy_(self.plprod_poly[:ln4])
        L.ntt_device(c, cs.data_ptr(), self.log_m, self.npolys, inverse=False, coset_mont=self.gen, in_len=n + 3,
                     stream=st)
        self.quot.zero_()
        if ultra:
            L.plonk_quotient_ultra_device(c, self.log_n, self.log_m, cs[0].data_ptr(), cs[o_sig].data_ptr(),
                                          cs[o_w].data_ptr(), cs[o_z].data_ptr(), cs[o_pi].data_ptr(),
                                          cs[o_lk].data_ptr(), self.k, self.ch, self.quot.data_ptr(), stream=st)
        else:
            L.plonk_quotient_turbo_device(c, self.log_n, self.log_m, cs[0].data_ptr(), cs[o_sig].data_ptr(),
                                          cs[o_w].data_ptr(), cs[o_z].data_ptr(), cs[o_pi].data_ptr(), self.k, self.ch,
                                          self.quot.data_ptr(), stream=st)
        L.ntt_device(c, self.quot.data_ptr(), self.log_m, 1, inverse=True, coset_mont=self.gen, stream=st)
        # split_quotient_polynomial (prover.rs:894-943): W chunks of n + 2 coefficients
        chunks = [self.quot[j * (n + 2) * 4:(j + 1) * (n + 2) * 4] for j in range(W)]
        self._commit_many(slot, chunks)
        slot += W
        # Round 4 (prover.rs:205-229): evaluations of wires, sigmas (all but the last) at zeta and z at zeta*omega
        at_zeta = [(self.wire_polys[j], n + 2) for j in range(W)] + [(self.sigmas[j], n) for j in range(W - 1)]
        at_zw = [(self.perm_poly, n + 3)]
        if ultra:
            # Round 4.5 (prover.rs:233-296)
            rt, kt, tds, qds = (self.table_polys[j] for j in range(4))
            ql = self.selectors[NSEL - 1]
            at_zeta += [(rt, n), (kt, n), (self.h_polys[0], n + 3), (ql, n), (tds, n), (qds, n)]
            at_zw += [(self.plprod_poly, n + 3), (rt, n), (kt, n), (self.h_polys[0], n + 3), (self.h_polys[1], n + 3),
                      (ql, n), (self.wire_polys[3], n + 2), (self.wire_polys[4], n + 2), (tds, n)]
        L.poly_eval_batch_device(c, [p.data_ptr() for p, _ in at_zeta], [l for _, l in at_zeta], self.zeta,
                                 self.evals.data_ptr(), stream=st)
        L.poly_eval_batch_device(c, [p.data_ptr() for p, _ in at_zw], [l for _, l in at_zw], self.zeta_omega,
                                 self.evals.data_ptr() + len(at_zeta) * 32, stream=st)
        # linearisation polynomial (prover.rs:299-352, 946-1087): 13 gate selectors, z, last sigma, quotient chunks
        # (+ prod_lookup and h_2 for Plookup, prover.rs:1017-1087)
        ptrs = [self.selectors[j].data_ptr() for j in range(NUM_SELECTORS)]
        lens = [n] * NUM_SELECTORS
        ptrs += [self.perm_poly.data_ptr(), self.sigmas[W - 1].data_ptr()]
        lens += [n + 3, n]
        ptrs += [ch.data_ptr() for ch in chunks]
        lens += [n + 2] * W
        if ultra:
            ptrs += [self.plprod_poly.data_ptr(), self.h_polys[1].data_ptr()]
            lens += [n + 3, n + 3]
        L.poly_lincomb_device(c, ptrs, lens, self.lin_coeffs[:len(ptrs)], self.lin_poly.data_ptr(), n + 3, stream=st)
        # Round 5 (prover.rs:356-413, 470-488): batched opening polynomial at zeta, divide, commit; same at zeta*omega
        ptrs = [self.lin_poly.data_ptr()] + [p.data_ptr() for p, _ in at_zeta]
        lens = [n + 3] + [l for _, l in at_zeta]
        L.poly_lincomb_device(c, ptrs, lens, self.lin_coeffs[:len(ptrs)], self.open_poly.data_ptr(), n + 3, stream=st)
        L.poly_divide_linear_device(c, self.open_poly.data_ptr(), n + 3, self.zeta, self.scratch.data_ptr(), stream=st)
        self._commit_many(slot, [self.scratch[4:4 + (n + 2) * 4]])
        slot += 1
        if len(at_zw) == 1:
            L.poly_divide_linear_device(c, self.perm_poly.data_ptr(), n + 3, self.zeta_omega, self.scratch.data_ptr(),
                                        stream=st)
        else:
            L.poly_lincomb_device(c, [p.data_ptr() for p, _ in at_zw], [l for _, l in at_zw],
                                  self.lin_coeffs[:len(at_zw)], self.open_poly.data_ptr(), n + 3, stream=st)
            L.poly_divide_linear_device(c, self.open_poly.data_ptr(), n + 3, self.zeta_omega, self.scratch.data_ptr(),
                                        stream=st)
        self._commit_many(slot, [self.scratch[4:4 + (n + 2) * 4]])
        return self.commits


class TurboPlonkHotPath(PlonkHotPath):
    def __init__(self, curve, log_n, srs, rand_fr, stream=None, sharded=False, group=None):
        super().__init__(curve, log_n, srs, rand_fr, stream, sharded, group, ultra=False)


class UltraPlonkHotPath(PlonkHotPath):
    def __init__(self, curve, log_n, srs, rand_fr, stream=None, sharded=False, group=None):
        super().__init__(curve, log_n, srs, rand_fr, stream, sharded, group, ultra=True)
