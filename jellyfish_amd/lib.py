"""ctypes binding of libjfhip.so -- one Python function per C-ABI entry point of include/jfhip.h.

Arrays are numpy ``uint64`` (Montgomery little-endian limbs, exactly the bytes arkworks holds
in ``Fp.0.0``); device pointers are plain integers (e.g. ``torch.Tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

BN254 = 0
BLS12_381 = 1
CURVE_IDS = {"bn254": BN254, "bls12_381": BLS12_381}
FQ_LIMBS64 = {BN254: 4, BLS12_381: 6}

_LIB = None


class JfHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"jfhip error {code}: {msg}")
        self.code = code
        self.msg = msg


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libjfhip.so")


# every symbol include/jfhip.h declares
SYMBOLS = [
    "jfhip_init", "jfhip_shutdown", "jfhip_cache_clear", "jfhip_stream_release_workspace", "jfhip_last_error", "jfhip_device_count", "jfhip_stream_create", "jfhip_stream_destroy",
    "jfhip_stream_sync", "jfhip_stream_raw", "jfhip_stream_wrap", "jfhip_srs_load", "jfhip_srs_load_device",
    "jfhip_srs_load_serialized", "jfhip_srs_lagrange", "jfhip_msm_small",
    "jfhip_srs_gen_for_testing", "jfhip_srs_free", "jfhip_srs_release_caches", "jfhip_srs_len", "jfhip_srs_window_bits", "jfhip_srs_read",
    "jfhip_msm", "jfhip_msm_batch", "jfhip_msm_batch_device", "jfhip_msm_device", "jfhip_msm_partial_device", "jfhip_msm_batch_partial_device", "jfhip_g1_sum_partials",
    "jfhip_kzg_open", "jfhip_kzg_multi_point_open", "jfhip_kzg_multi_open_rou", "jfhip_mle_open", "jfhip_mle_fold_device", "jfhip_ntt", "jfhip_ntt_device", "jfhip_plonk_quotient_turbo_device", "jfhip_plonk_quotient_ultra_device", "jfhip_plonk_quotient_coset_device", "jfhip_plonk_quotient_masked_device", "jfhip_plonk_perm_product_device", "jfhip_plonk_lookup_product_device", "jfhip_plonk_lookup_merge_device", "jfhip_plonk_lookup_sorted_device", "jfhip_witness_gather_device", "jfhip_poly_divide_linear_device", "jfhip_poly_divide_linear_noalloc_device",
    "jfhip_poly_lincomb_device", "jfhip_poly_eval_batch_device", "jfhip_field_op", "jfhip_g1_op", "jfhip_probe",
    "jfhip_marker_push", "jfhip_marker_pop", "jfhip_set_profiling", "jfhip_last_phase_ms", "jfhip_phase_name", "jfhip_keccak_f1600",
]


def load_library():
    """Load libjfhip.so; raises if it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch ships its own copy of the HIP runtime; if libjfhip.so pulled in /opt/rocm's first, a later
    # `import torch` would bring up a second runtime in the process and find no devices.  Python users
    # of this binding hold device memory in torch tensors anyway, so load torch's runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `make -C jellyfish_amd/csrc` (or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    for s in SYMBOLS:
        getattr(lib, s).restype = ctypes.c_int32
    lib.jfhip_srs_len.restype = ctypes.c_size_t
    lib.jfhip_stream_raw.restype = ctypes.c_void_p
    lib.jfhip_phase_name.restype = ctypes.c_char_p
    _LIB = lib
    return lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(600)
    load_library().jfhip_last_error(buf, ctypes.c_size_t(600))
    return buf.value.decode(errors="replace")


def _check(rc: int):
    if rc != 0:
        raise JfHipError(rc, last_error())


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _vp(x):
    return ctypes.c_void_p(int(x)) if x is not None else None


def init(device: int = -1):
    _check(load_library().jfhip_init(device))


def shutdown():
    _check(load_library().jfhip_shutdown())


def cache_clear():
    _check(load_library().jfhip_cache_clear())


def device_count() -> int:
    return load_library().jfhip_device_count()


class Stream:
    """jfhip_stream: a HIP stream plus its workspace (warmup_new_stream, pcs/src/lib.rs:430-435)."""

    def __init__(self, wrap_raw: int | None = None):
        h = ctypes.c_void_p()
        if wrap_raw is None:
            _check(load_library().jfhip_stream_create(ctypes.byref(h)))
        else:
            _check(load_library().jfhip_stream_wrap(_vp(wrap_raw), ctypes.byref(h)))
        self.h = h

    def sync(self):
        _check(load_library().jfhip_stream_sync(self.h))

    def release_workspace(self):
        _check(load_library().jfhip_stream_release_workspace(self.h))

    @property
    def raw(self) -> int:
        return int(load_library().jfhip_stream_raw(self.h) or 0)

    def close(self):
        if self.h:
            load_library().jfhip_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _sh(stream):
    return stream.h if stream is not None else None


class Srs:
    """Device-resident prover parameters (powers_of_g), GPUCommittable::load_prover_param_to_gpu."""

    def __init__(self, handle, curve: int):
        self.h = handle
        self.curve = curve

    @classmethod
    def load(cls, curve: int, points, window_bits: int = 0, stride_bytes: int | None = None) -> "Srs":
        pts = np.ascontiguousarray(points)
        pt_bytes = 16 * FQ_LIMBS64[curve]
        if stride_bytes is None:
            stride_bytes = pt_bytes
        n = pts.nbytes // stride_bytes
        h = ctypes.c_void_p()
        _check(load_library().jfhip_srs_load(curve, _p(pts), ctypes.c_size_t(n), ctypes.c_size_t(stride_bytes),
                                             window_bits, ctypes.byref(h)))
        return cls(h, curve)

    @classmethod
    def load_serialized(cls, curve: int, data: bytes, n: int, compressed: bool = True, validate: bool = True,
                        window_bits: int = 0) -> "Srs":
        """n ark-serialize G1Affine records (the Vec length prefix already skipped), decoded on the device"""
        buf = np.frombuffer(data, dtype=np.uint8)
        h = ctypes.c_void_p()
        _check(load_library().jfhip_srs_load_serialized(curve, _p(buf), ctypes.c_size_t(buf.size), ctypes.c_size_t(n),
                                                        int(compressed), int(validate), window_bits, ctypes.byref(h)))
        return cls(h, curve)

    @classmethod
    def load_device(cls, curve: int, d_points: int, n: int, window_bits: int = 0) -> "Srs":
        h = ctypes.c_void_p()
        _check(load_library().jfhip_srs_load_device(curve, _vp(d_points), ctypes.c_size_t(n), window_bits,
                                                    ctypes.byref(h)))
        return cls(h, curve)

    @classmethod
    def gen_for_testing(cls, curve: int, beta_mont, g_xy, n: int, window_bits: int = 0) -> "Srs":
        beta, g = _u64(beta_mont), _u64(g_xy)
        h = ctypes.c_void_p()
        _check(load_library().jfhip_srs_gen_for_testing(curve, _p(beta), _p(g), ctypes.c_size_t(n), window_bits,
                                                        ctypes.byref(h)))
        return cls(h, curve)

    def lagrange(self, log_n: int, blinders: int = 3, window_bits: int = 0) -> "Srs":
        """a NEW handle: the Lagrange-basis commit key of the radix-2 domain of size 2^log_n, [L_i(beta)]g, followed by
        `blinders` points [beta^(n+k)]g - [beta^k]g (jfhip_srs_lagrange): an MSM over (evaluations, blinders) on it is the
        commitment of the blinded polynomial with those evaluations"""
        h = ctypes.c_void_p()
        _check(load_library().jfhip_srs_lagrange(self.h, ctypes.c_uint32(log_n), ctypes.c_uint32(blinders), window_bits,
                                                 ctypes.byref(h)))
        return Srs(h, self.curve)

    def __len__(self) -> int:
        return int(load_library().jfhip_srs_len(self.h))

    @property
    def window_bits(self) -> int:
        return int(load_library().jfhip_srs_window_bits(self.h))

    def read(self, offset: int, n: int) -> np.ndarray:
        out = np.zeros((n, 2 * FQ_LIMBS64[self.curve]), dtype=np.uint64)
        _check(load_library().jfhip_srs_read(self.h, ctypes.c_size_t(offset), ctypes.c_size_t(n), _p(out)))
        return out

    def release_caches(self):
        """give back the FK23 transforms this handle keeps (rebuilt on demand)"""
        _check(load_library().jfhip_srs_release_caches(self.h))

    def free(self):
        if self.h:
            load_library().jfhip_srs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm(srs: Srs, scalars_mont, offset: int = 0, stream: Stream | None = None):
    """-> (packed affine point as uint64[2*fq_limbs], is_identity)"""
    sc = _u64(scalars_mont)
    n = sc.size // 4
    out = np.zeros(2 * FQ_LIMBS64[srs.curve], dtype=np.uint64)
    inf = ctypes.c_uint8(0)
    _check(load_library().jfhip_msm(srs.h, ctypes.c_size_t(offset), _p(sc) if n else None, ctypes.c_size_t(n), _p(out),
                                    ctypes.byref(inf), _sh(stream)))
    return out, bool(inf.value)


def msm_small(curve: int, points_xy, scalars_mont, stream: Stream | None = None):
    """variable-base MSM on host points (verifier side) -> (packed affine point, is_identity)"""
    pts, sc = _u64(points_xy), _u64(scalars_mont)
    n = sc.size // 4
    out = np.zeros(2 * FQ_LIMBS64[curve], dtype=np.uint64)
    inf = ctypes.c_uint8(0)
    _check(load_library().jfhip_msm_small(curve, _p(pts) if n else None, _p(sc) if n else None, ctypes.c_size_t(n),
                                          _p(out), ctypes.byref(inf), _sh(stream)))
    return out, bool(inf.value)


def msm_batch(srs: Srs, polys, stream: Stream | None = None):
    arrs = [_u64(p) for p in polys]
    batch = len(arrs)
    ql = 2 * FQ_LIMBS64[srs.curve]
    out = np.zeros((batch, ql), dtype=np.uint64)
    inf = np.zeros(batch, dtype=np.uint8)
    ptrs = (ctypes.c_void_p * max(batch, 1))(*[a.ctypes.data if a.size else None for a in arrs])
    lens = (ctypes.c_size_t * max(batch, 1))(*[a.size // 4 for a in arrs])
    _check(load_library().jfhip_msm_batch(srs.h, ptrs, lens, ctypes.c_size_t(batch), _p(out), _p(inf), _sh(stream)))
    return out, inf.astype(bool)


def msm_batch_device(srs: Srs, d_scalars, lens, d_out: int, stream: Stream | None = None):
    """batch of commits on device scalars; d_out: len(lens) records of ceil16(2*fq_bytes + 4) bytes"""
    batch = len(lens)
    ptrs = (ctypes.c_void_p * max(batch, 1))(*[int(p) for p in d_scalars])
    ls = (ctypes.c_size_t * max(batch, 1))(*[int(x) for x in lens])
    _check(load_library().jfhip_msm_batch_device(srs.h, ptrs, ls, ctypes.c_size_t(batch), _vp(d_out), _sh(stream)))


def msm_device(srs: Srs, d_scalars: int, n: int, d_out: int, offset: int = 0, stream: Stream | None = None):
    _check(load_library().jfhip_msm_device(srs.h, ctypes.c_size_t(offset), _vp(d_scalars), ctypes.c_size_t(n),
                                           _vp(d_out), _sh(stream)))


def msm_partial_device(srs: Srs, d_scalars: int, n: int, d_out_xyzz: int, offset: int = 0,
                       stream: Stream | None = None):
    _check(load_library().jfhip_msm_partial_device(srs.h, ctypes.c_size_t(offset), _vp(d_scalars), ctypes.c_size_t(n),
                                                   _vp(d_out_xyzz), _sh(stream)))


def msm_batch_partial_device(srs: Srs, d_scalars, lens, d_out_xyzz: int, offsets=None, stream: Stream | None = None):
    """un-normalised partial sums of a batch_commit round (one shared pipeline): len(lens) XYZZ elements, back to back"""
    batch = len(lens)
    ptrs = (ctypes.c_void_p * max(batch, 1))(*[int(p) for p in d_scalars])
    ls = (ctypes.c_size_t * max(batch, 1))(*[int(x) for x in lens])
    offs = None if offsets is None else (ctypes.c_size_t * max(batch, 1))(*[int(x) for x in offsets])
    _check(load_library().jfhip_msm_batch_partial_device(srs.h, offs, ptrs, ls, ctypes.c_size_t(batch), _vp(d_out_xyzz),
                                                         _sh(stream)))


def g1_sum_partials(curve: int, d_partials: int, count: int, d_out: int, stream: Stream | None = None):
    _check(load_library().jfhip_g1_sum_partials(curve, _vp(d_partials), ctypes.c_size_t(count), _vp(d_out),
                                                _sh(stream)))


def kzg_open(srs: Srs, coeffs_mont, point_mont, stream: Stream | None = None):
    """-> (proof packed affine, is_identity, eval limbs)"""
    c, z = _u64(coeffs_mont), _u64(point_mont)
    n = c.size // 4
    out = np.zeros(2 * FQ_LIMBS64[srs.curve], dtype=np.uint64)
    ev = np.zeros(4, dtype=np.uint64)
    inf = ctypes.c_uint8(0)
    _check(load_library().jfhip_kzg_open(srs.h, _p(c) if n else None, ctypes.c_size_t(n), _p(z), _p(out),
                                         ctypes.byref(inf), _p(ev), _sh(stream)))
    return out, bool(inf.value), ev


def kzg_multi_point_open(srs: Srs, coeffs_mont, points_mont, stream: Stream | None = None):
    """UnivariatePCS::multi_point_open -> (proof packed affine, is_identity, evals (num_points, 4))"""
    c, z = _u64(coeffs_mont), _u64(points_mont)
    n, k = c.size // 4, z.size // 4
    out = np.zeros(2 * FQ_LIMBS64[srs.curve], dtype=np.uint64)
    ev = np.zeros((max(k, 1), 4), dtype=np.uint64)
    inf = ctypes.c_uint8(0)
    _check(load_library().jfhip_kzg_multi_point_open(srs.h, _p(c) if n else None, ctypes.c_size_t(n),
                                                     _p(z) if k else None, ctypes.c_size_t(k), _p(out), ctypes.byref(inf),
                                                     _p(ev), _sh(stream)))
    return out, bool(inf.value), ev[:k]


def plonk_lookup_sorted_device(d_table: int, n_table: int, d_lookups: int, n_lookups: int, d_sorted: int,
                               stream: Stream | None = None):
    _check(load_library().jfhip_plonk_lookup_sorted_device(_vp(d_table), ctypes.c_size_t(n_table), _vp(d_lookups),
                                                           ctypes.c_size_t(n_lookups), _vp(d_sorted), _sh(stream)))


def kzg_multi_open_rou(srs: Srs, coeffs_mont, log_domain: int, num_points: int, stream: Stream | None = None):
    """UnivariatePCS::multi_open_rou_proofs -> (proofs (num_points, 2 * fq limbs), identity flags)"""
    c = _u64(coeffs_mont)
    n = c.size // 4
    out = np.zeros((num_points, 2 * FQ_LIMBS64[srs.curve]), dtype=np.uint64)
    inf = np.zeros(num_points, dtype=np.uint8)
    _check(load_library().jfhip_kzg_multi_open_rou(srs.h, _p(c) if n else None, ctypes.c_size_t(n), log_domain,
                                                   ctypes.c_size_t(num_points), _p(out) if num_points else None,
                                                   _p(inf) if num_points else None, _sh(stream)))
    return out, inf.astype(bool)


def mle_open(levels, evals_mont, point_mont, stream: Stream | None = None):
    """MultilinearKzgPCS::open (open_internal): levels[i] = Srs of powers_of_g[ignored + i].evals.
    -> (proofs (nv, 2 * fq limbs) packed affine, identity flags (nv,), eval limbs)"""
    ev, pt = _u64(evals_mont), _u64(point_mont)
    nv = len(levels)
    if ev.size != 4 << nv or pt.size != 4 * nv:
        raise ValueError("evaluation table must hold 2^nv values and the point nv values")
    curve = levels[0].curve if nv else 0
    out = np.zeros((nv, 2 * FQ_LIMBS64[curve]), dtype=np.uint64)
    inf = np.zeros(nv, dtype=np.uint8)
    val = np.zeros(4, dtype=np.uint64)
    hs = (ctypes.c_void_p * max(nv, 1))(*[lv.h for lv in levels])
    _check(load_library().jfhip_mle_open(hs, ctypes.c_size_t(nv), _p(ev), _p(pt) if nv else None, _p(out) if nv else None,
                                         _p(inf) if nv else None, _p(val), _sh(stream)))
    return out, inf.astype(bool), val


def mle_fold_device(curve: int, d_f: int, half: int, point_mont, d_q: int, d_r: int, stream: Stream | None = None):
    z = _u64(point_mont)
    _check(load_library().jfhip_mle_fold_device(curve, _vp(d_f), ctypes.c_size_t(half), _p(z), _vp(d_q), _vp(d_r),
                                                _sh(stream)))


def ntt(curve: int, data, log_n: int, inverse: bool = False, coset_mont=None, in_len: int = 0,
        stream: Stream | None = None) -> np.ndarray:
    """data: (batch, 2^log_n, 4) uint64 -> transformed copy (natural order in and out)."""
    d = np.array(data, dtype=np.uint64, order="C", copy=True)
    n = 1 << log_n
    batch = d.size // (4 * n)
    if batch * 4 * n != d.size:
        raise ValueError("data size is not a multiple of the domain size")
    cs = _u64(coset_mont) if coset_mont is not None else None
    _check(load_library().jfhip_ntt(curve, _p(d), log_n, ctypes.c_size_t(batch), int(bool(inverse)),
                                    _p(cs) if cs is not None else None, ctypes.c_size_t(in_len), _sh(stream)))
    return d


def ntt_device(curve: int, d_data: int, log_n: int, batch: int, inverse: bool = False, coset_mont=None,
               in_len: int = 0, stream: Stream | None = None):
    cs = _u64(coset_mont) if coset_mont is not None else None
    _check(load_library().jfhip_ntt_device(curve, _vp(d_data), log_n, ctypes.c_size_t(batch), int(bool(inverse)),
                                           _p(cs) if cs is not None else None, ctypes.c_size_t(in_len), _sh(stream)))


def plonk_quotient_turbo_device(curve: int, log_n: int, log_m: int, d_selectors: int, d_sigmas: int, d_wires: int,
                                d_prod_perm: int, d_pub_input: int, k_mont, challenges_mont, d_quot: int,
                                stream: Stream | None = None):
    """TurboPlonk quotient evaluations on device arrays (see include/jfhip.h)."""
    k, ch = _u64(k_mont), _u64(challenges_mont)
    assert k.size == 20 and ch.size == 16
    _check(load_library().jfhip_plonk_quotient_turbo_device(curve, log_n, log_m, _vp(d_selectors), _vp(d_sigmas),
                                                            _vp(d_wires), _vp(d_prod_perm), _vp(d_pub_input), _p(k),
                                                            _p(ch), _vp(d_quot), _sh(stream)))


def plonk_quotient_ultra_device(curve: int, log_n: int, log_m: int, d_selectors: int, d_sigmas: int, d_wires: int,
                                d_prod_perm: int, d_pub_input: int, d_plookup: int, k_mont, challenges_mont, d_quot: int,
                                stream: Stream | None = None):
    """UltraPlonk (Plookup) quotient evaluations on device arrays (see include/jfhip.h)."""
    k, ch = _u64(k_mont), _u64(challenges_mont)
    assert k.size == 24 and ch.size == 20
    _check(load_library().jfhip_plonk_quotient_ultra_device(curve, log_n, log_m, _vp(d_selectors), _vp(d_sigmas),
                                                            _vp(d_wires), _vp(d_prod_perm), _vp(d_pub_input),
                                                            _vp(d_plookup), _p(k), _p(ch), _vp(d_quot), _sh(stream)))


def plonk_quotient_coset_device(curve: int, num_wire_types: int, log_n: int, log_m: int, coset_offset_mont,
                                d_selectors: int, d_sigmas: int, d_wires: int, d_prod_perm: int, d_pub_input: int,
                                d_plookup: int, k_mont, challenges_mont, d_quot: int, stream: Stream | None = None):
    """quotient evaluations over one slice (size 2^log_m, offset coset_offset_mont) of the quotient coset"""
    off, k, ch = _u64(coset_offset_mont), _u64(k_mont), _u64(challenges_mont)
    assert off.size == 4 and k.size == 4 * num_wire_types and ch.size == 4 * (num_wire_types - 1)
    _check(load_library().jfhip_plonk_quotient_coset_device(curve, num_wire_types, log_n, log_m, _p(off),
                                                            _vp(d_selectors), _vp(d_sigmas), _vp(d_wires),
                                                            _vp(d_prod_perm), _vp(d_pub_input),
                                                            _vp(d_plookup) if d_plookup else None, _p(k), _p(ch),
                                                            _vp(d_quot), _sh(stream)))


def plonk_quotient_masked_device(curve: int, num_wire_types: int, log_n: int, log_m: int, coset_offset_mont,
                                 d_selectors: int, d_sigmas: int, d_wires: int, d_prod_perm: int, d_pub_input: int,
                                 d_plookup: int, k_mont, challenges_mont, zero_selector_mask: int, d_quot: int,
                                 stream: Stream | None = None):
    """plonk_quotient_coset_device that skips identically-zero selectors (mask bit j) and a NULL public input"""
    off, k, ch = _u64(coset_offset_mont), _u64(k_mont), _u64(challenges_mont)
    assert off.size == 4 and k.size == 4 * num_wire_types and ch.size == 4 * (num_wire_types - 1)
    _check(load_library().jfhip_plonk_quotient_masked_device(curve, num_wire_types, log_n, log_m, _p(off),
                                                             _vp(d_selectors) if d_selectors else None, _vp(d_sigmas),
                                                             _vp(d_wires), _vp(d_prod_perm),
                                                             _vp(d_pub_input) if d_pub_input else None,
                                                             _vp(d_plookup) if d_plookup else None, _p(k), _p(ch),
                                                             ctypes.c_uint32(zero_selector_mask), _vp(d_quot),
                                                             _sh(stream)))


def plonk_perm_product_device(curve: int, log_n: int, num_wire_types: int, d_wire_vals: int, d_id_perm: int,
                              d_sigma_vals: int, beta_mont, gamma_mont, d_out: int, stream: Stream | None = None):
    """permutation grand product values on H (see include/jfhip.h)"""
    b, g = _u64(beta_mont), _u64(gamma_mont)
    _check(load_library().jfhip_plonk_perm_product_device(curve, log_n, num_wire_types, _vp(d_wire_vals), _vp(d_id_perm),
                                                          _vp(d_sigma_vals), _p(b), _p(g), _vp(d_out), _sh(stream)))


def plonk_lookup_product_device(curve: int, log_n: int, d_lookup_wire: int, d_table: int, d_sorted: int, beta_mont,
                                gamma_mont, d_out: int, stream: Stream | None = None):
    """Plookup grand product values on H (see include/jfhip.h)"""
    b, g = _u64(beta_mont), _u64(gamma_mont)
    _check(load_library().jfhip_plonk_lookup_product_device(curve, log_n, _vp(d_lookup_wire), _vp(d_table), _vp(d_sorted),
                                                            _p(b), _p(g), _vp(d_out), _sh(stream)))


def plonk_lookup_merge_device(curve: int, log_n: int, d_wire_vals: int, d_table_vals: int, tau_mont, d_merged_table: int,
                              d_merged_lookup: int, stream: Stream | None = None):
    """Plookup merged table / merged lookup witness values on H (see include/jfhip.h)"""
    t = _u64(tau_mont)
    _check(load_library().jfhip_plonk_lookup_merge_device(curve, log_n, _vp(d_wire_vals), _vp(d_table_vals), _p(t),
                                                          _vp(d_merged_table), _vp(d_merged_lookup), _sh(stream)))


def poly_divide_linear_device(curve: int, d_coeffs: int, n: int, point_mont, d_out: int, stream: Stream | None = None):
    """d_out[0] = f(z), d_out[1..n) = (f - f(z)) / (X - z)   (device arrays)."""
    z = _u64(point_mont)
    _check(load_library().jfhip_poly_divide_linear_device(curve, _vp(d_coeffs), ctypes.c_size_t(n), _p(z), _vp(d_out),
                                                          _sh(stream)))


def poly_divide_linear_noalloc_device(curve: int, d_coeffs: int, n: int, point_mont, d_out: int, stream: Stream):
    """poly_divide_linear_device on the library's static workspace: no allocation (n <= 2^23, stream required)."""
    z = _u64(point_mont)
    _check(load_library().jfhip_poly_divide_linear_noalloc_device(curve, _vp(d_coeffs), ctypes.c_size_t(n), _p(z),
                                                                  _vp(d_out), _sh(stream)))


def poly_eval_batch_device(curve: int, d_polys, lens, point_mont, d_out: int, stream: Stream | None = None):
    """d_out[k] = d_polys[k](z)   (device arrays; d_out: len(lens) field elements)"""
    count = len(d_polys)
    ptrs = (ctypes.c_void_p * count)(*[int(p) for p in d_polys])
    ls = (ctypes.c_size_t * count)(*[int(x) for x in lens])
    z = _u64(point_mont)
    _check(load_library().jfhip_poly_eval_batch_device(curve, ptrs, ls, ctypes.c_size_t(count), _p(z), _vp(d_out),
                                                       _sh(stream)))


def witness_gather_device(d_witness: int, num_vars: int, d_wire_variables: int, count: int, d_out: int,
                          check: bool = False, stream: Stream | None = None):
    """d_out[i] = d_witness[d_wire_variables[i]] (32-byte rows; uint32 indices on the device)"""
    _check(load_library().jfhip_witness_gather_device(_vp(d_witness), ctypes.c_size_t(num_vars), _vp(d_wire_variables),
                                                      ctypes.c_size_t(count), _vp(d_out), int(bool(check)), _sh(stream)))


def poly_lincomb_device(curve: int, d_polys, lens, coeffs_mont, d_out: int, n: int, stream: Stream | None = None):
    """d_out[i] = sum_k coeffs[k] * d_polys[k][i]   (device arrays, ragged lengths)."""
    count = len(d_polys)
    ptrs = (ctypes.c_void_p * count)(*[int(p) for p in d_polys])
    ls = (ctypes.c_size_t * count)(*[int(x) for x in lens])
    c = _u64(coeffs_mont)
    assert c.size == 4 * count
    _check(load_library().jfhip_poly_lincomb_device(curve, ptrs, ls, ctypes.c_size_t(count), _p(c), _vp(d_out),
                                                    ctypes.c_size_t(n), _sh(stream)))


FIELD_OPS = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "inv": 4, "from_mont": 5, "to_mont": 6,
             # unsaturated 29-bit-limb arithmetic of csrc/fp29.hpp through its conversions (test surface)
             "mul29": 7, "sqr29": 8, "sub29": 9, "chain29": 10, "dot29": 11,
             # the two Euclid-type inversions of csrc/fp.hpp (division steps in batches of 30; binary)
             "inv_safegcd": 12, "inv_binary": 13}


def field_op(field_id: int, op: str, a, b=None) -> np.ndarray:
    a = _u64(a)
    nl = 6 if field_id == 3 else 4
    n = a.size // nl
    out = np.zeros_like(a)
    bb = _u64(b) if b is not None else None
    _check(load_library().jfhip_field_op(field_id, FIELD_OPS[op], _p(a), _p(bb) if bb is not None else None, _p(out),
                                         ctypes.c_size_t(n)))
    return out


def g1_add(curve: int, P, Q, limbs29: bool = False) -> np.ndarray:
    """P + Q on packed affine points; limbs29: through the MSM tail's 29-bit-limb group law (same result)."""
    P, Q = _u64(P), _u64(Q)
    out = np.zeros(2 * FQ_LIMBS64[curve], dtype=np.uint64)
    _check(load_library().jfhip_g1_op(curve, 2 if limbs29 else 0, _p(P), _p(Q), _p(out)))
    return out


def g1_mul(curve: int, P, k_mont, limbs29=False) -> np.ndarray:
    """k * P; limbs29: True = double-and-add on the 29-bit-limb group law, "naf" = the group-element DFT's non-adjacent-
    form multiplication on the same arithmetic (same results)"""
    P, k = _u64(P), _u64(k_mont)
    out = np.zeros(2 * FQ_LIMBS64[curve], dtype=np.uint64)
    op = 4 if limbs29 == "naf" else (3 if limbs29 else 1)
    _check(load_library().jfhip_g1_op(curve, op, _p(P), _p(k), _p(out)))
    return out


def probe(kind: int, arg: int = 0) -> float:
    r = ctypes.c_double(0)
    _check(load_library().jfhip_probe(kind, arg, ctypes.byref(r)))
    return r.value


class marker:
    """roctx range around a block (jfhip_marker_push / pop): a no-op unless JFHIP_MARKERS=1 and a roctx library exists"""

    def __init__(self, name: str):
        self.name = name.encode()

    def __enter__(self):
        load_library().jfhip_marker_push(self.name)
        return self

    def __exit__(self, *exc):
        load_library().jfhip_marker_pop()
        return False


def set_profiling(enabled: bool):
    load_library().jfhip_set_profiling(int(enabled))


def last_phase_ms(stream: Stream | None = None, kind: int = 0) -> dict:
    arr = (ctypes.c_double * 16)()
    n = load_library().jfhip_last_phase_ms(_sh(stream), arr, 16)
    out = {}
    for i in range(n):
        name = load_library().jfhip_phase_name(kind, i).decode()
        out[name or f"phase{i}"] = arr[i]
    return out


def keccak_f1600(state: bytearray) -> None:
    """in-place Keccak-f[1600] on a 200-byte state (host code in libjfhip)"""
    buf = (ctypes.c_uint64 * 25).from_buffer(state)
    _check(load_library().jfhip_keccak_f1600(buf))
