"""ctypes binding of oracle/libjforacle.so (the C CPU restatement).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py as the checker.  Arrays are numpy uint64, Montgomery little-endian limbs.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FIELD_IDS = {"bn254_fr": 0, "bn254_fq": 1, "bls12_381_fr": 2, "bls12_381_fq": 3}


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libjforacle.so")
    srcs = [os.path.join(_HERE, f) for f in ("jf_oracle.c", "fp_tmpl.h", "ec_tmpl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjforacle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libjforacle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        for name in ("jfo_fp_binop", "jfo_fp_unop", "jfo_ec_add", "jfo_ec_mul", "jfo_ec_on_curve", "jfo_srs_gen",
                     "jfo_msm", "jfo_msm_naive", "jfo_poly_eval", "jfo_open_witness", "jfo_ntt", "jfo_domain_root",
                     "jfo_threads", "jfo_field_limbs", "jfo_plonk_quotient_turbo", "jfo_plonk_quotient_ultra", "jfo_plonk_perm_product",
                     "jfo_plonk_lookup_product"):
            getattr(_LIB, name).restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def ql(curve: int) -> int:
    return 4 if curve == 0 else 6


def fp_binop(field_id: int, op: str, a, b):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    nl = lib().jfo_field_limbs(field_id)
    rc = lib().jfo_fp_binop(field_id, {"add": 0, "sub": 1, "mul": 2}[op], _p(a), _p(b), _p(out),
                            ctypes.c_size_t(a.size // nl))
    assert rc == 0
    return out


def fp_unop(field_id: int, op: str, a):
    a = _u64(a)
    out = np.empty_like(a)
    nl = lib().jfo_field_limbs(field_id)
    rc = lib().jfo_fp_unop(field_id, {"neg": 0, "inv": 1, "sqr": 2, "from_mont": 3, "to_mont": 4}[op], _p(a), _p(out),
                           ctypes.c_size_t(a.size // nl))
    assert rc == 0
    return out


def ec_add(curve: int, P, Q):
    P, Q = _u64(P), _u64(Q)
    out = np.zeros(2 * ql(curve), dtype=np.uint64)
    assert lib().jfo_ec_add(curve, _p(P), _p(Q), _p(out)) == 0
    return out


def ec_mul(curve: int, P, k_mont):
    P, k = _u64(P), _u64(k_mont)
    out = np.zeros(2 * ql(curve), dtype=np.uint64)
    assert lib().jfo_ec_mul(curve, _p(P), _p(k), _p(out)) == 0
    return out


def ec_on_curve(curve: int, P) -> bool:
    P = _u64(P)
    return bool(lib().jfo_ec_on_curve(curve, _p(P)))


def srs_gen(curve: int, beta_mont, g_xy, n: int, threads: int = 0):
    beta, g = _u64(beta_mont), _u64(g_xy)
    out = np.zeros((n, 2 * ql(curve)), dtype=np.uint64)
    assert lib().jfo_srs_gen(curve, _p(beta), _p(g), ctypes.c_size_t(n), _p(out), threads or nthreads()) == 0
    return out


def msm(curve: int, bases, scalars_mont, threads: int = 0):
    bases, sc = _u64(bases), _u64(scalars_mont)
    n = sc.size // 4
    assert bases.size == n * 2 * ql(curve)
    out = np.zeros(2 * ql(curve), dtype=np.uint64)
    assert lib().jfo_msm(curve, _p(bases), _p(sc), ctypes.c_size_t(n), _p(out), threads or nthreads()) == 0
    return out


def msm_naive(curve: int, bases, scalars_mont):
    bases, sc = _u64(bases), _u64(scalars_mont)
    n = sc.size // 4
    out = np.zeros(2 * ql(curve), dtype=np.uint64)
    assert lib().jfo_msm_naive(curve, _p(bases), _p(sc), ctypes.c_size_t(n), _p(out)) == 0
    return out


def poly_eval(curve: int, coeffs, z):
    c, z = _u64(coeffs), _u64(z)
    out = np.zeros(4, dtype=np.uint64)
    assert lib().jfo_poly_eval(curve, _p(c), ctypes.c_size_t(c.size // 4), _p(z), _p(out)) == 0
    return out


def open_witness(curve: int, coeffs, z):
    c, z = _u64(coeffs), _u64(z)
    n = c.size // 4
    q = np.zeros((max(n - 1, 0), 4), dtype=np.uint64)
    ev = np.zeros(4, dtype=np.uint64)
    assert lib().jfo_open_witness(curve, _p(c), ctypes.c_size_t(n), _p(z), _p(q), _p(ev)) == 0
    return q, ev


def ntt(curve: int, data, log_n: int, inverse: bool = False, coset_mont=None, threads: int = 0):
    """data: (batch, 2^log_n, 4) uint64; returns a transformed copy."""
    d = np.array(data, dtype=np.uint64, order="C", copy=True)
    n = 1 << log_n
    batch = d.size // (4 * n)
    assert batch * 4 * n == d.size
    cp = None
    if coset_mont is not None:
        cm = _u64(coset_mont)
        cp = _p(cm)
    assert lib().jfo_ntt(curve, _p(d), log_n, ctypes.c_size_t(batch), int(bool(inverse)), cp,
                         threads or nthreads()) == 0
    return d


def plonk_quotient_turbo(curve: int, log_n: int, log_m: int, sel, sig, wires, zperm, pi, k, alpha, beta, gamma,
                         alpha_base, out=None, threads: int = 0):
    """arrays [poly][m][4] uint64 Montgomery; returns out (+= alpha_base * t)."""
    m = 1 << log_m
    sel, sig, wires, zperm, pi = (_u64(x) for x in (sel, sig, wires, zperm, pi))
    assert sel.size == 13 * m * 4 and sig.size == 5 * m * 4 and wires.size == 5 * m * 4
    k, alpha, beta, gamma, alpha_base = (_u64(x) for x in (k, alpha, beta, gamma, alpha_base))
    o = np.zeros((m, 4), dtype=np.uint64) if out is None else np.array(out, dtype=np.uint64, copy=True)
    assert lib().jfo_plonk_quotient_turbo(curve, log_n, log_m, 5, _p(sel), _p(sig), _p(wires), _p(zperm), _p(pi), _p(k),
                                          _p(alpha), _p(beta), _p(gamma), _p(alpha_base), _p(o),
                                          threads or nthreads()) == 0
    return o


def plonk_quotient_ultra(curve: int, log_n: int, log_m: int, sel, sig, wires, zperm, pi, lookup, k, alpha, beta, gamma,
                         tau, alpha_base, out=None, threads: int = 0):
    """UltraPlonk: sel [14][m][4] (q_lookup last), sig/wires [6][m][4], lookup [7][m][4] in the order
    range_table, key_table, table_dom_sep, q_dom_sep, h1, h2, prod_lookup; returns out (+= alpha_base * t)."""
    m = 1 << log_m
    sel, sig, wires, zperm, pi, lookup = (_u64(x) for x in (sel, sig, wires, zperm, pi, lookup))
    assert sel.size == 14 * m * 4 and sig.size == 6 * m * 4 and wires.size == 6 * m * 4 and lookup.size == 7 * m * 4
    k, alpha, beta, gamma, tau, alpha_base = (_u64(x) for x in (k, alpha, beta, gamma, tau, alpha_base))
    o = np.zeros((m, 4), dtype=np.uint64) if out is None else np.array(out, dtype=np.uint64, copy=True)
    assert lib().jfo_plonk_quotient_ultra(curve, log_n, log_m, _p(sel), _p(sig), _p(wires), _p(zperm), _p(pi),
                                          _p(lookup), _p(k), _p(alpha), _p(beta), _p(gamma), _p(tau), _p(alpha_base),
                                          _p(o), threads or nthreads()) == 0
    return o


def plonk_perm_product(curve: int, log_n: int, wires, idp, sig, beta, gamma):
    """wires / idp / sig: [nw][n][4]; returns the n grand-product values on H"""
    wires, idp, sig, beta, gamma = (_u64(x) for x in (wires, idp, sig, beta, gamma))
    n = 1 << log_n
    nw = wires.size // (4 * n)
    out = np.zeros((n, 4), dtype=np.uint64)
    assert lib().jfo_plonk_perm_product(curve, log_n, nw, _p(wires), _p(idp), _p(sig), _p(beta), _p(gamma), _p(out)) == 0
    return out


def plonk_lookup_product(curve: int, log_n: int, f, t, s, beta, gamma):
    f, t, s, beta, gamma = (_u64(x) for x in (f, t, s, beta, gamma))
    n = 1 << log_n
    assert s.size == (2 * n - 1) * 4
    out = np.zeros((n, 4), dtype=np.uint64)
    assert lib().jfo_plonk_lookup_product(curve, log_n, _p(f), _p(t), _p(s), _p(beta), _p(gamma), _p(out)) == 0
    return out


def domain_root(curve: int, log_n: int):
    out = np.zeros(4, dtype=np.uint64)
    assert lib().jfo_domain_root(curve, log_n, _p(out)) == 0
    return out


def nthreads() -> int:
    # the GPU box reports the whole machine's CPUs; its share for one GPU is 16
    env = os.environ.get("OMP_NUM_THREADS")
    if env:
        return max(1, int(env))
    return max(1, min(16, os.cpu_count() or 1))
