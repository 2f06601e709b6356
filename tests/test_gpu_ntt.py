"""GPU parity for the radix-2 (coset) NTT / iNTT through the C ABI: golden vectors, the C oracle
on seeded inputs, and at BASELINE config-3 size the round trip / linearity / spot-evaluation
properties (the check the reference's check_polynomial does,
relation/src/constraint_system.rs:2262-2268).  Bar: bit-exact."""
import hashlib

import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import from_mont_arr, golden, h2i, mont_arr, rand_fr_mont, ref_kat_cases, seeded_scalars

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ntt_golden(hip, cv):
    for case in golden("ntt")[cv.name]["cases"]:
        log_n, off, in_len, seed = case["log_n"], case["offset"], case["in_len"], case["seed"]
        n = 1 << log_n
        data = np.zeros((1, n, 4), dtype=np.uint64)
        data[0, :in_len] = mont_arr(cv.fr, seeded_scalars(cv, seed, in_len))
        cos = None if off == 1 else mont_arr(cv.fr, [off])[0]
        # rubbish beyond in_len must be ignored (zero-extension is the library's job)
        dirty = data.copy()
        dirty[0, in_len:] = np.uint64(0x1111)
        ev = hip.ntt(cv.curve_id, dirty, log_n, False, cos, in_len=in_len)
        assert hashlib.sha256(ev.tobytes()).hexdigest() == case["fwd_sha256"], case
        for i, v in case["spots"].items():
            assert from_mont_arr(cv.fr, ev[0, int(i)])[0] == h2i(v)
        assert (hip.ntt(cv.curve_id, ev, log_n, True, cos) == data).all()
        evals_in = mont_arr(cv.fr, seeded_scalars(cv, seed ^ 0x5555, n)).reshape(1, n, 4)
        co = hip.ntt(cv.curve_id, evals_in, log_n, True, cos)
        assert hashlib.sha256(co.tobytes()).hexdigest() == case["inv_sha256"], case


def test_intt_reference_known_answers(hip):
    """jfhip_ntt(BLS12_381, inverse) against the literal ark-poly outputs the reference's own test asserts
    (pcs/src/multilinear_kzg/util.rs:428-563): the one byte-level pin on arkworks' FFT convention that exists."""
    cv = R.BLS12_381
    cases, _ = ref_kat_cases()
    for cite, log_n, evals, coeffs in cases:
        data = mont_arr(cv.fr, evals).reshape(1, 1 << log_n, 4)
        got = hip.ntt(cv.curve_id, data, log_n, True, None)
        assert from_mont_arr(cv.fr, got) == coeffs, cite
        assert (hip.ntt(cv.curve_id, got, log_n, False, None) == data).all(), cite
    # the same vectors embedded (zero-extended evaluations are not what interpolate does; instead batch them)
    four = [c for c in cases if c[1] == 2]
    batch = np.stack([mont_arr(cv.fr, c[2]) for c in four])
    got = hip.ntt(cv.curve_id, batch, 2, True, None)
    for row, c in zip(got, four):
        assert from_mont_arr(cv.fr, row) == c[3], c[0]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [0, 1, 5, 8, 9, 11, 13, 16, 17, 18])
def test_ntt_vs_oracle(hip, cv, log_n):
    """every digit plan (1, 2 and 3 passes), batched, all four variants"""
    n = 1 << log_n
    batch = 3
    data = rand_fr_mont(cv, batch * n, 40 + log_n).reshape(batch, n, 4)
    cos = mont_arr(cv.fr, [cv.fr_generator])[0]
    for inverse in (False, True):
        for c in (None, cos):
            got = hip.ntt(cv.curve_id, data, log_n, inverse, c)
            exp = C.ntt(cv.curve_id, data, log_n, inverse, c)
            assert (got == exp).all(), (log_n, inverse, c is not None)
    # zero-extended short input (quotient-shape: n/8 + 3 coefficients)
    if n >= 16:
        k = n // 8 + 3
        short = data.copy()
        short[:, k:] = 0
        got = hip.ntt(cv.curve_id, data, log_n, False, cos, in_len=k)
        assert (got == C.ntt(cv.curve_id, short, log_n, False, cos)).all()


def test_ntt_config3_shape_properties(hip):
    """8 polys x 2^22 BLS12-381 Fr, coset h = 7 (BASELINE config 3): round trip, linearity and
    64 spot evaluations by Horner on the host."""
    cv = R.BLS12_381
    log_n, batch = 22, 8
    n = 1 << log_n
    cos = mont_arr(cv.fr, [cv.fr_generator])[0]
    data = rand_fr_mont(cv, batch * n, 77).reshape(batch, n, 4)
    ev = hip.ntt(cv.curve_id, data, log_n, False, cos)
    assert (hip.ntt(cv.curve_id, ev, log_n, True, cos) == data).all()
    # linearity on two of the polynomials
    s = C.fp_binop(2, "add", data[0], data[1]).reshape(1, n, 4)
    es = hip.ntt(cv.curve_id, s, log_n, False, cos)
    assert (C.fp_binop(2, "add", ev[0], ev[1]).reshape(1, n, 4) == es).all()
    # spot evaluations: ev[p][i] == poly_p(h w^i)
    w = R.domain_root(cv, log_n)
    rng = np.random.default_rng(5)
    for _ in range(64 // batch):
        for p in range(batch):
            i = int(rng.integers(0, n))
            x = cv.fr_generator * pow(w, i, cv.fr.p) % cv.fr.p
            exp = C.poly_eval(cv.curve_id, data[p], mont_arr(cv.fr, [x])[0])
            assert (ev[p, i] == exp).all()


def test_ntt_rejects_oversize(hip):
    """log_n beyond the field's two-adicity (28 for BN254 Fr) is refused before any launch"""
    import ctypes

    lib = hip.load_library()
    rc = lib.jfhip_ntt_device(0, ctypes.c_void_p(256), 29, ctypes.c_size_t(1), 0, None, ctypes.c_size_t(0), None)
    assert rc == -1 and "two-adicity" in hip.last_error()
