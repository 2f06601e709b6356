"""GPU parity for UnivariatePCS::multi_open_rou_proofs (FK23, pcs/src/univariate_kzg/mod.rs:328-395) through the C ABI.
The checker is what the reference's own test asserts (mod.rs:1237-1301): the amortised proofs equal the per-point
openings, here commit((f - f(w^j)) / (X - w^j)) from the C oracle."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


def _per_point(cv, srs_host, coeffs, log_dom, num_points):
    w = R.domain_root(cv, log_dom)
    out, z = [], 1
    for _ in range(num_points):
        q, _ = C.open_witness(cv.curve_id, coeffs, mont_arr(cv.fr, [z])[0])
        out.append(C.msm(cv.curve_id, srs_host[:len(q)], q))
        z = z * w % cv.fr.p
    return np.stack(out)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,log_dom,num_points", [(2, 1, 2), (3, 2, 3), (6, 3, 8), (34, 6, 50), (101, 7, 128), (129, 7, 5)])
def test_multi_open_rou_equals_per_point_open(hip, cv, n, log_dom, num_points):
    beta = mont_arr(cv.fr, [0xF23 + n])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    srs_host = srs.read(0, n)
    coeffs = rand_fr_mont(cv, n, 700 + n)
    proofs, infs = hip.kzg_multi_open_rou(srs, coeffs, log_dom, num_points)
    want = _per_point(cv, srs_host, coeffs, log_dom, num_points)
    assert (proofs == want).all()
    assert not infs.any()
    srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_multi_open_rou_reuses_the_srs_transform(hip, cv):
    """The group DFT of the SRS prefix is kept on the handle after the first call of a shape (csrc/ecntt.hip): a second
    polynomial of the same degree, a shorter one (another shape) and the first one again all match per-point openings."""
    n, log_dom = 40, 6
    srs = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0xFC23])[0], pt_arr(cv, cv.g), n)
    srs_host = srs.read(0, n)
    polys = [rand_fr_mont(cv, n, 31), rand_fr_mont(cv, n, 32), rand_fr_mont(cv, 33, 33), rand_fr_mont(cv, n, 31)]
    for coeffs in polys:
        proofs, infs = hip.kzg_multi_open_rou(srs, coeffs, log_dom, 9)
        assert (proofs == _per_point(cv, srs_host, coeffs, log_dom, 9)).all()
        assert not infs.any()
    srs.free()


def test_multi_open_rou_edge_cases(hip):
    cv = R.BN254
    beta = mont_arr(cv.fr, [77])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), 64)
    # constant and empty polynomials: zero quotients
    for n in (0, 1):
        proofs, infs = hip.kzg_multi_open_rou(srs, rand_fr_mont(cv, n, 3), 3, 8)
        assert infs.all() and not proofs.any()
    # sparse polynomial with zero coefficients in the middle and at the top of the padded range
    coeffs = rand_fr_mont(cv, 20, 9)
    coeffs[5:15] = 0
    proofs, infs = hip.kzg_multi_open_rou(srs, coeffs, 5, 32)
    assert (proofs == _per_point(cv, srs.read(0, 20), coeffs, 5, 32)).all()
    # h does not fit the domain (poly.rs:143-150); more points than the domain has; polynomial longer than the SRS
    with pytest.raises(hip.JfHipError, match="smaller domain"):
        hip.kzg_multi_open_rou(srs, rand_fr_mont(cv, 40, 1), 5, 4)
    with pytest.raises(hip.JfHipError):
        hip.kzg_multi_open_rou(srs, rand_fr_mont(cv, 10, 1), 3, 9)
    with pytest.raises(hip.JfHipError):
        hip.kzg_multi_open_rou(srs, rand_fr_mont(cv, 70, 1), 7, 4)
    srs.free()


def test_multi_open_rou_4096_points(hip):
    """degree 4095 over the 2^12 domain (three group DFTs of 2^13 / 2^12 elements); spot-checked per point"""
    import time

    cv = R.BN254
    n, log_dom = 4096, 12
    beta = mont_arr(cv.fr, [0x4096])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    srs_host = srs.read(0, n)
    coeffs = rand_fr_mont(cv, n, 4096)
    hip.kzg_multi_open_rou(srs, coeffs[:8], 3, 8)
    t0 = time.perf_counter()
    proofs, infs = hip.kzg_multi_open_rou(srs, coeffs, log_dom, 1 << log_dom)
    print("multi_open_rou 4096 proofs: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    w = R.domain_root(cv, log_dom)
    for j in (0, 1, 2047, 4095):
        q, _ = C.open_witness(cv.curve_id, coeffs, mont_arr(cv.fr, [pow(w, j, cv.fr.p)])[0])
        assert (proofs[j] == C.msm(cv.curve_id, srs_host[:len(q)], q)).all(), j
    assert not infs.any()
    srs.free()
