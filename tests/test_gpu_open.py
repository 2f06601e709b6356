"""GPU parity for UnivariateKzgPCS::open (pcs/src/univariate_kzg/mod.rs:144-180) through the C ABI:
witness polynomial by synthetic division + its commitment + the evaluation."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import from_mont_arr, golden, gpt, h2i, mont_arr, pt_arr, pt_from, rand_fr_mont

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_open_golden(hip, cv):
    g = golden("open")[cv.name]
    beta = mont_arr(cv.fr, [h2i(g["beta"])])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, gpt(g["g"])), 512)
    for case in g["cases"]:
        coeffs = mont_arr(cv.fr, [h2i(c) for c in case["coeffs"]])
        z = mont_arr(cv.fr, [h2i(case["z"])])[0]
        proof, inf, ev = hip.kzg_open(srs, coeffs, z)
        assert from_mont_arr(cv.fr, ev)[0] == h2i(case["eval"]), case["n"]
        assert pt_from(cv, proof) == gpt(case["proof"]), case["n"]
        assert inf == (case["proof"] is None)
        out, _ = hip.msm(srs, coeffs)
        assert pt_from(cv, out) == gpt(case["commitment"])
    # empty polynomial
    proof, inf, ev = hip.kzg_open(srs, np.zeros((0, 4), dtype=np.uint64), beta)
    assert inf and not ev.any()
    srs.free()


@pytest.mark.parametrize("n", [63, 64, 65, 4096, 4097, 100000])
def test_open_vs_oracle(hip, n):
    """chunk boundaries of the scan (K = 64) and a multi-level size"""
    cv = R.BN254
    beta = mont_arr(cv.fr, [424242])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    srs_host = srs.read(0, n)
    coeffs = rand_fr_mont(cv, n, 900 + n)
    z = rand_fr_mont(cv, 1, 5)[0]
    proof, inf, ev = hip.kzg_open(srs, coeffs, z)
    q, e = C.open_witness(cv.curve_id, coeffs, z)
    assert (ev == e).all()
    assert (proof == C.msm(cv.curve_id, srs_host[:n - 1], q)).all()
    srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_device_polynomial_glue(hip, cv):
    """jfhip_poly_lincomb_device and jfhip_poly_divide_linear_device against the oracle"""
    import torch

    fid = 0 if cv.curve_id == 0 else 2
    lens = [1000, 777, 1003, 1]
    n = 1003
    polys = [rand_fr_mont(cv, l, 60 + i) for i, l in enumerate(lens)]
    coeffs = rand_fr_mont(cv, len(lens), 70)
    dev = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
    out = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    hip.poly_lincomb_device(cv.curve_id, [d.data_ptr() for d in dev], lens, coeffs, out.data_ptr(), n, stream=st)
    st.sync()
    exp = np.zeros((n, 4), dtype=np.uint64)
    for p, c in zip(polys, coeffs):
        term = C.fp_binop(fid, "mul", p, np.tile(c, (len(p), 1)))
        exp[: len(p)] = C.fp_binop(fid, "add", exp[: len(p)], term)
    got = out.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert (got == exp).all()
    # divide the combination by (X - z)
    z = rand_fr_mont(cv, 1, 71)[0]
    res = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    hip.poly_divide_linear_device(cv.curve_id, out.data_ptr(), n, z, res.data_ptr(), stream=st)
    st.sync()
    r = res.cpu().numpy().view(np.uint64).reshape(n, 4)
    q, ev = C.open_witness(cv.curve_id, exp, z)
    assert (r[0] == ev).all() and (r[1:] == q).all()
    st.close()


@pytest.mark.parametrize("lens", [[1], [63, 64, 65], [4096, 4097, 1, 0, 300000]])
def test_poly_eval_batch(hip, lens):
    import torch

    cv = R.BN254
    polys = [rand_fr_mont(cv, max(l, 1), 80 + i)[:l] for i, l in enumerate(lens)]
    dev = [torch.from_numpy(np.ascontiguousarray(p).view(np.int64).reshape(-1)).cuda() if len(p) else
           torch.zeros(4, dtype=torch.int64, device="cuda") for p in polys]
    z = rand_fr_mont(cv, 1, 81)[0]
    out = torch.full((len(lens) * 4,), 7, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    hip.poly_eval_batch_device(0, [d.data_ptr() for d in dev], lens, z, out.data_ptr(), stream=st)
    st.sync()
    got = out.cpu().numpy().view(np.uint64).reshape(len(lens), 4)
    for i, p in enumerate(polys):
        assert (got[i] == C.poly_eval(0, p, z)).all(), lens[i]
    st.close()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_multi_point_open(hip, cv):
    """UnivariatePCS::multi_point_open (pcs/src/univariate_kzg/mod.rs:416-468) through the C ABI against the restated
    long division: several points, k = 1 (= open), repeated points, as many points as coefficients and more (zero
    quotient), the empty polynomial, and the reference's error for no points."""
    beta_i = 0x5EED1234
    beta = mont_arr(cv.fr, [beta_i])[0]
    n = 700
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    coeffs_m = rand_fr_mont(cv, n, 31)
    coeffs = from_mont_arr(cv.fr, coeffs_m)
    zs = from_mont_arr(cv.fr, rand_fr_mont(cv, 6, 32))
    for pts in (zs[:4], zs[:1], [zs[0], zs[1], zs[0], zs[0]], zs[:6]):
        proof, inf, ev = hip.kzg_multi_point_open(srs, coeffs_m, mont_arr(cv.fr, pts))
        w, evals = R.multi_point_open_witness(cv, coeffs, pts)
        assert from_mont_arr(cv.fr, ev) == evals
        assert len(w) == n - len(pts)
        assert pt_from(cv, proof) == R.commit_via_beta(cv, beta_i, cv.g, w) and not inf
    # one point is open()
    p1, _, e1 = hip.kzg_multi_point_open(srs, coeffs_m, mont_arr(cv.fr, zs[:1]))
    p2, _, e2 = hip.kzg_open(srs, coeffs_m, mont_arr(cv.fr, zs[:1])[0])
    assert (p1 == p2).all() and (e1[0] == e2).all()
    # short polynomials: degree below the number of points -> zero witness, identity proof
    for m in (3, 4, 5):
        short = coeffs_m[:m]
        proof, inf, ev = hip.kzg_multi_point_open(srs, short, mont_arr(cv.fr, zs[:4]))
        w, evals = R.multi_point_open_witness(cv, coeffs[:m], zs[:4])
        assert from_mont_arr(cv.fr, ev) == evals
        assert pt_from(cv, proof) == (R.commit_via_beta(cv, beta_i, cv.g, w) if w else None)
        assert inf == (not w)
    proof, inf, ev = hip.kzg_multi_point_open(srs, np.zeros((0, 4), dtype=np.uint64), mont_arr(cv.fr, zs[:2]))
    assert inf and not ev.any() and not proof.any()
    with pytest.raises(hip.JfHipError) as e:
        hip.kzg_multi_point_open(srs, coeffs_m, np.zeros((0, 4), dtype=np.uint64))
    assert e.value.code == -1 and "no point to evaluate and open" in e.value.msg
    srs.free()
