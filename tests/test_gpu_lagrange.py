"""GPU parity for the Lagrange-basis commit key (jfhip_srs_lagrange; no counterpart in the reference, which commits
coefficient vectors only: pcs/src/univariate_kzg/mod.rs:90-125).  What it must satisfy is an identity of the reference's
own objects: for evaluations e_i on the radix-2 domain and blinders b_k, the MSM over (e, b) on the derived key is the
reference's commitment to  iFFT(e) + (b_0 + b_1 X + ..)(X^n - 1)  (prover.rs:77-91 with mask_polynomial :457-466) on the
monomial key.  The right-hand side is computed by the C oracle (its iNTT, then its MSM over the host SRS); the points of
the key are also checked one by one against [L_i(beta)] g computed from the trapdoor."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import from_mont_arr, mont_arr, pt_arr, rand_fr_mont

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]
FID = {"bn254": 0, "bls12_381": 2}


def _masked_coeffs(cv, evals_mont, blinders, log_n):
    """coefficients (Montgomery rows) of iFFT(evals) + (sum_k b_k X^k)(X^n - 1), n + len(blinders) of them"""
    f = cv.fr
    n = 1 << log_n
    co = from_mont_arr(f, C.ntt(cv.curve_id, np.ascontiguousarray(evals_mont), log_n, inverse=True))
    co = co + [0] * len(blinders)
    for k, b in enumerate(blinders):
        co[k] = (co[k] - b) % f.p
        co[n + k] = (co[n + k] + b) % f.p
    return mont_arr(f, co)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_lagrange_key_points_from_the_trapdoor(hip, cv):
    f = cv.fr
    beta_i = 0x1234567 + cv.curve_id
    beta = mont_arr(f, [beta_i])[0]
    for log_n in (0, 1, 3, 5):
        n = 1 << log_n
        srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n + 3)
        lag = srs.lagrange(log_n, 3)
        assert len(lag) == n + 3
        got = lag.read(0, n + 3)
        w = from_mont_arr(f, C.domain_root(cv.curve_id, log_n).reshape(1, -1))[0] if log_n else 1
        zh = (pow(beta_i, n, f.p) - 1) % f.p
        g = pt_arr(cv, cv.g).reshape(1, -1)
        for i in range(n):
            wi = pow(w, i, f.p)
            # L_i(beta) = w^i (beta^n - 1) / (n (beta - w^i))
            li = wi * zh % f.p * pow(n * (beta_i - wi) % f.p, -1, f.p) % f.p
            exp = C.msm(cv.curve_id, g, mont_arr(f, [li]))
            assert (got[i] == exp).all(), (log_n, i)
        for k in range(3):
            exp = C.msm(cv.curve_id, g, mont_arr(f, [pow(beta_i, k, f.p) * zh % f.p]))
            assert (got[n + k] == exp).all(), (log_n, "vanishing", k)
        lag.free()
        srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [4, 10, 13])
def test_commit_of_evaluations_equals_commit_of_masked_coefficients(hip, cv, log_n):
    f = cv.fr
    n = 1 << log_n
    beta = mont_arr(f, [0xBEEF0000 + log_n])[0]
    host = C.srs_gen(cv.curve_id, beta, pt_arr(cv, cv.g), n + 3)
    srs = hip.Srs.load(cv.curve_id, host)
    lag = srs.lagrange(log_n, 3)
    small = mont_arr(f, [(7 * i) % 251 for i in range(n)])            # witness-like: bytes
    cases = [(rand_fr_mont(cv, n, 5), [3, 5, 0]), (small, [f.p - 1, 12345, 99]), (np.zeros((n, 4), dtype=np.uint64), [1, 2, 3]),
             (rand_fr_mont(cv, n, 6), [0, 0, 0])]
    for evals, bl in cases:
        sc = np.concatenate([evals, mont_arr(f, bl)])
        got, inf = hip.msm(lag, np.ascontiguousarray(sc))
        co = _masked_coeffs(cv, evals, bl, log_n)
        exp = C.msm(cv.curve_id, host, np.ascontiguousarray(co))
        assert (got == exp).all() and inf == (0 if exp.any() else 1)
        # two blinders only (the wire polynomials): a shorter MSM on the same key
        sc2 = np.concatenate([evals, mont_arr(f, bl[:2])])
        got2, _ = hip.msm(lag, np.ascontiguousarray(sc2))
        co2 = _masked_coeffs(cv, evals, bl[:2], log_n)
        assert (got2 == C.msm(cv.curve_id, host[:n + 2], np.ascontiguousarray(co2))).all()
    lag.free()
    srs.free()


def test_lagrange_key_argument_errors(hip):
    cv = R.BN254
    srs = hip.Srs.gen_for_testing(0, mont_arr(cv.fr, [5])[0], pt_arr(cv, cv.g), 18)
    with pytest.raises(hip.JfHipError) as e:
        srs.lagrange(4, 3)          # 16 + 3 points needed, 18 held
    assert "needs 19" in e.value.msg
    with pytest.raises(hip.JfHipError):
        srs.lagrange(29, 0)         # beyond the two-adicity of BN254 Fr
    lag = srs.lagrange(4, 2)
    assert len(lag) == 18
    lag.free()
    srs.free()
