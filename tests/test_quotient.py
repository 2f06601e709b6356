"""TurboPlonk quotient evaluations.

CPU part: the C restatement equals the big-integer restatement of
plonk/src/proof_system/prover.rs:587-752, and -- the reference's own check
(test_plonk_prover_polynomials, plonk/src/proof_system/snark.rs:1166-1472: the constraint identities
vanish on H) -- for a SATISFIED circuit the quotient evaluations interpolate to a polynomial of the
expected degree (the division by Z_H is exact).
GPU part: the HIP kernel equals the oracle bit for bit."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import from_mont_arr, mont_arr, rand_fr_mont

CURVES = [R.BN254, R.BLS12_381]


def random_instance(cv, log_n, log_m, seed):
    m = 1 << log_m
    f = lambda polys, s: rand_fr_mont(cv, polys * m, seed + s).reshape(polys, m, 4)
    return dict(sel=f(13, 1), sig=f(5, 2), wires=f(5, 3), zperm=f(1, 4)[0], pi=f(1, 5)[0],
                k=rand_fr_mont(cv, 5, seed + 6), ch=rand_fr_mont(cv, 4, seed + 7))


def satisfied_instance(cv, log_n, log_m, seed):
    """random wires and selectors on H with q_c solving the gate equation, identity permutation
    (sigma_j = k_j X, so z = 1 satisfies the copy constraints), pub input 0"""
    r = cv.fr.p
    n = 1 << log_n
    rng = R.SplitMix64(seed)
    w = [[rng.field(cv.fr) for _ in range(n)] for _ in range(5)]
    sel = [[rng.field(cv.fr) for _ in range(n)] for _ in range(13)]
    for i in range(n):
        wi = [w[j][i] for j in range(5)]
        rest = (sum(sel[j][i] * wi[j] for j in range(4)) + sel[4][i] * wi[0] * wi[1] + sel[5][i] * wi[2] * wi[3]
                + sel[12][i] * wi[0] * wi[1] * wi[2] * wi[3] * wi[4]
                + sum(sel[6 + j][i] * pow(wi[j], 5, r) for j in range(4)) - sel[10][i] * wi[4])
        sel[11][i] = (-rest) % r
    k = [1] + [rng.field(cv.fr) for _ in range(4)]
    g = cv.fr_generator
    coset = lambda evals_on_h: R.fft(cv, R.ifft(cv, evals_on_h, log_n), log_m, g)
    wn = R.domain_root(cv, log_n)
    xs = [pow(wn, i, r) for i in range(n)]
    sig = [coset([k[j] * x % r for x in xs]) for j in range(5)]
    return dict(sel=[coset(s) for s in sel], sig=sig, wires=[coset(x) for x in w], zperm=coset([1] * n),
                pi=coset([0] * n), k=k)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_quotient_c_matches_bigint(cv):
    log_n, log_m = 2, 5
    inst = random_instance(cv, log_n, log_m, 100)
    ints = lambda a: [from_mont_arr(cv.fr, p) for p in a]
    ch = from_mont_arr(cv.fr, inst["ch"])
    exp = R.quotient_evals_turbo(cv, log_n, log_m, ints(inst["sel"]), ints(inst["sig"]), ints(inst["wires"]),
                                 from_mont_arr(cv.fr, inst["zperm"]), from_mont_arr(cv.fr, inst["pi"]),
                                 from_mont_arr(cv.fr, inst["k"]), ch[0], ch[1], ch[2], ch[3])
    got = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                 inst["pi"], inst["k"], *inst["ch"])
    assert from_mont_arr(cv.fr, got) == exp
    # accumulation: out += alpha_base * t
    twice = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                   inst["pi"], inst["k"], *inst["ch"], out=got)
    assert from_mont_arr(cv.fr, twice) == [2 * e % cv.fr.p for e in exp]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_quotient_of_satisfied_circuit_is_a_low_degree_polynomial(cv):
    log_n, log_m = 3, 6
    n, m = 1 << log_n, 1 << log_m
    inst = satisfied_instance(cv, log_n, log_m, 7)
    rng = R.SplitMix64(99)
    alpha, beta, gamma = (rng.field(cv.fr) for _ in range(3))
    t = R.quotient_evals_turbo(cv, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"], inst["pi"],
                               inst["k"], alpha, beta, gamma)
    coeffs = R.ifft(cv, t, log_m, cv.fr_generator)
    # numerator degree <= 6(n-1) (q_ecc * w0..w4), minus deg Z_H = n  ->  quotient degree <= 5n - 6
    assert any(coeffs[: 5 * n - 5]) and not any(coeffs[5 * n - 5:])
    # an unsatisfied circuit does not divide: flipping one selector value breaks the bound
    bad = [list(p) for p in inst["sel"]]
    bad[11] = R.fft(cv, [(c + (1 if i == 0 else 0)) % cv.fr.p for i, c in
                         enumerate(R.ifft(cv, inst["sel"][11], log_m, cv.fr_generator))], log_m, cv.fr_generator)
    t2 = R.quotient_evals_turbo(cv, log_n, log_m, bad, inst["sig"], inst["wires"], inst["zperm"], inst["pi"], inst["k"],
                                alpha, beta, gamma)
    assert any(R.ifft(cv, t2, log_m, cv.fr_generator)[5 * n - 5:])
    # the C restatement agrees on the satisfied instance
    M = lambda polys: np.array([mont_arr(cv.fr, p) for p in polys])
    got = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, M(inst["sel"]), M(inst["sig"]), M(inst["wires"]),
                                 mont_arr(cv.fr, inst["zperm"]), mont_arr(cv.fr, inst["pi"]), mont_arr(cv.fr, inst["k"]),
                                 *mont_arr(cv.fr, [alpha, beta, gamma, 1]))
    assert from_mont_arr(cv.fr, got) == t


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n,log_m", [(2, 5), (6, 9), (10, 13), (13, 16)])
def test_quotient_gpu_matches_oracle(hip, cv, log_n, log_m):
    import torch

    inst = random_instance(cv, log_n, log_m, 500 + log_n)
    m = 1 << log_m
    dev = {k: torch.from_numpy(np.ascontiguousarray(inst[k]).view(np.int64)).cuda()
           for k in ("sel", "sig", "wires", "zperm", "pi")}
    quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    for rep in range(2):  # second call accumulates, and reuses the cached domain tables
        hip.plonk_quotient_turbo_device(cv.curve_id, log_n, log_m, dev["sel"].data_ptr(), dev["sig"].data_ptr(),
                                        dev["wires"].data_ptr(), dev["zperm"].data_ptr(), dev["pi"].data_ptr(),
                                        inst["k"], inst["ch"], quot.data_ptr(), stream=st)
    st.sync()
    got = quot.cpu().numpy().view(np.uint64).reshape(m, 4)
    once = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                  inst["pi"], inst["k"], *inst["ch"])
    exp = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                 inst["pi"], inst["k"], *inst["ch"], out=once)
    assert (got == exp).all()
    st.close()
