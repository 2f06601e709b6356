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


def random_ultra_instance(cv, log_n, log_m, seed):
    m = 1 << log_m
    f = lambda polys, s: rand_fr_mont(cv, polys * m, seed + s).reshape(polys, m, 4)
    return dict(sel=f(14, 1), sig=f(6, 2), wires=f(6, 3), zperm=f(1, 4)[0], pi=f(1, 5)[0], lookup=f(7, 8),
                k=rand_fr_mont(cv, 6, seed + 6), ch=rand_fr_mont(cv, 5, seed + 7))  # alpha beta gamma tau alpha_base


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ultra_quotient_c_matches_bigint(cv):
    """prover.rs:766-882 (Plookup part) on top of the 6-wire gate / copy-constraint parts"""
    log_n, log_m = 2, 5
    inst = random_ultra_instance(cv, log_n, log_m, 300)
    ints = lambda a: [from_mont_arr(cv.fr, p) for p in a]
    ch = from_mont_arr(cv.fr, inst["ch"])
    exp = R.quotient_evals_ultra(cv, log_n, log_m, ints(inst["sel"]), ints(inst["sig"]), ints(inst["wires"]),
                                 from_mont_arr(cv.fr, inst["zperm"]), from_mont_arr(cv.fr, inst["pi"]),
                                 ints(inst["lookup"]), from_mont_arr(cv.fr, inst["k"]), ch[0], ch[1], ch[2], ch[3], ch[4])
    got = C.plonk_quotient_ultra(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                 inst["pi"], inst["lookup"], inst["k"], *inst["ch"])
    assert from_mont_arr(cv.fr, got) == exp


def satisfied_ultra_instance(cv, log_n, log_m, seed):
    """A satisfied UltraPlonk instance in the reference's own shape (relation/src/constraint_system.rs:
    1495-1624): gates as in satisfied_instance with a 6th (range) wire, identity permutation, and a
    Plookup argument over a range table {0..n-2}: every lookup wire value is a table element,
    q_lookup = 0 (pure range lookups), h1/h2 = halves of the sorted merged vector, prod_lookup = the
    running product of compute_lookup_prod_vector."""
    r = cv.fr.p
    n = 1 << log_n
    rng = R.SplitMix64(seed)
    w = [[rng.field(cv.fr) for _ in range(n)] for _ in range(5)]
    w.append([rng.next() % n for _ in range(n)])  # range wire: looked-up values (reduced into the table below)
    sel = [[rng.field(cv.fr) for _ in range(n)] for _ in range(13)] + [[0] * n]
    for i in range(n):
        wi = [w[j][i] for j in range(5)]
        rest = (sum(sel[j][i] * wi[j] for j in range(4)) + sel[4][i] * wi[0] * wi[1] + sel[5][i] * wi[2] * wi[3]
                + sel[12][i] * wi[0] * wi[1] * wi[2] * wi[3] * wi[4]
                + sum(sel[6 + j][i] * pow(wi[j], 5, r) for j in range(4)) - sel[10][i] * wi[4])
        sel[11][i] = (-rest) % r
    k = [1] + [rng.field(cv.fr) for _ in range(5)]
    return w, sel, k


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ultra_quotient_of_satisfied_circuit_is_a_low_degree_polynomial(cv):
    log_n, log_m = 3, 6
    n, m = 1 << log_n, 1 << log_m
    r = cv.fr.p
    w, sel, k = satisfied_ultra_instance(cv, log_n, log_m, 11)
    rng = R.SplitMix64(5)
    alpha, beta, gamma, tau = (rng.field(cv.fr) for _ in range(4))
    g = cv.fr_generator
    coset = lambda evals_on_h: R.fft(cv, R.ifft(cv, evals_on_h, log_n), log_m, g)
    wn = R.domain_root(cv, log_n)
    xs = [pow(wn, i, r) for i in range(n)]
    # Plookup vectors on H exactly as the reference builds them (relation/src/constraint_system.rs):
    # compute_range_table :1630-1645, compute_lookup_sorted_vec_polynomials :1577-1624,
    # compute_lookup_prod_polynomial :1516-1575.  q_lookup = 0, so merged table = range table and the
    # merged lookup witness = the range wire.
    range_size = n // 2
    table = list(range(range_size)) + [0] * (n - range_size)
    w[5] = [v % range_size for v in w[5]]
    lookups = w[5][: n - 1]
    cnt = {}
    for v in lookups:
        cnt[v] = cnt.get(v, 0) + 1
    srt = []
    for v in table:
        if v in cnt:
            srt += [v] * (1 + cnt.pop(v))
        else:
            srt.append(v)
    assert len(srt) == 2 * n - 1
    h1, h2 = srt[:n], srt[n - 1:]
    b1 = (1 + beta) % r
    gb = gamma * b1 % r
    prod = [1]
    for j in range(n - 2):
        num = b1 * (gamma + lookups[j]) % r * (gb + table[j] + beta * table[j + 1]) % r
        den = (gb + srt[j] + beta * srt[j + 1]) % r * (gb + srt[n - 1 + j] + beta * srt[n + j]) % r
        prod.append(prod[-1] * num % r * pow(den, -1, r) % r)
    prod.append(1)
    zero = [0] * n
    lookup = [coset(table), coset(zero), coset(zero), coset(zero), coset(h1), coset(h2), coset(prod)]
    sig = [coset([k[j] * x % r for x in xs]) for j in range(6)]
    t = R.quotient_evals_ultra(cv, log_n, log_m, [coset(s) for s in sel], sig, [coset(x) for x in w], coset([1] * n),
                               coset(zero), lookup, k, alpha, beta, gamma, tau)
    coeffs = R.ifft(cv, t, log_m, g)
    assert any(coeffs[: 6 * n]) and not any(coeffs[6 * n:])


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


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n,log_m", [(2, 5), (6, 9), (11, 14)])
def test_ultra_quotient_gpu_matches_oracle(hip, cv, log_n, log_m):
    import torch

    inst = random_ultra_instance(cv, log_n, log_m, 900 + log_n)
    m = 1 << log_m
    dev = {k: torch.from_numpy(np.ascontiguousarray(inst[k]).view(np.int64)).cuda()
           for k in ("sel", "sig", "wires", "zperm", "pi", "lookup")}
    quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    for rep in range(2):
        hip.plonk_quotient_ultra_device(cv.curve_id, log_n, log_m, dev["sel"].data_ptr(), dev["sig"].data_ptr(),
                                        dev["wires"].data_ptr(), dev["zperm"].data_ptr(), dev["pi"].data_ptr(),
                                        dev["lookup"].data_ptr(), inst["k"], inst["ch"], quot.data_ptr(), stream=st)
    st.sync()
    got = quot.cpu().numpy().view(np.uint64).reshape(m, 4)
    args = (cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"], inst["pi"],
            inst["lookup"], inst["k"], *inst["ch"])
    exp = C.plonk_quotient_ultra(*args, out=C.plonk_quotient_ultra(*args))
    assert (got == exp).all()
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("ultra", [False, True], ids=["turbo", "ultra"])
@pytest.mark.parametrize("log_n,log_m,G", [(2, 5, 2), (6, 9, 4), (6, 9, 8), (10, 13, 8)])
def test_quotient_coset_slices_gpu_match_oracle(hip, cv, ultra, log_n, log_m, G):
    """SURVEY.md 8e coset sharding: rank g of G evaluates the points i = g (mod G) of the quotient coset as the coset of
    size m / G with offset h w_m^g (jfhip_plonk_quotient_coset_device); the G slices together are the oracle's evaluation
    of the whole coset, bit for bit (G = 8 = m / n: one point of every Z_H class per slice, "next" = index + 1)."""
    import torch

    inst = (random_ultra_instance if ultra else random_instance)(cv, log_n, log_m, 1300 + log_n + G)
    m = 1 << log_m
    if ultra:
        exp = C.plonk_quotient_ultra(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                     inst["pi"], inst["lookup"], inst["k"], *inst["ch"])
    else:
        exp = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, inst["sel"], inst["sig"], inst["wires"], inst["zperm"],
                                     inst["pi"], inst["k"], *inst["ch"])
    r = cv.fr.p
    w_m = R.domain_root(cv, log_m)
    st = hip.Stream()
    ml = m // G
    for g in range(G):
        sl = lambda a: torch.from_numpy(np.ascontiguousarray(a[..., g::G, :]).view(np.int64)).cuda()
        dev = {k: sl(np.asarray(inst[k])) for k in ("sel", "sig", "wires", "zperm", "pi") + (("lookup",) if ultra else ())}
        quot = torch.zeros(ml * 4, dtype=torch.int64, device="cuda")
        off = mont_arr(cv.fr, [cv.fr_generator * pow(w_m, g, r) % r])[0]
        hip.plonk_quotient_coset_device(cv.curve_id, 6 if ultra else 5, log_n, ml.bit_length() - 1, off,
                                        dev["sel"].data_ptr(), dev["sig"].data_ptr(), dev["wires"].data_ptr(),
                                        dev["zperm"].data_ptr(), dev["pi"].data_ptr(),
                                        dev["lookup"].data_ptr() if ultra else 0, inst["k"], inst["ch"], quot.data_ptr(),
                                        stream=st)
        st.sync()
        got = quot.cpu().numpy().view(np.uint64).reshape(ml, 4)
        assert (got == np.asarray(exp)[g::G]).all(), g
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("ultra", [False, True], ids=["turbo", "ultra"])
@pytest.mark.parametrize("mask", [0, 0b1011111001100, 0x1FFF], ids=["none", "bench-like", "all-turbo"])
def test_quotient_masked_gpu_matches_oracle(hip, cv, ultra, mask):
    """jfhip_plonk_quotient_masked_device: selectors flagged identically zero are never read (their rows hold garbage
    here) and a NULL public input stands for pub_inp_poly = 0; the result equals the oracle's evaluation with zero
    polynomials in those places, bit for bit."""
    import torch

    log_n, log_m = 6, 9
    inst = (random_ultra_instance if ultra else random_instance)(cv, log_n, log_m, 4100 + mask % 97)
    m = 1 << log_m
    nsel = 14 if ultra else 13
    sel_zeroed = np.array(inst["sel"]).copy()
    sel_garbage = np.array(inst["sel"]).copy()
    for j in range(nsel):
        if (mask >> j) & 1:
            sel_zeroed[j] = 0
            sel_garbage[j] = np.uint64(0xFFFFFFFFFFFFFFFF)     # not even a field element: must not be read
    pi_zero = np.zeros_like(np.asarray(inst["pi"]))
    if ultra:
        exp = C.plonk_quotient_ultra(cv.curve_id, log_n, log_m, sel_zeroed, inst["sig"], inst["wires"], inst["zperm"],
                                     pi_zero, inst["lookup"], inst["k"], *inst["ch"])
    else:
        exp = C.plonk_quotient_turbo(cv.curve_id, log_n, log_m, sel_zeroed, inst["sig"], inst["wires"], inst["zperm"],
                                     pi_zero, inst["k"], *inst["ch"])
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    dev = {k: up(np.asarray(inst[k])) for k in ("sig", "wires", "zperm") + (("lookup",) if ultra else ())}
    dev["sel"] = up(sel_garbage)
    quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    off = mont_arr(cv.fr, [cv.fr_generator])[0]
    hip.plonk_quotient_masked_device(cv.curve_id, 6 if ultra else 5, log_n, log_m, off, dev["sel"].data_ptr(),
                                     dev["sig"].data_ptr(), dev["wires"].data_ptr(), dev["zperm"].data_ptr(), 0,
                                     dev["lookup"].data_ptr() if ultra else 0, inst["k"], inst["ch"], mask,
                                     quot.data_ptr(), stream=st)
    st.sync()
    got = quot.cpu().numpy().view(np.uint64).reshape(m, 4)
    assert (got == np.asarray(exp)).all()
    with pytest.raises(hip.JfHipError):   # a mask bit beyond this circuit type's selectors
        hip.plonk_quotient_masked_device(cv.curve_id, 6 if ultra else 5, log_n, log_m, off, dev["sel"].data_ptr(),
                                         dev["sig"].data_ptr(), dev["wires"].data_ptr(), dev["zperm"].data_ptr(), 0,
                                         dev["lookup"].data_ptr() if ultra else 0, inst["k"], inst["ch"], 1 << nsel,
                                         quot.data_ptr(), stream=st)
    st.close()
