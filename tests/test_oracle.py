"""CPU suite: pins the oracle (C restatement + big-int Python) against the golden fixtures and
the algebraic identities the reference's own tests assert."""
import hashlib

import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import (FIELD_ID, from_mont_arr, golden, gpt, h2i, mont_arr, pt_arr, pt_from, rand_fr_mont, ref_kat_cases,
                        seeded_scalars)

CURVES = [R.BN254, R.BLS12_381]


def test_constants_known_answers():
    g = golden("constants")
    for cv in CURVES:
        k = g[cv.name]
        assert h2i(k["fq_p"]) == cv.fq.p and h2i(k["fr_p"]) == cv.fr.p
        assert R.on_curve(cv, cv.g)
        assert R.ec_mul(cv, cv.fr.p, cv.g) is None
        assert gpt(k["g2"]) == R.ec_add(cv, cv.g, cv.g)
        assert pow(cv.fr_generator, (cv.fr.p - 1) >> cv.two_adicity, cv.fr.p) == cv.two_adic_root
        # C side: generator on curve, 2G
        assert C.ec_on_curve(cv.curve_id, pt_arr(cv, cv.g))
        assert pt_from(cv, C.ec_add(cv.curve_id, pt_arr(cv, cv.g), pt_arr(cv, cv.g))) == gpt(k["g2"])
        assert from_mont_arr(cv.fr, C.domain_root(cv.curve_id, cv.two_adicity))[0] == cv.two_adic_root
    # EIP-196 alt_bn128 doubling vector
    assert g["bn254"]["g2"][0] == "0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3"


@pytest.mark.parametrize("fname", list(FIELD_ID))
def test_field_golden(fname):
    g = golden("field")[fname]
    f = {x.name: x for cv in CURVES for x in (cv.fr, cv.fq)}[fname]
    a = mont_arr(f, [h2i(x) for x in g["a"]])
    b = mont_arr(f, [h2i(x) for x in g["b"]])
    for op in ("add", "sub", "mul"):
        got = C.fp_binop(FIELD_ID[fname], op, a, b)
        assert (got == mont_arr(f, [h2i(x) for x in g[op]])).all(), op
    for op in ("neg", "inv"):
        got = C.fp_unop(FIELD_ID[fname], op, a)
        assert (got == mont_arr(f, [h2i(x) for x in g[op]])).all(), op
    # Montgomery round trip
    raw = np.array([f.limbs(h2i(x)) for x in g["a"]], dtype=np.uint64)
    assert (C.fp_unop(FIELD_ID[fname], "to_mont", raw) == a).all()
    assert (C.fp_unop(FIELD_ID[fname], "from_mont", a) == raw).all()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_msm_golden(cv):
    g = golden("msm")[cv.name]
    beta, gen = h2i(g["beta"]), gpt(g["g"])
    srs = C.srs_gen(cv.curve_id, mont_arr(cv.fr, [beta])[0], pt_arr(cv, gen), (1 << 12) + 8)
    for i, p in g["srs_spot"].items():
        assert pt_from(cv, srs[int(i)]) == gpt(p)
    for case in g["cases"]:
        n, off = case["n"], case["offset"]
        sc = [h2i(s) for s in case["scalars"]] if "scalars" in case else seeded_scalars(cv, case["scalar_seed"], n)
        got = pt_from(cv, C.msm(cv.curve_id, srs[off:off + n], mont_arr(cv.fr, sc)))
        assert got == gpt(case["expect"]), (n, case["set"])
    bases = np.array([pt_arr(cv, gpt(p)) for p in g["arbitrary_bases"]])
    for case in g["arbitrary_cases"]:
        sc = mont_arr(cv.fr, [h2i(s) for s in case["scalars"]])
        assert pt_from(cv, C.msm(cv.curve_id, bases, sc)) == gpt(case["expect"]), case["set"]
        assert pt_from(cv, C.msm_naive(cv.curve_id, bases, sc)) == gpt(case["expect"]), case["set"]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ntt_golden(cv):
    g = golden("ntt")[cv.name]
    for k, v in g["roots"].items():
        assert from_mont_arr(cv.fr, C.domain_root(cv.curve_id, int(k)))[0] == h2i(v)
    for case in g["cases"]:
        log_n, off, in_len, seed = case["log_n"], case["offset"], case["in_len"], case["seed"]
        n = 1 << log_n
        coeffs = seeded_scalars(cv, seed, in_len)
        data = np.zeros((1, n, 4), dtype=np.uint64)
        data[0, :in_len] = mont_arr(cv.fr, coeffs)
        cos = None if off == 1 else mont_arr(cv.fr, [off])[0]
        ev = C.ntt(cv.curve_id, data, log_n, False, cos)
        assert hashlib.sha256(ev.tobytes()).hexdigest() == case["fwd_sha256"], case
        for i, v in case["spots"].items():
            assert from_mont_arr(cv.fr, ev[0, int(i)])[0] == h2i(v)
        back = C.ntt(cv.curve_id, ev, log_n, True, cos)
        assert (back == data).all()
        evals_in = mont_arr(cv.fr, seeded_scalars(cv, seed ^ 0x5555, n)).reshape(1, n, 4)
        co = C.ntt(cv.curve_id, evals_in, log_n, True, cos)
        assert hashlib.sha256(co.tobytes()).hexdigest() == case["inv_sha256"], case


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_open_golden(cv):
    g = golden("open")[cv.name]
    for case in g["cases"]:
        coeffs = mont_arr(cv.fr, [h2i(c) for c in case["coeffs"]])
        z = mont_arr(cv.fr, [h2i(case["z"])])[0]
        q, ev = C.open_witness(cv.curve_id, coeffs, z)
        assert from_mont_arr(cv.fr, ev)[0] == h2i(case["eval"])
        assert from_mont_arr(cv.fr, q) == [h2i(c) for c in case["witness"]]
        assert (C.poly_eval(cv.curve_id, coeffs, z) == ev).all()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_commit_equals_p_of_beta_medium(cv):
    """MSM(powers_of_g, coeffs) == [p(beta)] g at a size the Python oracle cannot reach."""
    n = 1 << 14
    beta = mont_arr(cv.fr, [0x1234567 + cv.curve_id])[0]
    srs = C.srs_gen(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    sc = rand_fr_mont(cv, n, 7)
    got = C.msm(cv.curve_id, srs, sc)
    pb = C.poly_eval(cv.curve_id, sc, beta)
    assert (got == C.ec_mul(cv.curve_id, pt_arr(cv, cv.g), pb)).all()
    # linearity of the NTT and round trip at a medium size
    a = rand_fr_mont(cv, 1 << 12, 1).reshape(1, -1, 4)
    b = rand_fr_mont(cv, 1 << 12, 2).reshape(1, -1, 4)
    s = C.fp_binop(0 if cv.curve_id == 0 else 2, "add", a, b)
    cos = mont_arr(cv.fr, [cv.fr_generator])[0]
    fa, fb, fs = (C.ntt(cv.curve_id, x, 12, False, cos) for x in (a, b, s))
    assert (C.fp_binop(0 if cv.curve_id == 0 else 2, "add", fa, fb) == fs).all()
    assert (C.ntt(cv.curve_id, fs, 12, True, cos) == s).all()


def test_intt_reference_known_answers():
    """The one place the reference holds literal ark-poly FFT outputs: test_build_l_helper,
    pcs/src/multilinear_kzg/util.rs:428-563 (Evaluations::interpolate = domain.ifft, util.rs:177,184).  Pins the oracle's
    root choice, ordering and n^-1 scaling of the inverse transform (and, through the round trip, the forward one)."""
    cv = R.BLS12_381
    cases, g = ref_kat_cases()
    # the size-4 root the reference's comment names is the oracle's domain generator
    assert R.domain_root(cv, 2) == h2i(g["size4_root_of_unity"])
    assert from_mont_arr(cv.fr, C.domain_root(cv.curve_id, 2))[0] == h2i(g["size4_root_of_unity"])
    for cite, log_n, evals, coeffs in cases:
        # big-integer restatement
        assert R.ifft(cv, evals, log_n) == coeffs, cite
        assert R.fft(cv, coeffs, log_n) == evals, cite
        # C restatement
        data = mont_arr(cv.fr, evals).reshape(1, 1 << log_n, 4)
        got = C.ntt(cv.curve_id, data, log_n, True, None)
        assert from_mont_arr(cv.fr, got) == coeffs, cite
        assert (C.ntt(cv.curve_id, got, log_n, False, None) == data).all(), cite


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_mle_oracle_identities(cv):
    """the multilinear restatement against the scheme's own defining identities with the trapdoor known
    (pcs/src/multilinear_kzg/srs.rs:337-340 asserts the first of them; verify_internal, mod.rs:368-420, the second in
    the exponent): commit = [f(t)] g and f(t) - f(z) = sum_i q_i(t_(i+1..)) (t_i - z_i), proofs[i] = [q_i(t_(i+1..))] g"""
    r = cv.fr.p
    nv = 4
    rng = R.SplitMix64(77 + cv.curve_id)
    t = [rng.field(cv.fr) for _ in range(nv)]
    z = [rng.field(cv.fr) for _ in range(nv)]
    evals = [rng.field(cv.fr) for _ in range(1 << nv)]
    srs = R.mle_srs_for_testing(cv, cv.g, t)
    assert [len(l) for l in srs] == [16, 8, 4, 2, 1] and srs[nv] == [cv.g]
    ft = R.mle_evaluate(cv, evals, t)
    assert R.mle_commit(cv, srs, nv, evals) == R.ec_mul(cv, ft, cv.g)
    proofs, val = R.mle_open(cv, srs, nv, evals, z)
    assert val == R.mle_evaluate(cv, evals, z)
    # exponents of the proofs: q_i over the remaining variables, evaluated at t_(i+1..)
    f, acc = list(evals), 0
    for i in range(nv):
        q = [(f[2 * b + 1] - f[2 * b]) % r for b in range(len(f) // 2)]
        f = [(f[2 * b] + q[b] * z[i]) % r for b in range(len(f) // 2)]
        qt = R.mle_evaluate(cv, q, t[i + 1:]) if i + 1 < nv else q[0]
        assert proofs[i] == R.ec_mul(cv, qt, cv.g)
        acc = (acc + qt * (t[i] - z[i])) % r
    assert acc == (ft - val) % r
    # a polynomial with fewer variables than the parameter: the first levels are skipped
    sub = evals[:4]
    assert R.mle_commit(cv, srs, nv, sub) == R.ec_mul(cv, R.mle_evaluate(cv, sub, t[2:]), cv.g)
