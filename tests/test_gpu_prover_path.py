"""GPU: the device-side operation sequence of one TurboPlonk proof (jellyfish_amd/prover_path.py)
against the same sequence run on the oracle: round-1/2 commitments, the quotient commitments (coset
NTTs -> quotient evaluations -> inverse coset NTT -> split -> commit) and the opening commitments."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_prover_path_matches_oracle(hip, cv):
    from jellyfish_amd import prover_path as P

    log_n = 7
    n = 1 << log_n
    assert P.domain_size_ratio(n, 5) == 6 and P.quot_domain_size(n, 5) == 8 * n
    assert P.domain_size_ratio(1 << 20, 6) == 7 and P.quot_domain_size(1 << 20, 6) == 8 << 20
    cid = cv.curve_id
    fid = 0 if cid == 0 else 2
    beta = mont_arr(cv.fr, [987654321])[0]
    srs = hip.Srs.gen_for_testing(cid, beta, pt_arr(cv, cv.g), n + 8)
    srs_host = srs.read(0, n + 8)
    rand = lambda count, seed: rand_fr_mont(cv, count, 9000 + seed)
    hp = P.TurboPlonkHotPath(cid, log_n, srs, rand)
    out = hp.prove().cpu().numpy()
    hp.stream.sync()
    pt = 16 * cv.fq.limbs64
    got = [out[i * hp.rec:i * hp.rec + pt].view(np.uint64) for i in range(13)]

    # ---- the same sequence on the oracle ----
    m, log_m = hp.m, hp.log_m
    blind = rand(64, 17)
    wires = C.ntt(cid, rand(5 * n, 14).reshape(5, n, 4), log_n, True)
    pub = C.ntt(cid, rand(n, 16).reshape(1, n, 4), log_n, True)[0]
    ch = rand(4, 18)
    perm_vals = C.plonk_perm_product(cid, log_n, rand(5 * n, 14).reshape(5, n, 4), rand(5 * n, 15).reshape(5, n, 4),
                                     rand(5 * n, 25).reshape(5, n, 4), ch[1], ch[2])
    perm = C.ntt(cid, perm_vals.reshape(1, n, 4), log_n, True)[0]
    wire_polys = [np.concatenate([wires[j], blind[2 * j:2 * j + 2]]) for j in range(5)]
    perm_poly = np.concatenate([perm, blind[10:13]])
    for j in range(5):
        assert (got[j] == C.msm(cid, srs_host[:n + 2], wire_polys[j])).all(), j
    assert (got[5] == C.msm(cid, srs_host[:n + 3], perm_poly)).all()
    h = C.fp_unop(fid, "to_mont", np.array([cv.fr_generator, 0, 0, 0], dtype=np.uint64))

    def coset(poly):
        buf = np.zeros((1, m, 4), dtype=np.uint64)
        buf[0, :len(poly)] = poly
        return C.ntt(cid, buf, log_m, False, h)[0]

    sel_c = rand(13 * n, 11).reshape(13, n, 4)
    sig_c = rand(5 * n, 12).reshape(5, n, 4)
    quot_ev = C.plonk_quotient_turbo(cid, log_n, log_m, np.array([coset(p) for p in sel_c]),
                                     np.array([coset(p) for p in sig_c]), np.array([coset(p) for p in wire_polys]),
                                     coset(perm_poly), coset(pub), rand(5, 13), *ch)
    quot = C.ntt(cid, quot_ev.reshape(1, m, 4), log_m, True, h)[0]
    for j in range(5):
        chunk = quot[j * (n + 2):(j + 1) * (n + 2)]
        assert (got[6 + j] == C.msm(cid, srs_host[:n + 2], chunk)).all(), j

    def lincomb(polys, coeffs, length):
        acc = np.zeros((length, 4), dtype=np.uint64)
        for p, c in zip(polys, coeffs):
            p = p[:length]
            acc[:len(p)] = C.fp_binop(fid, "add", acc[:len(p)], C.fp_binop(fid, "mul", p, np.tile(c, (len(p), 1))))
        return acc

    lc = rand(32, 21)
    lin = lincomb(list(sel_c) + [perm_poly, sig_c[4]] + [quot[j * (n + 2):(j + 1) * (n + 2)] for j in range(5)], lc,
                  n + 3)
    opening = lincomb([lin] + wire_polys + list(sig_c[:4]), lc, n + 3)
    q, _ = C.open_witness(cid, opening, rand(1, 19)[0])
    assert (got[11] == C.msm(cid, srs_host[:n + 2], q)).all()
    # round-4 evaluations
    ev = hp.evals.cpu().numpy().view(np.uint64).reshape(10, 4)
    for j in range(5):
        assert (ev[j] == C.poly_eval(cid, wire_polys[j], rand(1, 19)[0])).all()
    for j in range(4):
        assert (ev[5 + j] == C.poly_eval(cid, sig_c[j], rand(1, 19)[0])).all()
    assert (ev[9] == C.poly_eval(cid, perm_poly, rand(1, 20)[0])).all()
    q2, _ = C.open_witness(cid, perm_poly, rand(1, 20)[0])
    assert (got[12] == C.msm(cid, srs_host[:n + 2], q2)).all()
    srs.free()


@pytest.mark.gpu
def test_sharded_commit_mode_matches_single_device(hip):
    """world size 1: the sharded-commit code path (partials -> fold) gives the same 13 commitments"""
    from jellyfish_amd import prover_path as P

    cv = R.BN254
    log_n = 8
    n = 1 << log_n
    beta = mont_arr(cv.fr, [24680])[0]
    srs = hip.Srs.gen_for_testing(0, beta, pt_arr(cv, cv.g), n + 8)
    rand = lambda count, seed: rand_fr_mont(cv, count, 4000 + seed)
    a = P.TurboPlonkHotPath(0, log_n, srs, rand).prove().cpu().numpy().copy()
    b = P.TurboPlonkHotPath(0, log_n, srs, rand, sharded=True).prove().cpu().numpy().copy()
    rec = a.size // 13
    for j in range(13):
        assert (a[j * rec:j * rec + 68] == b[j * rec:j * rec + 68]).all(), j


@pytest.mark.gpu
@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_ultra_prover_path_matches_oracle(hip, cv):
    """UltraPlonk (Plookup) sequence: 18 commitments and 27 evaluations against the oracle"""
    from jellyfish_amd import prover_path as P

    log_n = 6
    n = 1 << log_n
    cid = cv.curve_id
    fid = 0 if cid == 0 else 2
    beta = mont_arr(cv.fr, [192837465])[0]
    srs = hip.Srs.gen_for_testing(cid, beta, pt_arr(cv, cv.g), n + 8)
    srs_host = srs.read(0, n + 8)
    rand = lambda count, seed: rand_fr_mont(cv, count, 12000 + seed)
    hp = P.UltraPlonkHotPath(cid, log_n, srs, rand)
    assert hp.m == 8 * n and hp.npolys == 35 and hp.ncommit == 18
    out = hp.prove().cpu().numpy()
    hp.stream.sync()
    pt = 16 * cv.fq.limbs64
    got = [out[i * hp.rec:i * hp.rec + pt].view(np.uint64) for i in range(18)]
    m, log_m = hp.m, hp.log_m
    blind = rand(64, 17)
    intt = lambda vals: C.ntt(cid, vals.reshape(-1, n, 4), log_n, True)
    commit = lambda poly: C.msm(cid, srs_host[:len(poly)], poly)
    wire_polys = [np.concatenate([w, blind[2 * j:2 * j + 2]]) for j, w in enumerate(intt(rand(6 * n, 14)))]
    pub = intt(rand(n, 16))[0]
    ch = rand(5, 18)
    srt = rand(2 * n - 1, 26)
    h_polys = [np.concatenate([h, blind[12 + 4 * j:15 + 4 * j]])
               for j, h in enumerate(intt(np.concatenate([srt[:n], srt[n - 1:]])))]
    perm_vals = C.plonk_perm_product(cid, log_n, rand(6 * n, 14).reshape(6, n, 4), rand(6 * n, 15).reshape(6, n, 4),
                                     rand(6 * n, 25).reshape(6, n, 4), ch[1], ch[2])
    perm_poly = np.concatenate([intt(perm_vals)[0], blind[20:23]])
    plprod_vals = C.plonk_lookup_product(cid, log_n, rand(n, 23), rand(n, 24), srt, ch[1], ch[2])
    plprod_poly = np.concatenate([intt(plprod_vals)[0], blind[24:27]])
    for j in range(6):
        assert (got[j] == commit(wire_polys[j])).all(), j
    assert (got[6] == commit(h_polys[0])).all() and (got[7] == commit(h_polys[1])).all()
    assert (got[8] == commit(perm_poly)).all() and (got[9] == commit(plprod_poly)).all()
    h = C.fp_unop(fid, "to_mont", np.array([cv.fr_generator, 0, 0, 0], dtype=np.uint64))

    def coset(poly):
        buf = np.zeros((1, m, 4), dtype=np.uint64)
        buf[0, :len(poly)] = poly
        return C.ntt(cid, buf, log_m, False, h)[0]

    sel_c = rand(14 * n, 11).reshape(14, n, 4)
    sig_c = rand(6 * n, 12).reshape(6, n, 4)
    tab_c = rand(4 * n, 22).reshape(4, n, 4)
    lookup = [coset(p) for p in tab_c] + [coset(h_polys[0]), coset(h_polys[1]), coset(plprod_poly)]
    quot_ev = C.plonk_quotient_ultra(cid, log_n, log_m, np.array([coset(p) for p in sel_c]),
                                     np.array([coset(p) for p in sig_c]), np.array([coset(p) for p in wire_polys]),
                                     coset(perm_poly), coset(pub), np.array(lookup), rand(6, 13), *ch)
    quot = C.ntt(cid, quot_ev.reshape(1, m, 4), log_m, True, h)[0]
    chunks = [quot[j * (n + 2):(j + 1) * (n + 2)] for j in range(6)]
    for j in range(6):
        assert (got[10 + j] == commit(chunks[j])).all(), j

    def lincomb(polys, coeffs, length):
        acc = np.zeros((length, 4), dtype=np.uint64)
        for p, c in zip(polys, coeffs):
            p = p[:length]
            acc[:len(p)] = C.fp_binop(fid, "add", acc[:len(p)], C.fp_binop(fid, "mul", p, np.tile(c, (len(p), 1))))
        return acc

    lc = rand(32, 21)
    zeta, zw = rand(1, 19)[0], rand(1, 20)[0]
    lin = lincomb(list(sel_c[:13]) + [perm_poly, sig_c[5]] + chunks + [plprod_poly, h_polys[1]], lc, n + 3)
    at_zeta = wire_polys + list(sig_c[:5]) + [tab_c[0], tab_c[1], h_polys[0], sel_c[13], tab_c[2], tab_c[3]]
    at_zw = [perm_poly, plprod_poly, tab_c[0], tab_c[1], h_polys[0], h_polys[1], sel_c[13], wire_polys[3],
             wire_polys[4], tab_c[2]]
    q, _ = C.open_witness(cid, lincomb([lin] + at_zeta, lc, n + 3), zeta)
    assert (got[16] == commit(q)).all()
    q2, _ = C.open_witness(cid, lincomb(at_zw, lc, n + 3), zw)
    assert (got[17] == commit(q2)).all()
    ev = hp.evals.cpu().numpy().view(np.uint64).reshape(27, 4)
    for j, poly in enumerate(at_zeta):
        assert (ev[j] == C.poly_eval(cid, poly, zeta)).all(), j
    for j, poly in enumerate(at_zw):
        assert (ev[17 + j] == C.poly_eval(cid, poly, zw)).all(), j
    srs.free()
