"""Grand-product builders: compute_prod_permutation_polynomial and
compute_lookup_prod_polynomial (relation/src/constraint_system.rs:1394-1420, 1516-1575).

CPU: C restatement == big-integer restatement, and the reference's own identity
(test_arithmetization / check_polynomial-style, relation/src/constraint_system.rs:2356-2419): for a valid
copy-constraint system / lookup the running product closes to 1.
GPU: batch inversion + parallel prefix product == the serial oracle, bit for bit."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import from_mont_arr, mont_arr, rand_fr_mont

CURVES = [R.BN254, R.BLS12_381]


def copy_constraint_system(cv, log_n, nw, seed):
    """witness values constant on the cycles of a random permutation of the nw * n wire positions"""
    r = cv.fr.p
    n = 1 << log_n
    rng = R.SplitMix64(seed)
    k = [1] + [rng.field(cv.fr) for _ in range(nw - 1)]
    wn = R.domain_root(cv, log_n)
    idp = [[k[i] * pow(wn, j, r) % r for j in range(n)] for i in range(nw)]
    pos = list(range(nw * n))
    perm = pos[:]
    for i in range(len(perm) - 1, 0, -1):  # Fisher-Yates
        j = rng.next() % (i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    val = [None] * (nw * n)
    for p in pos:
        if val[p] is None:
            v = rng.field(cv.fr)
            q = p
            while val[q] is None:
                val[q] = v
                q = perm[q]
    wires = [val[i * n:(i + 1) * n] for i in range(nw)]
    sig = [[idp[perm[i * n + j] // n][perm[i * n + j] % n] for j in range(n)] for i in range(nw)]
    return wires, idp, sig


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("nw", [5, 6])
def test_perm_product_closes_and_c_matches_bigint(cv, nw):
    r = cv.fr.p
    log_n = 4
    n = 1 << log_n
    wires, idp, sig = copy_constraint_system(cv, log_n, nw, 77 + nw)
    beta, gamma = 1234567, 7654321
    z = R.perm_product(cv, wires, idp, sig, beta, gamma)
    # the last row's ratio brings the product back to 1 (the permutation argument)
    a = b = 1
    for i in range(nw):
        t = wires[i][n - 1] + gamma
        a = a * (t + beta * idp[i][n - 1]) % r
        b = b * (t + beta * sig[i][n - 1]) % r
    assert z[n - 1] * a % r == b % r
    # a broken copy constraint does not close
    bad = [list(w) for w in wires]
    bad[0][3] = (bad[0][3] + 1) % r
    zb = R.perm_product(cv, bad, idp, sig, beta, gamma)
    t0 = [(bad[i][n - 1] + gamma) % r for i in range(nw)]
    ab = bb = 1
    for i in range(nw):
        ab = ab * (t0[i] + beta * idp[i][n - 1]) % r
        bb = bb * (t0[i] + beta * sig[i][n - 1]) % r
    assert zb[n - 1] * ab % r != bb % r
    M = lambda rows: np.array([mont_arr(cv.fr, x) for x in rows])
    got = C.plonk_perm_product(cv.curve_id, log_n, M(wires), M(idp), M(sig), mont_arr(cv.fr, [beta])[0],
                               mont_arr(cv.fr, [gamma])[0])
    assert from_mont_arr(cv.fr, got) == z


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_lookup_product_closes_and_c_matches_bigint(cv):
    r = cv.fr.p
    log_n = 4
    n = 1 << log_n
    rng = R.SplitMix64(31)
    range_size = n // 2
    table = list(range(range_size)) + [0] * (n - range_size)
    f = [rng.next() % range_size for _ in range(n)]
    cnt = {}
    for v in f[: n - 1]:
        cnt[v] = cnt.get(v, 0) + 1
    s = []
    for v in table:  # compute_lookup_sorted_vec_polynomials, constraint_system.rs:1597-1620
        s += [v] * (1 + cnt.pop(v)) if v in cnt else [v]
    assert len(s) == 2 * n - 1
    beta, gamma = 424242, 171717
    p = R.lookup_product(cv, f, table, s, beta, gamma)
    b1, gb = (1 + beta) % r, gamma * (1 + beta) % r
    j = n - 2
    a = b1 * (gamma + f[j]) % r * (gb + table[j] + beta * table[j + 1]) % r
    b = (gb + s[j] + beta * s[j + 1]) % r * (gb + s[n - 1 + j] + beta * s[n + j]) % r
    assert p[n - 2] * a % r == b % r and p[n - 1] == 1
    got = C.plonk_lookup_product(cv.curve_id, log_n, mont_arr(cv.fr, f), mont_arr(cv.fr, table), mont_arr(cv.fr, s),
                                 mont_arr(cv.fr, [beta])[0], mont_arr(cv.fr, [gamma])[0])
    assert from_mont_arr(cv.fr, got) == p


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n,nw", [(0, 5), (1, 5), (3, 6), (5, 5), (10, 6), (13, 5)])
def test_perm_product_gpu_matches_oracle(hip, cv, log_n, nw):
    import torch

    n = 1 << log_n
    w, idp, sig = (rand_fr_mont(cv, nw * n, 70 + k + log_n).reshape(nw, n, 4) for k in range(3))
    bg = rand_fr_mont(cv, 2, 99)
    dev = [torch.from_numpy(x.view(np.int64)).cuda() for x in (w, idp, sig)]
    out = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    hip.plonk_perm_product_device(cv.curve_id, log_n, nw, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                                  bg[0], bg[1], out.data_ptr(), stream=st)
    st.sync()
    exp = C.plonk_perm_product(cv.curve_id, log_n, w, idp, sig, bg[0], bg[1])
    assert (out.cpu().numpy().view(np.uint64).reshape(n, 4) == exp).all()
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [1, 2, 5, 10, 13])
def test_lookup_product_gpu_matches_oracle(hip, cv, log_n):
    import torch

    n = 1 << log_n
    f, t = (rand_fr_mont(cv, n, 80 + k + log_n) for k in range(2))
    s = rand_fr_mont(cv, 2 * n - 1, 83 + log_n)
    bg = rand_fr_mont(cv, 2, 98)
    dev = [torch.from_numpy(x.view(np.int64).reshape(-1)).cuda() for x in (f, t, s)]
    out = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    hip.plonk_lookup_product_device(cv.curve_id, log_n, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                                    bg[0], bg[1], out.data_ptr(), stream=st)
    st.sync()
    exp = C.plonk_lookup_product(cv.curve_id, log_n, f, t, s, bg[0], bg[1])
    assert (out.cpu().numpy().view(np.uint64).reshape(n, 4) == exp).all()
    st.close()
