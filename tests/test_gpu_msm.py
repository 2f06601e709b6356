"""GPU parity for the MSM path (UnivariateKzgPCS::commit / batch_commit, SRS loading) through the
C ABI, following the protocol of the reference's test_gpu_e2e
(pcs/src/univariate_kzg/mod.rs:1359-1414): GPU commitment == CPU commitment, as affine points,
bit for bit."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import golden, gpt, h2i, mont_arr, pt_arr, pt_from, rand_fr_mont, seeded_scalars

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


@pytest.fixture(scope="module", params=CURVES, ids=lambda c: c.name)
def setup(request, hip):
    cv = request.param
    g = golden("msm")[cv.name]
    beta, gen = h2i(g["beta"]), gpt(g["g"])
    n = (1 << 12) + 8
    srs_host = C.srs_gen(cv.curve_id, mont_arr(cv.fr, [beta])[0], pt_arr(cv, gen), n)
    srs = hip.Srs.load(cv.curve_id, srs_host)
    yield cv, g, srs_host, srs
    srs.free()


def test_srs_gen_matches_oracle_and_golden(hip, setup):
    cv, g, srs_host, _ = setup
    beta, gen = h2i(g["beta"]), gpt(g["g"])
    dev = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [beta])[0], pt_arr(cv, gen), len(srs_host))
    got = dev.read(0, len(srs_host))
    assert (got == srs_host).all()
    for i, p in g["srs_spot"].items():
        assert pt_from(cv, got[int(i)]) == gpt(p)
    dev.free()


def test_msm_golden_cases(hip, setup):
    cv, g, srs_host, srs = setup
    for case in g["cases"]:
        n, off = case["n"], case["offset"]
        sc = [h2i(s) for s in case["scalars"]] if "scalars" in case else seeded_scalars(cv, case["scalar_seed"], n)
        out, inf = hip.msm(srs, mont_arr(cv.fr, sc), offset=off)
        exp = gpt(case["expect"])
        assert pt_from(cv, out) == exp, (n, case["set"])
        assert inf == (exp is None)


@pytest.mark.parametrize("c", [4, 7, 13])
def test_msm_window_sizes_and_arbitrary_bases(hip, setup, c):
    """exceptional additions (P+P, P-P, identity bases) at several window sizes"""
    cv, g, _, _ = setup
    bases = np.array([pt_arr(cv, gpt(p)) for p in g["arbitrary_bases"]])
    srs = hip.Srs.load(cv.curve_id, bases, window_bits=c)
    assert srs.window_bits == c
    for case in g["arbitrary_cases"]:
        sc = mont_arr(cv.fr, [h2i(s) for s in case["scalars"]])
        out, inf = hip.msm(srs, sc)
        assert pt_from(cv, out) == gpt(case["expect"]), (c, case["set"])
    srs.free()


def test_srs_load_strided_with_infinity_flag(hip, setup):
    """arkworks' in-memory G1Affine {x, y, infinity: bool} records (stride 72 / 104 bytes)"""
    cv, g, srs_host, _ = setup
    n = 50
    pt_bytes = 16 * cv.fq.limbs64
    stride = pt_bytes + 8
    raw = np.zeros((n, stride), dtype=np.uint8)
    raw[:, :pt_bytes] = srs_host[:n].view(np.uint8).reshape(n, pt_bytes)
    raw[7, pt_bytes] = 1  # flagged identity with stale coordinates
    srs = hip.Srs.load(cv.curve_id, raw, stride_bytes=stride)
    sc = rand_fr_mont(cv, n, 3)
    bases = srs_host[:n].copy()
    bases[7] = 0
    out, _ = hip.msm(srs, sc)
    assert (out == C.msm(cv.curve_id, bases, sc)).all()
    srs.free()


def test_gpu_e2e_protocol(hip, setup):
    """10 rounds of random degree < 1025 (reference :1371-1387), then batches of 10..14 (:1389-1401),
    the empty batch (:1402-1405) and a ragged batch."""
    cv, _, srs_host, srs = setup
    rng = np.random.default_rng(17)
    for rnd in range(10):
        deg = int(rng.integers(1, 1025))
        sc = rand_fr_mont(cv, deg + 1, 100 + rnd)
        out, inf = hip.msm(srs, sc)
        assert (out == C.msm(cv.curve_id, srs_host[:deg + 1], sc)).all()
        assert not inf
    for bs in range(10, 15):
        polys = [rand_fr_mont(cv, 1025, 1000 + 20 * bs + i) for i in range(bs)]
        outs, infs = hip.msm_batch(srs, polys)
        for i, p in enumerate(polys):
            assert (outs[i] == C.msm(cv.curve_id, srs_host[:1025], p)).all()
    outs, infs = hip.msm_batch(srs, [])
    assert outs.shape[0] == 0
    polys = [rand_fr_mont(cv, n, 4000 + n) for n in (1, 17, 1000)] + [np.zeros((0, 4), dtype=np.uint64)]
    outs, infs = hip.msm_batch(srs, polys)
    for i, p in enumerate(polys[:3]):
        assert (outs[i] == C.msm(cv.curve_id, srs_host[:len(p)], p)).all()
    assert infs[3] and not outs[3].any()


def test_degenerate_inputs(hip, setup):
    cv, _, srs_host, srs = setup
    # zero polynomial and empty polynomial -> identity
    out, inf = hip.msm(srs, np.zeros((64, 4), dtype=np.uint64))
    assert inf and not out.any()
    out, inf = hip.msm(srs, np.zeros((0, 4), dtype=np.uint64))
    assert inf and not out.any()
    # all coefficients equal (one bucket per window receives everything)
    n = 3000
    one = mont_arr(cv.fr, [1])[0]
    for val in (one, mont_arr(cv.fr, [cv.fr.p - 1])[0], mont_arr(cv.fr, [0xFFFFFFFFFFFF])[0]):
        sc = np.tile(val, (n, 1))
        out, inf = hip.msm(srs, sc)
        assert (out == C.msm(cv.curve_id, srs_host[:n], sc)).all()
    # sparse 0/1 witness-like vector
    sc = np.zeros((n, 4), dtype=np.uint64)
    idx = np.random.default_rng(3).integers(0, n, size=n // 3)
    sc[idx] = one
    out, _ = hip.msm(srs, sc)
    assert (out == C.msm(cv.curve_id, srs_host[:n], sc)).all()
    # degree check: "poly degree {} is larger than allowed {}" (mod.rs:100-106)
    with pytest.raises(hip.JfHipError) as e:
        hip.msm(srs, rand_fr_mont(cv, len(srs_host) + 1, 1))
    assert e.value.code == -1 and "larger than allowed" in e.value.msg


@pytest.mark.parametrize("log_n", [16, 18])
def test_msm_vs_oracle_large(hip, log_n):
    """BN254 at 2^16 / 2^18: full comparison with the C oracle; SRS generated on the device."""
    cv = R.BN254
    n = (1 << log_n) + 1
    beta = mont_arr(cv.fr, [0xBEEF0000 + log_n])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    srs_host = srs.read(0, n)
    # spot-check the device SRS against [beta^i] g from the oracle
    for i in (0, 1, n // 2, n - 1):
        bi = C.fp_unop(0, "to_mont", np.array(cv.fr.limbs(pow(0xBEEF0000 + log_n, i, cv.fr.p)), dtype=np.uint64))
        assert (srs_host[i] == C.ec_mul(cv.curve_id, pt_arr(cv, cv.g), bi)).all()
    sc = rand_fr_mont(cv, n, 11)
    out, inf = hip.msm(srs, sc)
    assert (out == C.msm(cv.curve_id, srs_host, sc)).all()
    srs.free()


def test_msm_full_size_closed_form(hip):
    """BASELINE config 2 size (2^20 BN254): commit == [p(beta)] g, a size-independent identity,
    plus linearity commit(a) + commit(b) == commit(a + b)."""
    cv = R.BN254
    n = 1 << 20
    beta = mont_arr(cv.fr, [0x6A660001])[0]
    G = pt_arr(cv, cv.g)
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, G, n)
    a = rand_fr_mont(cv, n, 21)
    b = rand_fr_mont(cv, n, 22)
    ca, _ = hip.msm(srs, a)
    cb, _ = hip.msm(srs, b)
    pa = C.poly_eval(cv.curve_id, a, beta)
    assert (ca == C.ec_mul(cv.curve_id, G, pa)).all()
    s = C.fp_binop(0, "add", a, b)
    cs, _ = hip.msm(srs, s)
    assert (cs == C.ec_add(cv.curve_id, ca, cb)).all()
    srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_short_msm_on_a_large_srs(hip, cv):
    """Commitments much shorter than the SRS (a small circuit against a universal SRS) run on a window table over a
    prefix of the points, built on first use (csrc/api.hip srs_route): lengths below and above the n / 16 threshold,
    both prefix levels (4096, 16384), an offset slice, and the identity, against the oracle."""
    n = (1 << 18) + 5
    beta = mont_arr(cv.fr, [0x51A0 + cv.curve_id])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    host = srs.read(0, 21000)
    for rnd, (off, m) in enumerate([(0, 1), (0, 700), (0, 4096), (0, 5000), (100, 3000), (0, 16384), (0, 20000),
                                    (0, 700)]):
        sc = rand_fr_mont(cv, m, 900 + rnd)
        out, inf = hip.msm(srs, sc, offset=off)
        assert (out == C.msm(cv.curve_id, host[off:off + m], sc)).all(), (off, m)
        assert not inf
    out, inf = hip.msm(srs, np.zeros((300, 4), dtype=np.uint64))
    assert inf and not out.any()
    polys = [rand_fr_mont(cv, m, 950 + m) for m in (33, 1025, 4097)]
    outs, _ = hip.msm_batch(srs, polys)
    for i, pl in enumerate(polys):
        assert (outs[i] == C.msm(cv.curve_id, host[:len(pl)], pl)).all()
    srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("limbs29", ["1", "0"], ids=["limbs29", "saturated"])
def test_msm_skewed_scalars_both_accumulators(hip, cv, limbs29, monkeypatch):
    """Scalar distributions that put most entries into a few buckets (all equal, 0/1 witnesses, one heavy value among
    random ones, r - 1): buckets spanning many lane slices go through k_msm_fixup_heavy.  Both bucket-accumulation
    kernels (29-bit limbs and saturated limbs, chosen when the SRS is loaded) against the C oracle, bit for bit."""
    monkeypatch.setenv("JFHIP_MSM_F29", limbs29)
    n = (1 << 15) + 3
    beta = mont_arr(cv.fr, [0xFACE + cv.curve_id])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    srs_host = srs.read(0, n)
    rnd = rand_fr_mont(cv, n, 77)
    same = np.repeat(rnd[:1], n, axis=0)
    bits = mont_arr(cv.fr, [0, 1])[np.arange(n) % 3 % 2]
    heavy = rnd.copy()
    heavy[::2] = rnd[5]
    minus1 = np.repeat(mont_arr(cv.fr, [cv.fr.p - 1]), n, axis=0)
    for sc in (same, bits, heavy, minus1):
        out, inf = hip.msm(srs, np.ascontiguousarray(sc))
        exp = C.msm(cv.curve_id, srs_host, np.ascontiguousarray(sc))
        assert (out == exp).all() and inf == (0 if exp.any() else 1)
    srs.free()


@pytest.mark.parametrize("f29", ["0", "1"])
def test_msm_sparse_scalars_skip_empty_buckets(hip, f29, monkeypatch):
    """A blinded all-zero wire polynomial (prover.rs:457-466 on an unused wire): four non-zero coefficients among
    2^17 + 2, so almost every bucket is empty between two sorted entries.  Same point as the oracle, for both
    accumulate kernels, and in a small fraction of a dense MSM's time (the empty buckets are searched, not walked:
    regression for 5.7 ms per such commitment at 2^20)."""
    import time

    cv = R.BN254
    n = (1 << 17) + 2
    beta = mont_arr(cv.fr, [0x5EED5EED])[0]
    monkeypatch.setenv("JFHIP_MSM_F29", f29)
    srs = hip.Srs.gen_for_testing(0, beta, pt_arr(cv, cv.g), n)
    bases = srs.read(0, n)
    dense = rand_fr_mont(cv, n, 77)
    sparse = np.zeros_like(dense)
    for i in (0, 1, n - 2, n - 1):
        sparse[i] = dense[i]
    got, inf = hip.msm(srs, sparse)
    idx = [0, 1, n - 2, n - 1]
    exp = C.msm(0, bases[idx], sparse[idx])
    assert (got == exp).all() and not inf

    # device-resident timing (no scalar upload in the timed region)
    import torch

    st = hip.Stream()
    out = torch.zeros(128, dtype=torch.uint8, device="cuda")

    def timed(sc_np):
        sc = torch.from_numpy(sc_np.view(np.int64)).cuda()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hip.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st)
            st.sync()
            ts.append(time.perf_counter() - t0)
        return min(ts[1:])

    t_dense, t_sparse = timed(dense), timed(sparse)
    assert t_sparse < 1.5 * t_dense, (t_sparse, t_dense)
    srs.free()


def _ser_uncompressed(cv, P) -> bytes:
    """ark-serialize uncompressed G1Affine (formats: include/jfhip.h, jfhip_srs_load_serialized)"""
    p = cv.fq.p
    nb = 8 * cv.fq.limbs64
    if cv.name == "bn254":
        if P is None:
            b = bytearray(2 * nb)
            b[-1] |= 0x40
            return bytes(b)
        b = bytearray(P[0].to_bytes(nb, "little") + P[1].to_bytes(nb, "little"))
        if P[1] > (p - P[1]) % p:
            b[-1] |= 0x80
        return bytes(b)
    if P is None:
        b = bytearray(2 * nb)
        b[0] |= 0x40
        return bytes(b)
    return P[0].to_bytes(nb, "big") + P[1].to_bytes(nb, "big")


def test_srs_load_serialized_decodes_on_the_device(hip, setup):
    """jfhip_srs_load_serialized: compressed and uncompressed ark-serialize records -> the same device SRS as loading the
    in-memory points; invalid records are refused with their index.  The byte formats are the restated ones of
    oracle/plonk_ref.py (no serialized G1 fixture exists in the reference: parity with arkworks' bytes is unpinned)."""
    from oracle import plonk_ref as PR

    cv, _, srs_host, _ = setup
    n = 700
    pts = [pt_from(cv, srs_host[i]) for i in range(n)]
    pts[5] = None
    pts[n - 1] = None
    want = srs_host[:n].copy()
    want[5] = 0
    want[n - 1] = 0
    for compressed, ser in ((True, lambda P: PR.ser_g1(cv, P)), (False, lambda P: _ser_uncompressed(cv, P))):
        blob = b"".join(ser(P) for P in pts)
        for validate in (True, False):
            s = hip.Srs.load_serialized(cv.curve_id, blob + b"\x00" * 7, n, compressed=compressed, validate=validate)
            assert len(s) == n
            assert (s.read(0, n) == want).all(), (compressed, validate)
            sc = rand_fr_mont(cv, n, 99)
            got, inf = hip.msm(s, sc)
            assert not inf and (got == C.msm(cv.curve_id, want, sc)).all()
            s.free()
        # too few bytes
        with pytest.raises(hip.JfHipError, match="records"):
            hip.Srs.load_serialized(cv.curve_id, blob[:-1], n, compressed=compressed)
    p, nb = cv.fq.p, 8 * cv.fq.limbs64
    b_coef = 3 if cv.name == "bn254" else 4
    order = "little" if cv.name == "bn254" else "big"

    def with_x(x, flag_compressed=True):
        b = bytearray(x.to_bytes(nb, order))
        if cv.name != "bn254":
            b[0] |= 0x80
        return bytes(b)

    good = PR.ser_g1(cv, pts[0])
    # x >= p
    with pytest.raises(hip.JfHipError, match="record 1 .*modulus"):
        hip.Srs.load_serialized(cv.curve_id, good + with_x(p) + good, 3)
    # x^3 + b is not a square
    x = 2
    while pow((x ** 3 + b_coef) % p, (p - 1) // 2, p) == 1:
        x += 1
    with pytest.raises(hip.JfHipError, match="record 2 .*curve"):
        hip.Srs.load_serialized(cv.curve_id, good + good + with_x(x), 3)
    # uncompressed point off the curve: caught only when validating
    bad = _ser_uncompressed(cv, (pts[0][0], (pts[0][1] + 1) % p))
    with pytest.raises(hip.JfHipError, match="record 0 .*curve"):
        hip.Srs.load_serialized(cv.curve_id, bad, 1, compressed=False, validate=True)
    hip.Srs.load_serialized(cv.curve_id, bad, 1, compressed=False, validate=False).free()
    if cv.name != "bn254":
        # on the curve but outside the prime-order subgroup (cofactor != 1): Validate::Yes refuses it
        x = 1
        while True:
            rhs = (x ** 3 + b_coef) % p
            if pow(rhs, (p - 1) // 2, p) == 1:
                y = pow(rhs, (p + 1) // 4, p)
                if R.ec_add(cv, R.ec_mul(cv, cv.fr.p - 1, (x, y)), (x, y)) is not None:
                    break
            x += 1
        with pytest.raises(hip.JfHipError, match="subgroup"):
            hip.Srs.load_serialized(cv.curve_id, with_x(x), 1, validate=True)
        s = hip.Srs.load_serialized(cv.curve_id, with_x(x), 1, validate=False)
        got = pt_from(cv, s.read(0, 1)[0])
        assert got[0] == x and got[1] in (y, p - y) and got[1] <= (p - got[1]) % p   # flag clear: the smaller root
        s.free()
        # compression flag missing
        with pytest.raises(hip.JfHipError, match="flags"):
            hip.Srs.load_serialized(cv.curve_id, pts[0][0].to_bytes(nb, "big"), 1)


@pytest.mark.parametrize("n", [0, 1, 2, 37, 300])
def test_msm_small_on_host_points(hip, setup, n):
    """jfhip_msm_small (verifier-side VariableBaseMSM over commitments): arbitrary bases, no SRS"""
    cv, _, srs_host, _ = setup
    rng = np.random.default_rng(n)
    idx = rng.integers(0, len(srs_host), size=n)
    pts = srs_host[idx].copy()
    if n > 5:
        pts[3] = 0          # an identity among the bases
        pts[4] = pts[2]     # a repeated base
    sc = rand_fr_mont(cv, n, 500 + n)
    got, inf = hip.msm_small(cv.curve_id, pts, sc)
    if n == 0:
        assert inf
        return
    exp = C.msm(cv.curve_id, pts, sc)
    assert (got == exp).all() and inf == (not exp.any())


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("limbs29", ["1", "0"], ids=["limbs29", "saturated"])
def test_msm_small_valued_scalars_fill_the_chip(hip, cv, limbs29, monkeypatch):
    """Witness-like scalars (the bench circuits' wires hold counters, booleans and constants: a few bits each) give non-zero
    digits in one or two of the ~15 windows, so the sorted entry list is a small fraction of n x windows.  The kernels
    after the sort derive the slice length from the entry count the sort leaves on the device (msm_slice_len), so every
    lane still gets an equal share: same points as the oracle, and a small-valued MSM must be much cheaper than a dense
    one of the same length (it took the dense time before: the first few lanes got full-length slices)."""
    import time

    import torch

    monkeypatch.setenv("JFHIP_MSM_F29", limbs29)   # read when the SRS is loaded: both accumulation kernels
    n = (1 << 18) + 3
    beta = mont_arr(cv.fr, [0x5EED0042])[0]
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    bases = srs.read(0, n)
    dense = rand_fr_mont(cv, n, 91)
    counter = mont_arr(cv.fr, list(range(n)))                        # w_i = i: two windows
    ones = np.repeat(mont_arr(cv.fr, [1]), n, axis=0)                # one bucket holds everything (heavy fix-up)
    bools = mont_arr(cv.fr, [0, 1])[(np.arange(n) * 7 // 3) % 2]      # half the entries vanish
    neg_small = mont_arr(cv.fr, [cv.fr.p - 1 - (i % 1000) for i in range(n)])   # -1 - k: small after the sign trick? (dense digits)
    for sc in (counter, ones, bools, neg_small):
        got, inf = hip.msm(srs, np.ascontiguousarray(sc))
        exp = C.msm(cv.curve_id, bases, np.ascontiguousarray(sc))
        assert (got == exp).all() and inf == (0 if exp.any() else 1)

    st = hip.Stream()
    out = torch.zeros(256, dtype=torch.uint8, device="cuda")

    def timed(sc_np):
        sc = torch.from_numpy(np.ascontiguousarray(sc_np).view(np.int64)).cuda()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hip.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st)
            st.sync()
            ts.append(time.perf_counter() - t0)
        return min(ts[1:])

    srs.free()
    if limbs29 == "0":
        return
    # timing property at the size of a prover's commitment (below ~2^18 both are bound by the latency of the tail)
    n = (1 << 20) + 3
    srs = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), n)
    t_dense = timed(rand_fr_mont(cv, n, 92))
    t_counter = timed(mont_arr(cv.fr, list(range(n))))
    assert t_counter < 0.75 * t_dense, (t_counter, t_dense)
    srs.free()
