"""GPU parity for the fused batch_commit pipeline (jfhip_msm_batch_device / jfhip_msm_batch_partial_device: the commits of
a round share one sort, one bucket accumulation and one tail; reference pcs/src/univariate_kzg/mod.rs:128-140 and the
GPU twin :731-746).  Every commitment is compared with the C oracle's MSM of the same polynomial, bit for bit, and with
the per-polynomial path (JFHIP_MSM_FUSED=0 is the same code as jfhip_msm_device)."""
import numpy as np
import pytest
import torch

from oracle import coracle as C
from oracle import pyref as R
from tests.util import golden, gpt, h2i, mont_arr, pt_arr, rand_fr_mont

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


def _srs(hip, cv, n, window_bits=0):
    g = golden("msm")[cv.name]
    beta, gen = h2i(g["beta"]), gpt(g["g"])
    host = C.srs_gen(cv.curve_id, mont_arr(cv.fr, [beta])[0], pt_arr(cv, gen), n)
    return host, hip.Srs.load(cv.curve_id, host, window_bits=window_bits)


def _polys(cv, lens, seed):
    one = mont_arr(cv.fr, [1])[0]
    out = []
    for k, n in enumerate(lens):
        kind = k % 5
        if kind == 3 and n:          # all coefficients equal: one bucket per window takes everything
            p = np.tile(mont_arr(cv.fr, [cv.fr.p - 1])[0], (n, 1))
        elif kind == 4 and n > 8:    # a blinded unused wire: a few non-zero coefficients among zeros
            p = np.zeros((n, 4), dtype=np.uint64)
            p[[0, 1, n - 2, n - 1]] = rand_fr_mont(cv, 4, seed + k)
            p[n // 2] = one
        else:
            p = rand_fr_mont(cv, n, seed + k)
        out.append(p)
    return out


def _batch_device(hip, srs, cv, polys):
    rec = (16 * cv.fq.limbs64 + 4 + 15) // 16 * 16
    dev = [torch.from_numpy(p.view(np.int64).reshape(-1).copy()).cuda() if len(p) else torch.zeros(4, dtype=torch.int64, device="cuda") for p in polys]
    out = torch.zeros(len(polys) * rec, dtype=torch.uint8, device="cuda")
    st = hip.Stream(wrap_raw=torch.cuda.current_stream().cuda_stream)
    hip.msm_batch_device(srs, [d.data_ptr() for d in dev], [len(p) for p in polys], out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    raw = out.cpu().numpy().reshape(len(polys), rec)
    pts = raw[:, :16 * cv.fq.limbs64].copy().view(np.uint64)
    inf = raw[:, 16 * cv.fq.limbs64:16 * cv.fq.limbs64 + 4].copy().view(np.uint32).reshape(-1) != 0
    return pts, inf


def _check(cv, host, polys, pts, inf):
    for i, p in enumerate(polys):
        exp = C.msm(cv.curve_id, host[:len(p)], p) if len(p) else np.zeros(2 * cv.fq.limbs64, dtype=np.uint64)
        assert (pts[i] == exp).all(), i
        assert inf[i] == (not exp.any()), i


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("window_bits", [0, 16], ids=["auto_c", "c16_staged"])
def test_fused_batch_matches_oracle(hip, cv, window_bits):
    """ragged batches of 2 .. 17 polynomials (17 = two pipelines), with empty, constant-coefficient and sparse members;
    window_bits 16 gives <= 16 windows (the staged coarse scatter), the automatic choice at this size more (direct)"""
    n = (1 << 12) + 8
    host, srs = _srs(hip, cv, n, window_bits)
    for P, seed in ((2, 10), (5, 20), (16, 30), (17, 40)):
        lens = [[n, n - 1, 1, 0, 3000, 17, n - 5][k % 7] for k in range(P)]
        polys = _polys(cv, lens, seed)
        pts, inf = _batch_device(hip, srs, cv, polys)
        _check(cv, host, polys, pts, inf)
    # a batch of only empty / zero polynomials
    polys = [np.zeros((0, 4), dtype=np.uint64), np.zeros((64, 4), dtype=np.uint64)]
    pts, inf = _batch_device(hip, srs, cv, polys)
    assert inf.all() and not pts.any()
    srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_fused_batch_equals_per_polynomial_path_at_2_16(hip, cv):
    """a commit round of the prover's shape (5 polynomials of n + 2 .. n + 3 coefficients) at 2^16: the fused pipeline
    against jfhip_msm_device per polynomial and the oracle"""
    n = (1 << 16) + 3
    host, srs = _srs(hip, cv, n)
    polys = _polys(cv, [n, n - 1, n - 1, n, n - 1], 77)
    pts, inf = _batch_device(hip, srs, cv, polys)
    rec = (16 * cv.fq.limbs64 + 4 + 15) // 16 * 16
    st = hip.Stream(wrap_raw=torch.cuda.current_stream().cuda_stream)
    for i, p in enumerate(polys):
        d = torch.from_numpy(p.view(np.int64).reshape(-1).copy()).cuda()
        out = torch.zeros(rec, dtype=torch.uint8, device="cuda")
        hip.msm_device(srs, d.data_ptr(), len(p), out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        single = out.cpu().numpy()[:16 * cv.fq.limbs64].copy().view(np.uint64)
        assert (single == pts[i]).all(), i
    _check(cv, host, polys, pts, inf)   # member 3 has all-equal coefficients: its buckets go through the chunked heavy fix-up
    srs.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_batch_partial_with_offsets(hip, cv):
    """the multi-GPU leg of a commit round: every polynomial's coefficient range [lo, hi) as an un-normalised partial sum
    from ONE pipeline; two ranges per polynomial must add up to the whole commitment"""
    n = (1 << 12) + 8
    host, srs = _srs(hip, cv, n)
    polys = _polys(cv, [n, n - 3, 100], 5)
    xz = 32 * cv.fq.limbs64
    rec = 16 * cv.fq.limbs64 + 16
    st = hip.Stream(wrap_raw=torch.cuda.current_stream().cuda_stream)
    dev = [torch.from_numpy(p.view(np.int64).reshape(-1).copy()).cuda() for p in polys]
    parts = torch.zeros(2, len(polys), xz, dtype=torch.uint8, device="cuda")
    for half in range(2):
        los = [0 if half == 0 else len(p) // 3 for p in polys]
        his = [len(p) // 3 if half == 0 else len(p) for p in polys]
        hip.msm_batch_partial_device(srs, [d.data_ptr() + 32 * lo for d, lo in zip(dev, los)],
                                     [hi - lo for lo, hi in zip(los, his)], parts[half].data_ptr(), offsets=los, stream=st)
    out = torch.zeros(rec, dtype=torch.uint8, device="cuda")
    for i, p in enumerate(polys):
        both = parts[:, i, :].contiguous()
        hip.g1_sum_partials(cv.curve_id, both.data_ptr(), 2, out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        got = out.cpu().numpy()[:16 * cv.fq.limbs64].copy().view(np.uint64)
        assert (got == C.msm(cv.curve_id, host[:len(p)], p)).all(), i
    srs.free()
