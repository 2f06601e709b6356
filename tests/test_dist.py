"""CPU suite: the multi-GPU sharding layer (jellyfish_amd/dist.py) with world_size 2 over gloo.
The arithmetic engine on CPU is a checker engine built from the oracle (tests only); the sharding
logic (ranges, one-element all_gather, local fold) is the product code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jellyfish_amd import dist as jdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """checker engine: partial = packed affine sum of this rank's range, fold = affine additions"""

    def __init__(self, curve, bases):
        from oracle import coracle as C

        self.C, self.curve, self.bases = C, curve, bases
        self.partial_bytes = bases.shape[1] * 8

    def partial(self, scalars, offset=0):
        sc = scalars.numpy().view(np.uint64).reshape(-1, 4)
        out = self.C.msm(self.curve, self.bases[offset:offset + len(sc)], sc, threads=2)
        return torch.from_numpy(out.view(np.uint8).copy())

    def fold(self, partials, count):
        pts = partials.numpy().view(np.uint64).reshape(count, -1)
        acc = np.zeros_like(pts[0])
        for p in pts:
            acc = self.C.ec_add(self.curve, acc, p)
        inf = np.array([0 if acc.any() else 1], dtype=np.uint32)
        return torch.from_numpy(np.concatenate([acc.view(np.uint8), inf.view(np.uint8)]))


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _worker_body(rank, world, n, q)
    except Exception as e:  # report instead of leaving the parent waiting
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_body(rank, world, n, q):
    from oracle import coracle as C
    from oracle import pyref as R
    from tests.util import mont_arr, pt_arr, rand_fr_mont

    cv = R.BN254
    beta = mont_arr(cv.fr, [31337])[0]
    srs = C.srs_gen(0, beta, pt_arr(cv, cv.g), n, threads=2)
    sc = rand_fr_mont(cv, n, 5)
    lo, hi = jdist.shard_range(n, rank, world)
    eng = OracleEngine(0, srs[lo:hi])
    res = jdist.sharded_msm(eng, torch.from_numpy(sc[lo:hi].view(np.int64).copy()))
    full = C.msm(0, srs, sc, threads=2)
    ok = bool((res.numpy()[:64].view(np.uint64) == full).all()) and res.numpy()[64:68].view(np.uint32)[0] == 0
    # batch_commit with coefficient ranges sharded over the ranks (replicated SRS, ragged lengths)
    full_eng = OracleEngine(0, srs)
    lens = [n, n - 3, 5, 1]
    polys = [torch.from_numpy(sc[:l].view(np.int64).reshape(-1).copy()) for l in lens]
    outs = jdist.sharded_commit_batch(full_eng, polys)
    for l, o in zip(lens, outs):
        ok = ok and bool((o.numpy()[:64].view(np.uint64) == C.msm(0, srs[:l], sc[:l], threads=2)).all())
    # round-robin of independent units covers everything exactly once
    mine = jdist.round_robin(list(range(11)), rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    ok = ok and sorted(sum(gathered, [])) == list(range(11))
    # coset sharding of the quotient: this rank evaluates its slice of the coset, inverts the slice
    # transform, ONE all_gather, then the size-G DFT inverse per coefficient rebuilds the polynomial on every rank
    ok = ok and _coset_sharding_roundtrip(cv, 3, 6, world, rank, gather=True)
    q.put((rank, bool(ok)))


def _coset_sharding_roundtrip(cv, log_n, log_m, G, rank=None, gather=False):
    """t (degree < m) -> per-slice evaluations on {h w_m^(g + G j)} -> slice iNTT -> unshard_coeffs -> t"""
    from jellyfish_amd import hostglue as H
    from jellyfish_amd import plonk as JP
    from oracle import pyref as R

    cp = H.CURVES[cv.curve_id]
    r, m = cv.fr.p, 1 << log_m
    ml = m // G
    rng = R.SplitMix64(77 + G)
    t = [rng.field(cv.fr) for _ in range(m)]
    whole = R.fft(cv, t, log_m, cv.fr_generator)
    assert JP.coset_sharding_applies(1 << log_n, m, G)

    def slice_poly(g):
        off = JP.slice_offset(cp, m, g)
        # the slice of the whole coset IS the size-ml coset with this offset (of t reduced mod X^ml - off^ml)
        c = pow(off, ml, r)
        folded = [sum(t[k * ml + j] * pow(c, k, r) for k in range(G)) % r for j in range(ml)]
        assert R.fft(cv, folded, ml.bit_length() - 1, off) == whole[g::G]
        return R.ifft(cv, whole[g::G], ml.bit_length() - 1, off)

    if gather:
        mine = torch.tensor([[x >> (52 * k) & (2**52 - 1) for k in range(5)] for x in slice_poly(rank)], dtype=torch.int64)
        out = torch.empty(G * ml, 5, dtype=torch.int64)
        dist.all_gather_into_tensor(out, mine)
        out = out.view(G, ml, 5)
        polys = [[sum(int(w) << (52 * k) for k, w in enumerate(row)) for row in out[g].tolist()] for g in range(G)]
    else:
        polys = [slice_poly(g) for g in range(G)]
    coeffs = JP.unshard_coeffs(cp, m, G)
    back = []
    for c in range(G):
        back += [sum(coeffs[c][g] * polys[g][j] for g in range(G)) % r for j in range(ml)]
    return back == t


@pytest.mark.parametrize("G", [2, 4, 8])
def test_coset_sharding_algebra(G):
    from oracle import pyref as R

    for cv in (R.BN254, R.BLS12_381):
        assert _coset_sharding_roundtrip(cv, 2, 5, G)


def test_shard_range_partitions():
    for n in (0, 1, 7, 1024, 1000003):
        for world in (1, 2, 3, 8):
            spans = [jdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_msm_world2_gloo():
    world, n = 2, 3001
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok is True for _, ok in results), results
