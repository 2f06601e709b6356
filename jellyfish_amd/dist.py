"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device (rayon threads only); what shards on an
8 x MI355X node is:

* one MSM: the point range [k*N/G, (k+1)*N/G) and the matching scalars live on GPU k; each GPU
  produces ONE un-normalised partial sum (XYZZ, 128 B BN254 / 192 B BLS12-381); the only
  exchange is an all_gather of those G group elements, then every rank folds them locally
  (RCCL's sum cannot express elliptic-curve addition) and normalises to affine;
* batches of independent polynomial NTTs / commits: round-robin over ranks, no collective.

The arithmetic engine is passed in, so the same sharding logic runs on the HIP library
(product) and, in the CPU world_size-2 gloo tests, on a checker engine supplied by the tests.
"""
from __future__ import annotations

from typing import Protocol, Sequence

import torch
import torch.distributed as dist


class MsmEngine(Protocol):
    partial_bytes: int

    def partial(self, scalars: torch.Tensor, offset: int = 0) -> torch.Tensor:
        """partial sum of scalars[i] * P[offset + i] over this rank's resident points -> uint8[partial_bytes]"""

    def fold(self, partials: torch.Tensor, count: int) -> torch.Tensor:
        """sum `count` gathered partials and normalise -> uint8 packed affine + 4-byte flag"""


class HipMsmEngine:
    """libjfhip on the current CUDA(HIP) device; kernels run on torch's current stream."""

    def __init__(self, lib, srs, stream=None):
        self.lib = lib
        self.srs = srs
        self.curve = srs.curve
        self.fq_bytes = 8 * lib.FQ_LIMBS64[self.curve]
        self.partial_bytes = 4 * self.fq_bytes
        self.stream = stream or lib.Stream(wrap_raw=torch.cuda.current_stream().cuda_stream)
        self._out = torch.zeros(2 * self.fq_bytes + 16, dtype=torch.uint8, device="cuda")

    def partial(self, scalars: torch.Tensor, offset: int = 0) -> torch.Tensor:
        assert scalars.is_cuda and scalars.is_contiguous()
        n = scalars.numel() * scalars.element_size() // 32
        out = torch.empty(self.partial_bytes, dtype=torch.uint8, device="cuda")
        self.lib.msm_partial_device(self.srs, scalars.data_ptr(), n, out.data_ptr(), offset=offset, stream=self.stream)
        return out

    def partial_batch(self, scalars: Sequence[torch.Tensor], offsets: Sequence[int]) -> torch.Tensor:
        """the partial sums of a commit round in one shared pipeline (jfhip_msm_batch_partial_device)
        -> uint8[len(scalars) * partial_bytes]"""
        lens = [s.numel() * s.element_size() // 32 for s in scalars]
        out = torch.empty(len(scalars) * self.partial_bytes, dtype=torch.uint8, device="cuda")
        self.lib.msm_batch_partial_device(self.srs, [s.data_ptr() for s in scalars], lens, out.data_ptr(),
                                          offsets=list(offsets), stream=self.stream)
        return out

    def fold(self, partials: torch.Tensor, count: int) -> torch.Tensor:
        self.lib.g1_sum_partials(self.curve, partials.data_ptr(), count, self._out.data_ptr(), stream=self.stream)
        return self._out[: 2 * self.fq_bytes + 4]


def _all_gather_bytes(mine: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """all_gather of a small uint8 tensor.  RCCL ("nccl") gathers device tensors in place; the gloo
    backend (CPU tests, single-GPU rehearsals) is given host tensors."""
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        host = mine.cpu()
        out = torch.empty(world * host.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(mine.device)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous point range of `rank` (first ranks take the remainder)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_msm(engine: MsmEngine, local_scalars: torch.Tensor, group=None) -> torch.Tensor:
    """MSM whose point range is partitioned over the ranks of `group`.

    Every rank passes the scalars of ITS range (its SRS shard is resident in `engine`) and gets
    the full result: packed affine point + u32 identity flag, as uint8.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    part = engine.partial(local_scalars)
    if world == 1:
        return engine.fold(part, 1)
    return engine.fold(_all_gather_bytes(part, world, group), world)


def sharded_commit_batch(engine: MsmEngine, polys: Sequence[torch.Tensor], group=None) -> list:
    """P commitments (batch_commit, reference pcs/src/univariate_kzg/mod.rs:128-140) against an SRS that is
    REPLICATED on every rank: rank k takes the coefficient range shard_range(len, k, G) of every
    polynomial (so every GPU does 1/G of every MSM), and ONE all_gather carries the P partial sums of
    all ranks (P x 128 B per rank for BN254).  polys: flat int64 coefficient tensors (4 words per
    coefficient), identical on all ranks.  Returns P packed affine results (uint8 tensors)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    P = len(polys)
    if P == 0:
        return []
    ranges = [shard_range(poly.numel() // 4, rank, world) for poly in polys]
    if hasattr(engine, "partial_batch"):
        # the P partial sums of this rank in one sort / accumulate / fold pipeline
        mine = engine.partial_batch([poly[4 * lo:4 * hi] for poly, (lo, hi) in zip(polys, ranges)],
                                    [lo for lo, _ in ranges])
        parts = [mine[j * engine.partial_bytes:(j + 1) * engine.partial_bytes] for j in range(P)]
    else:
        parts = [engine.partial(poly[4 * lo:4 * hi], offset=lo) for poly, (lo, hi) in zip(polys, ranges)]
        mine = torch.cat(parts)
    if world == 1:
        return [engine.fold(parts[j], 1).clone() for j in range(P)]
    gathered = _all_gather_bytes(mine, world, group)
    per_poly = gathered.view(world, P, engine.partial_bytes).transpose(0, 1).contiguous()
    return [engine.fold(per_poly[j].reshape(-1), world).clone() for j in range(P)]


def round_robin(items: Sequence, rank: int, world: int) -> list:
    """independent units (polynomials to transform / commit, circuits to prove): unit i -> rank i % world"""
    return [x for i, x in enumerate(items) if i % world == rank]


def batch_prove_one_per_rank(hp, lib, group=None) -> dict:
    """batch_prove with one circuit per GPU (BASELINE config 5; reference
    plonk/src/proof_system/snark.rs:61-74 -> :163-430).

    `hp` is this rank's PlonkHotPath instance (its own circuit; its alpha_base challenge already carries
    the instance weight alpha^(3k) / alpha^(7k), prover.rs:644-655).  Every step of the aggregated proof
    that mixes instances is LINEAR in the instances -- the quotient sum, its split chunks, the
    linearisation polynomial and the two batched opening polynomials -- so each GPU commits its own
    contribution and the aggregated commitment is the plain group sum over ranks.  The only exchange is
    ONE all_gather of this rank's commitments (13 / 18 group elements); no polynomial crosses xGMI.

    Returns {"instance": uint8[world, n_inst, rec] per-instance commitments (wires, [h_1, h_2,] z,
    [prod_lookup]), "aggregated": uint8[W + 2, rec] (split-quotient commitments, opening, shifted opening)}."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    commits = hp.prove()
    rec, nc, W = hp.rec, hp.ncommit, hp.W
    n_inst = nc - (W + 2)                       # slots committed per instance; the last W + 2 slots aggregate
    if world == 1:
        allc = commits.view(1, nc, rec)
    else:
        allc = _all_gather_bytes(commits, world, group).view(world, nc, rec)
    fq = hp.fq_bytes
    fq_id = 1 if hp.curve == lib.BN254 else 3
    import numpy as np

    one_limbs = np.zeros(fq // 8, dtype=np.uint64)
    one_limbs[0] = 1
    one = torch.from_numpy(lib.field_op(fq_id, "to_mont", one_limbs).view(np.uint8)).to(allc.device)
    agg = torch.zeros(W + 2, rec, dtype=torch.uint8, device=allc.device)
    # affine (x, y, flag) -> XYZZ partial (x, y, 1, 1), identity -> zeros; then the same fold as a sharded MSM
    parts = torch.zeros(W + 2, world, 4 * fq, dtype=torch.uint8, device=allc.device)
    src = allc[:, n_inst:, :].transpose(0, 1)    # [slot][rank][rec]
    finite = (src[:, :, 2 * fq:2 * fq + 4].sum(dim=2, dtype=torch.int32) == 0).unsqueeze(-1)
    parts[:, :, :2 * fq] = src[:, :, :2 * fq]
    parts[:, :, 2 * fq:3 * fq] = one
    parts[:, :, 3 * fq:] = one
    parts = (parts * finite).contiguous()
    for j in range(W + 2):
        lib.g1_sum_partials(hp.curve, parts[j].data_ptr(), world, agg[j].data_ptr(), stream=hp.stream)
    return {"instance": allc[:, :n_inst, :], "aggregated": agg}
