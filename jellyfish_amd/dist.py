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

    def partial(self, scalars: torch.Tensor) -> torch.Tensor:
        """this rank's partial sum over its resident point range -> uint8[partial_bytes]"""

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

    def partial(self, scalars: torch.Tensor) -> torch.Tensor:
        assert scalars.is_cuda and scalars.is_contiguous()
        n = scalars.numel() * scalars.element_size() // 32
        out = torch.empty(self.partial_bytes, dtype=torch.uint8, device="cuda")
        self.lib.msm_partial_device(self.srs, scalars.data_ptr(), n, out.data_ptr(), stream=self.stream)
        return out

    def fold(self, partials: torch.Tensor, count: int) -> torch.Tensor:
        self.lib.g1_sum_partials(self.curve, partials.data_ptr(), count, self._out.data_ptr(), stream=self.stream)
        return self._out[: 2 * self.fq_bytes + 4]


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
    gathered = torch.empty(world * engine.partial_bytes, dtype=torch.uint8, device=part.device)
    dist.all_gather_into_tensor(gathered, part, group=group)
    return engine.fold(gathered, world)


def round_robin(items: Sequence, rank: int, world: int) -> list:
    """independent units (polynomials to transform / commit, circuits to prove): unit i -> rank i % world"""
    return [x for i, x in enumerate(items) if i % world == rank]
