"""RCCL ("nccl" backend) smoke test of the sharded MSM: one rank per GPU, the exchange of jellyfish_amd/dist.py (an
all_gather of one XYZZ element per rank) over the real backend.  Needs two devices: skipped on a one-GPU box, where
tests/test_gpu_bench_rehearsal.py and tests/test_gpu_dist_prove.py cover the same code over gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_over_rccl():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs two devices (one rank per GPU)")
    env = dict(os.environ, JF_BENCH_SECONDARY_TIMEOUT_S="240")
    env.pop("JF_BENCH_REHEARSE", None)
    outs = {}
    for gpus in (1, 2):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--log-n", "18", "--steps", "2",
               "--warmup", "1", "--headline-only"]
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        outs[gpus] = json.loads(lines[0])
    assert outs[2]["n_gpus"] == 2 and outs[2]["scaling"] == "strong"
    assert outs[2]["config"]["total_points"] == 1 << 18
    assert outs[2]["result_sha"] == outs[1]["result_sha"]
