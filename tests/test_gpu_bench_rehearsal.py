"""bench.py's N > 1 code path on the one-GPU box: two ranks sharing cuda:0 over gloo (JF_BENCH_REHEARSE=1), run as a
child process exactly as the driver starts it (`python bench.py --gpus N ...`).  The headline must be the metric's MSM
-- 2^log_n points IN TOTAL, partitioned over the ranks by point range (strong scaling) -- and the group element must be
the one the single-GPU run computes.  RCCL itself needs two devices: tests/test_gpu_nccl_smoke.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG_N = 18


def _bench(gpus: int) -> dict:
    env = dict(os.environ, JF_BENCH_REHEARSE="1", JF_BENCH_SECONDARY_TIMEOUT_S="240")
    # N = 1: the headline only (the secondary single-GPU sections are the driver's full run); N = 2: headline + the
    # weak-scaling extra key, no proofs
    extra = ["--headline-only"] if gpus == 1 else ["--no-cpu-baseline", "--no-ntt", "--no-prove", "--ultra-log-n", "0"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--log-n", str(LOG_N), "--steps", "2",
           "--warmup", "1"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    if p.returncode != 0:   # keep the whole output where a GPU-box run brings it back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_rehearsal_%d.err" % gpus), "w") as f:
            f.write(p.stdout + "\n==== stderr ====\n" + p.stderr)
    assert p.returncode == 0, p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_rehearsal_reports_the_partitioned_msm():
    one = _bench(1)
    two = _bench(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for line in (one, two):
        assert line["config"]["total_points"] == 1 << LOG_N
        assert line["scaling"] == "strong" and line["unit"] == "Mpts/s" and line["value"] > 0
    assert two["config"]["points_per_gpu"] == 1 << (LOG_N - 1)
    # the partitioned MSM is the same group element
    assert two["result_sha"] == one["result_sha"]
    # the former headline (2^log_n points per rank) stays as an extra key
    assert two["weak_scaling"]["total_points"] == 2 << LOG_N and two["weak_scaling"]["value"] > 0
