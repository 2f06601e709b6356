#!/bin/bash
# round-3 check 2: new entry points and the bench rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_open.py tests/test_gpu_cpp_mirror.py tests/test_gpu_bench_rehearsal.py tests/test_gpu_nccl_smoke.py tests/test_dist.py tests/test_gpu_msm_fused.py -x -q --timeout 600 --timeout-method thread > gpurun_out/r3_c2_tests.log 2>&1 || { tail -60 gpurun_out/r3_c2_tests.log; exit 1; }
tail -5 gpurun_out/r3_c2_tests.log
