#!/bin/bash
# round-3 check 1: fused batch_commit parity, then prove timing with the fused pipeline on / off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_msm_fused.py tests/test_gpu_msm.py tests/test_gpu_plonk.py tests/test_gpu_dist_prove.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_c1_tests.log 2>&1 || { tail -40 gpurun_out/r3_c1_tests.log; exit 1; }
tail -3 gpurun_out/r3_c1_tests.log
timeout -k 10 200 python -u scripts/plonk_prove_run.py 20 5 > gpurun_out/r3_c1_prove_fused.log 2>&1 || { tail -20 gpurun_out/r3_c1_prove_fused.log; exit 2; }
tail -5 gpurun_out/r3_c1_prove_fused.log
JFHIP_MSM_FUSED=0 timeout -k 10 200 python -u scripts/plonk_prove_run.py 20 5 > gpurun_out/r3_c1_prove_lanes.log 2>&1 || { tail -20 gpurun_out/r3_c1_prove_lanes.log; exit 3; }
tail -5 gpurun_out/r3_c1_prove_lanes.log
