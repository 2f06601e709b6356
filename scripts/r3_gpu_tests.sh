#!/bin/bash
# the whole GPU suite, log under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -60 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -4 gpurun_out/r3_gpu_tests.log
