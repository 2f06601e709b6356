#!/bin/bash
# GPU-box round check: full GPU test suite, then bench.py under rocprofv3 --kernel-trace --stats; summaries under gpurun_out/
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err || { tail -20 gpurun_out/bench_prof.err; exit 2; }
tail -c 600 gpurun_out/bench_prof.json
find gpurun_out/prof -name '*.db' | while read f; do python3 scripts/rocpd_stats.py "$f" gpurun_out/bench_kernel_stats.csv; done
find gpurun_out/prof -name '*.db' -delete
ls -la gpurun_out/prof | head
