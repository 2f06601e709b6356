#!/bin/bash
# One TurboPlonk proof (2^K gates, BN254) under rocprofv3 --kernel-trace --marker-trace with the library's roctx ranges
# on (JFHIP_MARKERS=1): per-round device time from the markers -> gpurun_out/prove_trace_summary.json
set -o pipefail
export TMPDIR=/tmp JFHIP_MARKERS=1
K=${1:-20}
rm -rf gpurun_out/pt
timeout -k 10 400 rocprofv3 --kernel-trace --marker-trace -d gpurun_out/pt -o p -- python3 scripts/plonk_prove_run.py $K 3 > gpurun_out/pt.log 2>&1 || { tail -20 gpurun_out/pt.log; exit 1; }
tail -4 gpurun_out/pt.log
f=$(find gpurun_out/pt -name '*.db' | head -1)
python3 scripts/prove_trace_summary.py "$f" gpurun_out/prove_trace_summary.json $K || exit 2
rm -rf gpurun_out/pt
