#!/bin/bash
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) over scripts/pmc_run.py -> profiles-style summary
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 scripts/pmc_run.py 24 > gpurun_out/pmc_f.log 2>&1 || { tail -5 gpurun_out/pmc_f.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 scripts/pmc_run.py 24 > gpurun_out/pmc_w.log 2>&1 || { tail -5 gpurun_out/pmc_w.log; exit 2; }
F=$(find gpurun_out/pmc_f -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmc_w -name '*counter_collection.csv' | head -1)
echo "$F $W"
python3 scripts/pmc_summary.py "$F" "$W" gpurun_out/pmc_latest.json
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
