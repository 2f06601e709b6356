#!/bin/bash
# PMC pass over the bucket-sort kernels of the 2^24 MSM (scripts/pmc_run.py with the NTT part skipped): instruction mix, LDS
# activity and wait states per kernel.  usage: scripts/pmc_sort.sh [tag]
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-sort}
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU"
rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2
timeout -s KILL 120 rocprofv3 --pmc $P1 -d gpurun_out/pmc_s1 -o v --output-format csv -- python3 scripts/pmc_run.py 23 > gpurun_out/pmc_s1.log 2>&1 || { tail -5 gpurun_out/pmc_s1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $P2 -d gpurun_out/pmc_s2 -o v --output-format csv -- python3 scripts/pmc_run.py 23 > gpurun_out/pmc_s2.log 2>&1 || { tail -5 gpurun_out/pmc_s2.log; exit 2; }
A=$(find gpurun_out/pmc_s1 -name '*counter_collection.csv' | head -1)
B=$(find gpurun_out/pmc_s2 -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_valu_summary.py gpurun_out/pmc_$TAG.json "$A" "$B" > /dev/null
python3 - gpurun_out/pmc_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    if "coarse" in k or "level" in k:
        print(k[:60])
        print("   ", {x: (round(y, 3) if y < 100 else int(y)) for x, y in v.items()})
PY
rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2
