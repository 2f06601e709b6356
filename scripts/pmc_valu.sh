#!/bin/bash
# VALU / wait-state / cache PMC passes over scripts/pmc_run.py (2^24 BN254 MSM + 8 x 2^22 BLS12-381 NTT), one rocprofv3
# run per pass (program directly after `--`), summarised per kernel into gpurun_out/pmc_valu.json.
# usage: scripts/pmc_valu.sh [tag]
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-latest}
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
rm -rf gpurun_out/pmc_v1 gpurun_out/pmc_v2
timeout -s KILL 300 rocprofv3 --pmc $P1 -d gpurun_out/pmc_v1 -o v --output-format csv -- python3 scripts/pmc_run.py 24 > gpurun_out/pmc_v1.log 2>&1 || { tail -5 gpurun_out/pmc_v1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc $P2 -d gpurun_out/pmc_v2 -o v --output-format csv -- python3 scripts/pmc_run.py 24 > gpurun_out/pmc_v2.log 2>&1 || { tail -5 gpurun_out/pmc_v2.log; exit 2; }
A=$(find gpurun_out/pmc_v1 -name '*counter_collection.csv' | head -1)
B=$(find gpurun_out/pmc_v2 -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_valu_summary.py gpurun_out/pmc_valu_$TAG.json "$A" "$B"
rm -rf gpurun_out/pmc_v1 gpurun_out/pmc_v2
