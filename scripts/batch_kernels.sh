#!/bin/bash
# per-kernel times of one fused batch_commit (P polynomials of 2^k points, BN254): rocprofv3 kernel trace over scripts/batch_probe.py
set -o pipefail
export TMPDIR=/tmp
K=${1:-20}; P=${2:-5}
timeout -k 10 200 python3 scripts/batch_probe.py $K $P > gpurun_out/bk_fused.log 2>&1 || { tail -5 gpurun_out/bk_fused.log; exit 1; }
cat gpurun_out/bk_fused.log
JFHIP_MSM_FUSED=0 timeout -k 10 200 python3 scripts/batch_probe.py $K $P > gpurun_out/bk_lanes.log 2>&1 || { tail -5 gpurun_out/bk_lanes.log; exit 1; }
cat gpurun_out/bk_lanes.log
rm -rf gpurun_out/bk
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/bk -o p -- python3 scripts/batch_probe.py $K $P > gpurun_out/bk.log 2>&1 || { tail -5 gpurun_out/bk.log; exit 1; }
f=$(find gpurun_out/bk -name '*.db' | head -1)
python3 - "$f" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last pipeline = kernels from the last k_msm_coarse<..., 2> (or 0) on
last = max(i for i, r in enumerate(rows) if "k_msm_coarse<" in r[0] and (", 0>" in r[0] or ", 2>" in r[0]))
t0 = rows[last][1]
for nm, s, e in rows[last:]:
    print("%8.3f ms  +%7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, nm[:100]))
PY
rm -rf gpurun_out/bk
