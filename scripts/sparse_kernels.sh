#!/bin/bash
# per-kernel times of one sparse MSM case: usage scripts/sparse_kernels.sh <curve 0|1> "<case name>"
set -o pipefail
export TMPDIR=/tmp
export JF_CURVE=$1 JF_CASE="$2"
rm -rf gpurun_out/sk
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/sk -o p -- python3 scripts/sparse_msm_bench.py > gpurun_out/sk.log 2>&1 || { tail -5 gpurun_out/sk.log; exit 1; }
f=$(find gpurun_out/sk -name '*.db' | head -1)
python3 - "$f" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
last = max(i for i, r in enumerate(rows) if "k_msm_coarse<" in r[0] and (", 0>" in r[0] or ", 2>" in r[0]))
t0 = rows[last][1]
for nm, s, e in rows[last:]:
    print("%8.3f ms  +%7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, nm[:90]))
PY
rm -rf gpurun_out/sk
