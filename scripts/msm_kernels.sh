#!/bin/bash
# per-kernel times of the 2^24 BN254 MSM (rocprofv3 kernel trace over scripts/msm_probe.py)
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/mk
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/mk -o p -- python3 scripts/msm_probe.py ${1:-24} > gpurun_out/mk.log 2>&1 || { tail -5 gpurun_out/mk.log; exit 1; }
f=$(find gpurun_out/mk -name '*.db' | head -1)
python3 - "$f" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last MSM = kernels after the last k_msm_coarse<..., 0>
last = max(i for i, r in enumerate(rows) if "k_msm_coarse<" in r[0] and (", 0>" in r[0] or ", 2>" in r[0]))
t0 = rows[last][1]
for nm, s, e in rows[last - 2:]:
    print("%8.3f ms  +%7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, nm[:90]))
PY
rm -rf gpurun_out/mk
