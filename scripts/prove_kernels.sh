#!/bin/bash
# per-kernel-name totals of the LAST proof of a short run (rocprofv3 kernel trace over scripts/plonk_prove_run.py)
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/pk
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/pk -o p -- python3 scripts/plonk_prove_run.py ${1:-20} 2 > gpurun_out/pk.log 2>&1 || { tail -5 gpurun_out/pk.log; exit 1; }
grep -E "prove|ms" gpurun_out/pk.log | tail -6
f=$(find gpurun_out/pk -name '*.db' | head -1)
python3 - "$f" <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_gather_rows32" in r[0]]
seg = rows[idx[-1]:] if idx else rows[len(rows) * 2 // 3:]
tot = collections.defaultdict(lambda: [0, 0])
for nm, s, e in seg:
    k = nm.split("(")[0][:70]
    tot[k][0] += e - s
    tot[k][1] += 1
print("last proof: span %.2f ms, kernel sum %.2f ms, %d launches" % ((max(r[2] for r in seg) - seg[0][1]) / 1e6, sum(v[0] for v in tot.values()) / 1e6, len(seg)))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:28]:
    print("%8.3f ms %5d  %s" % (v[0] / 1e6, v[1], k))
PY
rm -rf gpurun_out/pk
