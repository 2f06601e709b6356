#!/bin/bash
# every kernel >= 40 us of the LAST proof of a short run, in start order, with start / end offsets (ms): what overlaps what
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/gt
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/gt -o p -- python3 scripts/plonk_prove_run.py ${1:-20} 2 > gpurun_out/gt.log 2>&1 || { tail -5 gpurun_out/gt.log; exit 1; }
f=$(find gpurun_out/gt -name '*.db' | head -1)
python3 - "$f" > gpurun_out/gantt.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
idx = [i for i, r in enumerate(rows) if "k_witness_gather" in r[0]]
seg = rows[idx[-1]:] if idx else rows[len(rows) * 2 // 3:]
t0 = seg[0][1]
for nm, s, e, qid in seg:
    if e - s >= 40000:
        print("%8.3f %8.3f  +%6.3f  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qid, nm.split("(")[0].replace("void jf::", "")[:60]))
PY
tail -3 gpurun_out/gantt.txt
rm -rf gpurun_out/gt
