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
idx = [i for i, r in enumerate(rows) if "k_gather_rows32" in r[0]]
seg = rows[idx[-1]:] if idx else rows[len(rows) * 2 // 3:]
t0 = seg[0][1]
for nm, s, e, qid in seg:
    if e - s >= 40000:
        print("%8.3f %8.3f  +%6.3f  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qid, nm.split("(")[0].replace("void jf::", "")[:60]))
# device idle time: gaps in the union of all kernel intervals (host syncs, read-backs, launch latency)
ivs = sorted((s, e, nm) for nm, s, e, _ in seg)
cur_end, cur_nm, gaps = ivs[0][1], ivs[0][2], []
for s, e, nm in ivs[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, (cur_end - t0) / 1e6, cur_nm, nm))
    if e > cur_end:
        cur_end, cur_nm = e, nm
short = lambda x: x.split("(")[0].replace("void jf::", "")[:40]
print("idle: %.3f ms in %d gaps (span %.3f ms); gaps >= 30 us:" % (sum(g[0] for g in gaps) / 1e6, len(gaps), (cur_end - t0) / 1e6))
for g in sorted(gaps, key=lambda g: -g[0]):
    if g[0] >= 30000:
        print("  %7.3f ms at %8.3f  after %-40s before %s" % (g[0] / 1e6, g[1], short(g[2]), short(g[3])))
PY
grep -A40 "^idle" gpurun_out/gantt.txt | head -60
rm -rf gpurun_out/gt
