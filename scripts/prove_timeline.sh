#!/bin/bash
# GPU timeline of the LAST proof of a short run: union busy time, idle gaps and the kernels around them.
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o p -- python3 scripts/plonk_prove_run.py 20 2 > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
f=$(find gpurun_out/tl -name '*.db' | head -1)
python3 - "$f" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# last proof: find the last occurrence of the witness-gather marker = first k_ntt_pass after the last big gap... use the last 1/3 heuristically:
# locate starts of proofs by the vectorized_gather kernel (witness gather by index)
idx = [i for i, r in enumerate(rows) if "index" in r[0] and "gather" in r[0].lower() or "index_select" in r[0].lower() or "indexSelect" in r[0]]
print("kernels", len(rows), "gather markers", len(idx))
start_i = idx[-1] if idx else len(rows) * 2 // 3
seg = rows[start_i:]
t0, t1 = seg[0][1], max(r[2] for r in seg)
busy = 0; cur_s, cur_e = seg[0][1], seg[0][2]
gaps = []
for i, (nm, s, e) in enumerate(seg[1:], 1):
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, seg[i-1][0][:50], nm[:50], (cur_e - t0) / 1e6))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.3f ms busy %.3f ms idle %.3f ms kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(seg)))
gaps.sort(reverse=True)
for g in gaps[:25]:
    print("  gap %.3f ms at t=%.2f  after [%s]  before [%s]" % (g[0] / 1e6, g[3], g[1], g[2]))
print("gaps > 20us: %d totalling %.3f ms" % (sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
PY
rm -rf gpurun_out/tl
