#!/bin/bash
# GPU busy time per proof: total kernel time of (warm-up + R proofs) for two values of R under rocprofv3, differenced.
set -o pipefail
export TMPDIR=/tmp
for R in 2 12; do
  rm -rf gpurun_out/pb_$R
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pb_$R -o p -- python3 scripts/plonk_prove_run.py 20 $R > gpurun_out/pb_$R.log 2>&1 || { tail -5 gpurun_out/pb_$R.log; exit 1; }
  f=$(find gpurun_out/pb_$R -name '*.db' | head -1)
  python3 - "$f" $R <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tot, n = c.execute("select sum(duration), count(*) from kernels").fetchone()
print("R=%s kernels=%d total_ms=%.3f" % (sys.argv[2], n, tot / 1e6))
rows = c.execute("select name, count(*), sum(duration) from kernels group by name order by sum(duration) desc limit 14").fetchall()
for r in rows: print("   %-60s %6d %10.3f ms" % (r[0][:60], r[1], r[2] / 1e6))
PY
  rm -rf gpurun_out/pb_$R
done
grep "prove " gpurun_out/pb_12.log | tail -3
