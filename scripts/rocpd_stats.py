"""Kernel-stats summary (the table `rocprofv3 --kernel-trace --stats` prints) from a rocprofv3 .db file.
usage: python scripts/rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
    "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
    "order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR", "SGPR",
                "LDS", "Scratch"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.4f" % (100.0 * r[2] / tot)] + list(r[6:]))
print("wrote", out, len(rows), "kernels")
