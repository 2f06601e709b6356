"""Per-round device time of the LAST proof in a rocprofv3 --kernel-trace --marker-trace database, segmented by the roctx
marks jellyfish_amd/plonk.py leaves at the end of every step ("prove step end: <name>", JFHIP_MARKERS=1), plus the
library's own ranges (named after the reference's timers) counted per name.
usage: python scripts/prove_trace_summary.py results.db out.json [K]"""
import json
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
c = sqlite3.connect(db)
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]


def cols(t):
    return [r[1] for r in c.execute("pragma table_info('%s')" % t)]


# marker ranges: rocprofv3 stores roctx ranges in the view `regions` (category MARKER_CORE_RANGE_API, name =
# roctxThreadRangeA, the message in the JSON column extdata)
def message(ext, name):
    try:
        return json.loads(ext).get("message", name)
    except Exception:
        return name


table = "regions"
raw = c.execute("select name, extdata, start, end from regions where category like 'MARKER%' order by start").fetchall()
ranges = [(message(ext, nm), s, e) for nm, ext, s, e in raw]
rows = [r for r in ranges if r[0].startswith("prove step end:")]
if not rows:
    print("no marker rows found; tables / views:", names)
    sys.exit(3)
agg = {}
for nm, s, e in ranges:
    a = agg.setdefault(nm, [0, 0])
    a[0] += 1
    a[1] += e - s
all_ranges = sorted(((nm, v[0], v[1]) for nm, v in agg.items()), key=lambda r: -r[2])
kern = c.execute("select name, start, end from kernels order by start").fetchall()
# the last proof: from the last "start" mark on
starts = [i for i, r in enumerate(rows) if r[0].endswith(": start")]
seg = rows[starts[-1]:]
rounds, order = {}, []
for (n0, s0, e0), (n1, s1, e1) in zip(seg, seg[1:]):
    name = n1.split(": ", 1)[1]
    ks = [(nm, s, e) for nm, s, e in kern if s >= e0 and s < s1]
    busy = sum(e - s for _, s, e in ks)
    if name not in rounds:
        rounds[name] = {"wall_ms": 0.0, "device_busy_ms": 0.0, "kernels": 0}
        order.append(name)
    rounds[name]["wall_ms"] += (s1 - e0) / 1e6
    rounds[name]["device_busy_ms"] += busy / 1e6
    rounds[name]["kernels"] += len(ks)
for v in rounds.values():
    v["wall_ms"] = round(v["wall_ms"], 3)
    v["device_busy_ms"] = round(v["device_busy_ms"], 3)
span = (seg[-1][1] - seg[0][2]) / 1e6
res = {"what": "one PlonkKzgSnark::prove, TurboPlonk BN254 2^%d gates, under rocprofv3 --kernel-trace --marker-trace with "
               "JFHIP_MARKERS=1 (a device sync per step, so kernels lie between the marks of their step)" % K,
       "marker_table": table, "proof_span_ms": round(span, 3),
       "device_busy_ms": round(sum(v["device_busy_ms"] for v in rounds.values()), 3),
       "steps": {k: rounds[k] for k in order},
       "library_ranges": [{"name": n, "count": cnt, "host_ms": round((tot or 0) / 1e6, 3)} for n, cnt, tot in all_ranges
                          if not n.startswith("prove step end")][:40]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:1500])
