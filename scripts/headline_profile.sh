#!/bin/bash
# the metric's command alone under rocprofv3 --kernel-trace --stats: every k_msm_accumulate launch in the summary is a
# 2^24-point step, so its average must agree with roofline.avg_launch_ms of the JSON line printed by the same run
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/hl && mkdir -p gpurun_out/hl
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/hl -o hl -- python3 bench.py --headline-only > gpurun_out/headline_bench.json 2> gpurun_out/headline_bench.err || { tail -20 gpurun_out/headline_bench.err; exit 2; }
find gpurun_out/hl -name '*.db' | while read f; do python3 scripts/rocpd_stats.py "$f" gpurun_out/headline_kernel_stats.csv; done
rm -rf gpurun_out/hl
python3 - <<'PY'
import csv, json
d = json.loads(open("gpurun_out/headline_bench.json").read().strip().splitlines()[-1])
print("bench: value %.1f %s, roofline.avg_launch_ms %.4f, frac %.4f" % (d["value"], d["unit"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
for r in csv.DictReader(open("gpurun_out/headline_kernel_stats.csv")):
    if "accumulate" in r["Name"]:
        print("rocprofv3: %s calls %s avg %.4f ms min %.4f max %.4f" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
