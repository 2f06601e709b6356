"""Per-kernel averages of rocprofv3 --pmc passes (scripts/pmc_valu.sh): for every libjfhip kernel, each counter's mean per
launch, plus derived figures for the dominant kernels.  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves (MI355X_MICROARCH.md, cycle-constants table); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
usage: pmc_valu_summary.py out.json pass1.csv [pass2.csv ...]"""
import collections
import csv
import json
import sys

out_path, paths = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in paths:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "jf::" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha: ties this profile to the kernel sources it was taken on)

out = {"unit": "mean per launch", "git_head": __import__("os").environ.get("JF_GIT_HEAD", ""),
       "kernel_source_sha": bench.kernel_source_sha(), "kernels": {}}
for k, cs in sorted(acc.items()):
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if d.get("SQ_WAVE_CYCLES"):
        wc = d["SQ_WAVE_CYCLES"]
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM"):
            if c in d:
                d["frac_of_wave_cycles:" + c] = round(d[c] / wc, 4)
    if d.get("SQ_INSTS_VALU") and d.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
    if d.get("SQ_INSTS_VALU") and d.get("GRBM_GUI_ACTIVE"):
        # wave-instructions per shader cycle over the whole chip (1024 SIMDs): GRBM_GUI_ACTIVE / 8 = kernel cycles
        d["valu_wave_insts_per_cycle_chip"] = round(d["SQ_INSTS_VALU"] / (d["GRBM_GUI_ACTIVE"] / 8.0), 2)
    if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
    if d.get("TCP_UTCL1_REQUEST_sum"):
        d["utcl1_miss_rate"] = round(d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0) / d["TCP_UTCL1_REQUEST_sum"], 5)
    if d.get("TCP_TCC_READ_REQ_sum"):
        d["tcp_read_latency_cycles_per_req"] = round(d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / d["TCP_TCC_READ_REQ_sum"], 1)
    out["kernels"][k] = d
json.dump(out, open(out_path, "w"), indent=1)
for k, d in out["kernels"].items():
    if "accumulate" in k or "ntt_pass_r8" in k:
        print(k, json.dumps({x: d[x] for x in d if not x.startswith("TC")})[:900])
