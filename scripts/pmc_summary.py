"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) of scripts/pmc_run.py into
profiles/pmc_latest.json.  usage: python scripts/pmc_summary.py <fetch_counter_csv> <write_counter_csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B as rocprofv3 reports them (TCC_EA0_RDREQ x 64 B / 1024);
per MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports wide coalesced streaming reads by 2x on gfx950 and is
uncalibrated for other shapes: the bucket-accumulation kernel gathers 64-byte points, so its raw figure is reported
together with the doubled figure as an upper bound."""
import collections
import csv
import json
import sys


def total_over(path, counter, needle_a, needle_b):
    """sum of a counter over every launch of kernels whose name holds both needles -> bytes"""
    t = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle_a in r["Kernel_Name"] and needle_b in r["Kernel_Name"]:
            t += float(r["Counter_Value"])
    return t * 1024.0


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in d.items()}  # bytes per launch


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402

out = {"workload": "msm_bn254_g1_2^24", "git_head": __import__("os").environ.get("JF_GIT_HEAD", ""),
       "kernel_source_sha": bench.kernel_source_sha(), "unit": "bytes per launch (average over the profiled launches)",
       "note": "FETCH_SIZE raw; gfx950 halves wide streaming reads (MI355X_MICROARCH.md), so raw <= true <= 2 x raw",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "jf::" in k:
        out["kernels"][k.replace("void ", "")] = {"fetch_raw": fetch.get(k), "write": write.get(k)}
acc = [k for k in fetch if "k_msm_accumulate" in k]
if acc:
    f, w = fetch[acc[0]], write.get(acc[0], 0.0)
    out["k_msm_accumulate_hbm_bytes"] = f + w
    out["k_msm_accumulate_hbm_bytes_upper"] = 2 * f + w
# config 3: scripts/pmc_run.py runs 4 transforms (2 forward + 2 inverse) of 8 x 2^22 BLS12-381 Fr
nf = total_over(sys.argv[1], "FETCH_SIZE", "k_ntt_pass", "Bls381Fr")
nw = total_over(sys.argv[2], "WRITE_SIZE", "k_ntt_pass", "Bls381Fr")
if nf or nw:
    out["ntt_8x2^22_bls_hbm_bytes_per_transform"] = (nf + nw) / 4.0
    out["ntt_8x2^22_bls_hbm_bytes_per_transform_upper"] = (2 * nf + nw) / 4.0
    out["ntt_note"] = "sum over the digit passes of one transform, averaged over 2 forward + 2 inverse; algorithmic 2.147e9"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:600])
