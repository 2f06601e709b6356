"""jfhip_msm with host scalars (BASELINE config 2's timed region: scalar upload + kernels + one point back), BN254.
Run with JFHIP_MSM_UPLOAD_CHUNKS=1 for the single-upload path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from jellyfish_amd import lib as L

L.init(0)
print("JFHIP_MSM_UPLOAD_CHUNKS =", os.environ.get("JFHIP_MSM_UPLOAD_CHUNKS", "(unset)"))
beta = bench.uniform_fr(1, 0x6A660001, bench.BN254_R)[0]
for log_n in (18, 20, 22, 24):
    n = (1 << log_n) + (1 if log_n == 20 else 0)
    srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, n)
    sc = bench.uniform_fr(n, 7, bench.BN254_R)
    st = L.Stream()
    out = None
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        out, _ = L.msm(srs, sc, stream=st)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"2^{log_n}: {np.median(ts[2:]):8.3f} ms  sha {bytes(out.tobytes()).hex()[:16]}")
    srs.free()
