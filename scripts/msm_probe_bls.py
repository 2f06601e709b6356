"""Developer probe: MSM phase timings at 2^k points, BLS12-381 G1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
L.set_profiling(True)
for arg in sys.argv[1:] or ["22"]:
    logn, _, c = arg.partition(":")
    logn = int(logn); n = 1 << logn
    beta = bench.uniform_fr(1, 1, bench.BLS_R)[0]
    srs = L.Srs.gen_for_testing(1, beta, bench.BLS_G, n, window_bits=int(c) if c else 0)
    sc = torch.from_numpy(bench.uniform_fr(n, 2, bench.BLS_R).view(np.int64)).cuda()
    out = torch.zeros(256, dtype=torch.uint8, device="cuda")
    st = L.Stream()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st); st.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"bls msm 2^{logn} c={srs.window_bits} {min(ts[1:]):.2f} ms {n/min(ts[1:])/1e3:.1f} Mpts/s", {k: round(v, 2) for k, v in L.last_phase_ms(st, 0).items()}, flush=True)
    srs.free()
