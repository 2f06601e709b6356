"""Developer probe: MSM time against window_bits at 2^k points (BN254, uniform scalars)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
logn = int(sys.argv[1]); cs = [int(x) for x in sys.argv[2:]]
n = 1 << logn
beta = bench.uniform_fr(1, 1, bench.BN254_R)[0]
sc = torch.from_numpy(bench.uniform_fr(n, 2, bench.BN254_R).view(np.int64)).cuda()
out = torch.zeros(128, dtype=torch.uint8, device="cuda")
st = L.Stream()
L.set_profiling(True)
for c in cs:
    srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, n, window_bits=c)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st); st.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"2^{logn} c={c} {min(ts[1:]):.3f} ms", {k: round(v, 2) for k, v in L.last_phase_ms(st, 0).items()}, flush=True)
    srs.free()
