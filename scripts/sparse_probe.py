"""Developer probe: MSM phases for a sparse scalar vector (4 non-zero of 2^k + 2), BN254."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
L.set_profiling(True)
for logn in [int(x) for x in sys.argv[1:]] or [17, 20]:
    n = (1 << logn) + 2
    beta = bench.uniform_fr(1, 1, bench.BN254_R)[0]
    srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, n)
    dense = bench.uniform_fr(n, 2, bench.BN254_R)
    sparse = np.zeros_like(dense)
    for i in (0, 1, n - 2, n - 1):
        sparse[i] = dense[i]
    st = L.Stream()
    out = torch.zeros(128, dtype=torch.uint8, device="cuda")
    for name, sc_np in (("dense", dense), ("sparse", sparse)):
        sc = torch.from_numpy(sc_np.view(np.int64)).cuda()
        for _ in range(3):
            torch.cuda.synchronize(); t = time.time()
            L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st); st.sync()
            dt = time.time() - t
        print(f"2^{logn} c={srs.window_bits} {name} {dt*1e3:.2f} ms", {k: round(v, 3) for k, v in L.last_phase_ms(st, 0).items()}, flush=True)
    srs.free()
