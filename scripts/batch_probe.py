"""Developer probe: batch of P MSMs at 2^k points (BN254) against P single calls; phase timings of one MSM."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << logn
beta = bench.uniform_fr(1, 1, bench.BN254_R)[0]
srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, n)
scs = [torch.from_numpy(bench.uniform_fr(n, 10 + i, bench.BN254_R).view(np.int64)).cuda() for i in range(P)]
out = torch.zeros(P * 80, dtype=torch.uint8, device="cuda")
st = L.Stream()
L.set_profiling(True)
for _ in range(3):
    L.msm_device(srs, scs[0].data_ptr(), n, out.data_ptr(), stream=st); st.sync()
print("phases", {k: round(v, 3) for k, v in L.last_phase_ms(st, 0).items()}, flush=True)
L.set_profiling(False)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(P):
        L.msm_device(srs, scs[i].data_ptr(), n, out.data_ptr() + 80 * i, stream=st)
    st.sync()
    t1 = time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    L.msm_batch_device(srs, [s.data_ptr() for s in scs], [n] * P, out.data_ptr(), stream=st)
    st.sync()
    t2 = time.perf_counter() - t
    print(f"2^{logn} x {P}: singles {t1*1e3:.2f} ms, batch {t2*1e3:.2f} ms", flush=True)
