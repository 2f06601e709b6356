"""Developer probe: MSM time against the sparsity of the scalars (witness-like values), BN254 2^20 and BLS12-381 2^22."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
L.set_profiling(True)
ONLY_CURVE = os.environ.get("JF_CURVE")
ONLY_CASE = os.environ.get("JF_CASE")
for cid, logn, R_, G_ in ((0, 20, bench.BN254_R, bench.BN254_G), (1, 22, bench.BLS_R, bench.BLS_G)):
    if ONLY_CURVE is not None and int(ONLY_CURVE) != cid:
        continue
    n = (1 << logn) + 3
    beta = bench.uniform_fr(1, 1, R_)[0]
    srs = L.Srs.gen_for_testing(cid, beta, G_, n, 0)
    dense = bench.uniform_fr(n, 2, R_)
    fid = 0 if cid == 0 else 2
    def small(vals):
        a = np.zeros((n, 4), dtype=np.uint64)
        a[:, 0] = vals
        return L.field_op(fid, "to_mont", a)
    cases = {"dense": dense, "counter i": small(np.arange(n, dtype=np.uint64)), "all ones": small(np.ones(n, dtype=np.uint64)),
             "bytes": small((np.arange(n, dtype=np.uint64) * 2654435761) % 256), "booleans": small(np.arange(n, dtype=np.uint64) % 2),
             "64-bit": small((np.arange(n, dtype=np.uint64) + 1) * np.uint64(0x9E3779B97F4A7C15))}
    st = L.Stream()
    out = torch.zeros(256, dtype=torch.uint8, device="cuda")
    for name, sc_np in cases.items():
        if ONLY_CASE is not None and ONLY_CASE != name:
            continue
        sc = torch.from_numpy(np.ascontiguousarray(sc_np).view(np.int64)).cuda()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st); st.sync()
            ts.append(time.perf_counter() - t)
        ph = L.last_phase_ms(st, 0)
        print(f"curve {cid} 2^{logn} c={srs.window_bits} {name:10s} {min(ts[1:])*1e3:7.3f} ms", {k: round(v, 3) for k, v in ph.items()}, flush=True)
    srs.free()
