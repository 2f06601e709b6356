"""Developer probe: MSM phase timings at 2^k points (BN254)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
L.set_profiling(True)
for logn in [int(x) for x in sys.argv[1:]] or [24]:
    n = 1 << logn
    bls = os.environ.get("JF_PROBE_CURVE") == "bls"
    R_, G_, cid = (bench.BLS_R, bench.BLS_G, 1) if bls else (bench.BN254_R, bench.BN254_G, 0)
    beta = bench.uniform_fr(1, 1, R_)[0]
    srs = L.Srs.gen_for_testing(cid, beta, G_, n, int(os.environ.get('JF_PROBE_C', '0')))
    sc_np = bench.uniform_fr(n, 2, R_)
    if os.environ.get("JF_PROBE_SCALARS") == "same":     # one bucket per window, table read in index order
        sc_np[:] = sc_np[0]
    sc = torch.from_numpy(sc_np.view(np.int64)).cuda()
    out = torch.zeros(256, dtype=torch.uint8, device="cuda")
    st = L.Stream()
    for _ in range(3):
        torch.cuda.synchronize(); t = time.time()
        L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st); st.sync()
        dt = time.time() - t
    ph = L.last_phase_ms(st, 0)
    print(f"msm 2^{logn} c={srs.window_bits} {dt*1e3:.2f} ms {n/dt/1e6:.1f} Mpts/s", {k: round(v, 3) for k, v in ph.items()},
          "sha", __import__("hashlib").sha256(out.cpu().numpy().tobytes()).hexdigest()[:8], flush=True)
    srs.free()
