"""Developer probe: phase timings of the MSM / NTT kernels and the peak-rate micro-benchmarks."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import lib as L
import torch

def rand_fr(n, seed, bits):
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << (bits - 193)) - 1)
    return out

L.init(0)
print("mad/s %.3e" % L.probe(0), "copy B/s %.3e" % L.probe(1), flush=True)
for f in range(4):
    print("field", f, "mul/s %.3e" % L.probe(2, f), flush=True)
L.set_profiling(True)
G = np.zeros(8, dtype=np.uint64); 
# BN254 generator (1,2) in Montgomery form
ONE = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
TWO = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
G[:4] = ONE; G[4:] = TWO
beta = rand_fr(1, 1, 254)[0]
for logn in [int(x) for x in sys.argv[1:]] or [20]:
    n = 1 << logn
    for c in ([0] if logn > 22 else [0]):
        t = time.time(); srs = L.Srs.gen_for_testing(0, beta, G, n, c); t_srs = time.time() - t
        sc = rand_fr(n, 2, 254)
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        d_out = torch.zeros(64, dtype=torch.int64, device="cuda")
        st = L.Stream()
        for rep in range(3):
            torch.cuda.synchronize(); t = time.time()
            L.msm_device(srs, d_sc.data_ptr(), n, d_out.data_ptr(), stream=st); st.sync()
            dt = time.time() - t
        ph = L.last_phase_ms(st, 0)
        print(f"msm 2^{logn} c={srs.window_bits} srs_build={t_srs:.2f}s time={dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s", {k: round(v, 3) for k, v in ph.items()}, flush=True)
        srs.free(); st.close()
# NTT config 3
for (curve, logn, batch) in [(1, 22, 8), (0, 20, 8), (0, 23, 4)]:
    n = 1 << logn
    d = torch.from_numpy(rand_fr(batch * n, 3, 254).view(np.int64)).cuda()
    cos = np.array([7, 0, 0, 0], dtype=np.uint64)
    cos = L.field_op(2 if curve else 0, "to_mont", cos)
    st = L.Stream()
    for inverse in (False, True):
        for rep in range(3):
            torch.cuda.synchronize(); t = time.time()
            L.ntt_device(curve, d.data_ptr(), logn, batch, inverse, cos, stream=st); st.sync()
            dt = time.time() - t
        ph = L.last_phase_ms(st, 1)
        print(f"ntt curve={curve} 2^{logn} x{batch} inv={inverse} time={dt*1e3:.2f} ms  algGB/s={batch*n*64/dt/1e9:.0f}", {k: round(v, 3) for k, v in ph.items()}, flush=True)
    st.close()
