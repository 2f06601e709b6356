"""Timing experiment: bucket accumulation of a 2^24-point BN254 MSM when the random table gathers
are made slab-local.  Needs the experimental build of the library (scripts/micro/slabtest.patch applied to csrc/msm.hip,
compiled with -DJFHIP_SLABTEST into jellyfish_amd/libjfhip_slabtest.so); the sums are WRONG in modes 1-3 by construction,
only `bucket_accumulate` is read.  Modes: 0 the shipped kernel, 1 each XCD gathers from its own 1/8 of the window table,
2 XCD pairs from their 1/4, 3 every lane from window 0 only (1 GiB: no gather wall at all)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench

L.library_path = lambda: os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libjfhip_slabtest.so")
L.init(0)
L.set_profiling(True)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
beta = bench.uniform_fr(1, 1, bench.BN254_R)[0]
srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, n, 0)
sc = torch.from_numpy(bench.uniform_fr(n, 2, bench.BN254_R).view(np.int64)).cuda()
out = torch.zeros(256, dtype=torch.uint8, device="cuda")
st = L.Stream()
res = {"log_n": logn, "window_bits": srs.window_bits, "modes": {}}
for mode in (0, 1, 2, 3, 0):
    os.environ["JFHIP_SLABTEST"] = str(mode)
    acc = []
    for _ in range(4):
        L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st)
        st.sync()
        acc.append(L.last_phase_ms(st, 0)["bucket_accumulate"])
    res["modes"].setdefault(str(mode), []).append(round(min(acc[1:]), 4))
    print(mode, [round(a, 3) for a in acc], flush=True)
print(json.dumps(res))
srs.free()
