import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
L.init(0)
L.set_profiling(True)
st = L.Stream()
h = L.field_op(0, "to_mont", np.array([5, 0, 0, 0], dtype=np.uint64))
for (logn, batch, in_len, inv, coset) in [(23, 25, (1 << 20) + 3, False, True), (23, 25, 0, False, True), (23, 1, 0, True, True), (20, 6, 0, True, False), (22, 8, 0, False, True)]:
    n = 1 << logn
    d = torch.from_numpy(bench.uniform_fr(batch * n, 3, bench.BN254_R).view(np.int64).reshape(-1)).cuda()
    for rep in range(3):
        L.ntt_device(0, d.data_ptr(), logn, batch, inv, h if coset else None, in_len=in_len, stream=st); st.sync()
    ph = L.last_phase_ms(st, 1)
    print(f"ntt 2^{logn} x{batch} in_len={in_len} inv={inv} coset={coset}: total {sum(ph.values()):.2f} ms", {k: round(v, 2) for k, v in ph.items()}, flush=True)
