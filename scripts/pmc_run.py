"""One warm + two timed MSMs at the bench shape (2^24 BN254), for rocprofv3 --pmc passes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L
import bench
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
L.init(0)
beta = bench.uniform_fr(1, 0x6A660001, bench.BN254_R)[0]
srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, n)
sc = torch.from_numpy(bench.uniform_fr(n, 0xC0FFEE, bench.BN254_R).view(np.int64)).cuda()
out = torch.zeros(128, dtype=torch.uint8, device="cuda")
st = L.Stream()
for _ in range(3):
    L.msm_device(srs, sc.data_ptr(), n, out.data_ptr(), stream=st)
st.sync()
# BASELINE config 3 in the same passes: 8 x 2^22 BLS12-381 Fr coset NTT, 2 forward + 2 inverse transforms
if logn == 24:
    m, batch = 1 << 22, 8
    data = torch.from_numpy(bench.uniform_fr(batch * m, 0xFF7, bench.BLS_R).view(np.int64)).cuda()
    h = L.field_op(2, "to_mont", np.array([7, 0, 0, 0], dtype=np.uint64))
    for inverse in (False, True, False, True):
        L.ntt_device(L.BLS12_381, data.data_ptr(), 22, batch, inverse=inverse, coset_mont=h, stream=st)
    st.sync()
print("done")
