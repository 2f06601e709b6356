"""Short commitments against a large SRS: time of one device-resident MSM of 2^k terms on a 2^LOG_N-point BN254 SRS.
Run twice, with JFHIP_SRS_PREFIX=0 (the handle's own window table) and unset (prefix tables, csrc/api.hip srs_route)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from jellyfish_amd import lib as hip
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

cv = R.BN254
log_n = int(os.environ.get("LOG_N", "24"))
n = 1 << log_n
hip.init(0)
srs = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x1234567])[0], pt_arr(cv, cv.g), n)
print(f"SRS 2^{log_n}, window_bits {srs.window_bits}, JFHIP_SRS_PREFIX={os.environ.get('JFHIP_SRS_PREFIX', '(unset)')}")
for k in (10, 12, 14, 16, 18, 20):
    if (1 << k) > n:
        break
    m = 1 << k
    sc = torch.from_numpy(rand_fr_mont(cv, m, k).view(np.int64)).cuda()
    out = torch.zeros(80, dtype=torch.uint8, device="cuda")
    st = hip.Stream()
    for _ in range(3):
        hip.msm_device(srs, sc.data_ptr(), m, out.data_ptr(), stream=st)
    st.sync()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        hip.msm_device(srs, sc.data_ptr(), m, out.data_ptr(), stream=st)
    st.sync()
    print(f"  2^{k:2d} terms: {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms")
srs.free()
