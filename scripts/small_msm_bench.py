"""Verifier-side small MSM (jfhip_msm_small, host points + host scalars) and short SRS commitments: wall time per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from jellyfish_amd import lib as hip
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

hip.init(0)
for cv in (R.BN254, R.BLS12_381):
    srs = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x1234])[0], pt_arr(cv, cv.g), 4096)
    pts = srs.read(0, 4096)
    for n in (4, 32, 128, 1024, 4096):
        sc = rand_fr_mont(cv, n, n)
        for name, f in (("msm_small (host points)", lambda: hip.msm_small(cv.curve_id, pts[:n], sc)),
                        ("msm on the resident SRS", lambda: hip.msm(srs, sc))):
            f()
            t0 = time.perf_counter()
            for _ in range(10):
                f()
            print(f"{cv.name:10s} n={n:5d} {name:26s} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms")
    srs.free()
