"""Timings of the other PCS users of the path on one GPU: multilinear KZG commit / open and the
FK23 multi-open at roots of unity, BLS12-381 (the curve of the reference's own tests).  The SRS contents do not matter
for timing: every level is [beta^i] g of the right length, generated on the device."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from jellyfish_amd import lib as hip
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

cv = R.BLS12_381
hip.init(0)
G = pt_arr(cv, cv.g)


def timed(f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e3


for nv in (16, 20):
    levels = [hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x77 + i])[0], G, 1 << (nv - 1 - i)) for i in range(nv)]
    top = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x55])[0], G, 1 << nv)
    ev = rand_fr_mont(cv, 1 << nv, 3)
    pt = rand_fr_mont(cv, nv, 4)
    t_commit = timed(lambda: hip.msm(top, ev))
    t_open = timed(lambda: hip.mle_open(levels, ev, pt))
    print(f"multilinear KZG nv={nv}: commit {t_commit:8.2f} ms, open {t_open:8.2f} ms (host evaluations uploaded per call)")
    for lv in levels:
        lv.free()
    top.free()

for log_d in (10, 12):
    d = 1 << log_d
    srs = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x99])[0], G, d + 1)
    f = rand_fr_mont(cv, d, 5)       # degree d - 1
    t = timed(lambda: hip.kzg_multi_open_rou(srs, f, log_d, d), reps=2)
    z = rand_fr_mont(cv, 1, 6)[0]
    t1 = timed(lambda: hip.kzg_open(srs, f, z), reps=5)
    print(f"FK23 multi-open: {d} proofs of a degree-{d - 1} polynomial {t:8.2f} ms; one kzg_open {t1:6.2f} ms "
          f"(x {d} = {t1 * d:9.1f} ms)")
    srs.free()
