"""Throughput of the field inversions on the test surface (jfhip_field_op): Fermat ladder, binary Euclid, batched division steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from jellyfish_amd import lib as hip
from oracle import pyref as R
from tests.util import FIELD_ID, mont_arr

hip.init(0)
for fname, f in (("bn254_fq", R.BN254.fq), ("bls12_381_fq", R.BLS12_381.fq)):
    fid = FIELD_ID[f.name]
    rng = R.SplitMix64(5)
    base = mont_arr(f, [rng.field(f) for _ in range(1024)])
    for n in (64, 1 << 18):
        A = np.tile(base, ((n + 1023) // 1024, 1))[:n].copy()
        for op in ("neg", "inv", "inv_binary", "inv_safegcd"):
            hip.field_op(fid, op, A)
            t0 = time.perf_counter()
            for _ in range(5):
                out = hip.field_op(fid, op, A)
            dt = (time.perf_counter() - t0) / 5
            print(f"{f.name:14s} n={n:7d} {op:12s} {dt * 1e3:9.3f} ms")
