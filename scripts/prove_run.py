"""Runs the TurboPlonk hot-path sequence (2^20 gates, BN254) a few times: for rocprofv3."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import lib as L, prover_path as PP
import bench
log_g = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L.init(0)
beta = bench.uniform_fr(1, 1, bench.BN254_R)[0]
srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, (1 << log_g) + 8)
hp = PP.TurboPlonkHotPath(0, log_g, srs, lambda c, s: bench.uniform_fr(c, 7000 + s, bench.BN254_R))
import time
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    hp.prove(); torch.cuda.synchronize()
    print("prove ms", (time.perf_counter() - t) * 1e3, flush=True)
