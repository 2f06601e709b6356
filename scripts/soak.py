"""Soak: the same calls repeated many times must return the same bytes every time (races between the lane streams,
workspace reuse and the per-handle caches would show as a differing result).  Prints one line per section; exits 1 on
the first difference."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from jellyfish_amd import hostglue as H
from jellyfish_amd import lib as L
from jellyfish_amd import plonk as JP
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

torch.cuda.set_device(0)
L.init(0)


def same(name, f, reps):
    t0 = time.perf_counter()
    first = f()
    for i in range(1, reps):
        if f() != first:
            print(f"{name}: call {i} differs from call 0")
            sys.exit(1)
    print(f"{name}: {reps} identical results ({hashlib.sha256(first).hexdigest()[:12]}), {time.perf_counter() - t0:.1f} s", flush=True)


def circuit(ultra, curve, log_g):
    cs = JP.PlonkCircuit.new_ultra_plonk(curve, 8) if ultra else JP.PlonkCircuit.new_turbo_plonk(curve)
    a = cs.zero()
    for _ in range((1 << log_g) - 10):
        a = cs.add(a, cs.one())
    cs.finalize_for_arithmetization()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(curve, cs.eval_domain_size() + 2, 0x50AC + log_g)
    pk, _ = JP.PlonkKzgSnark.preprocess(srs, cs)
    return cs, pk, H.fr_to_mont(cs.cp, cs.witness)


for ultra, curve, log_g, reps in ((False, 0, 14, 150), (True, 1, 13, 60), (False, 0, 18, 30)):
    cs, pk, wit = circuit(ultra, curve, log_g)
    same(f"prove {'ultra' if ultra else 'turbo'} curve {curve} 2^{log_g}",
         lambda: JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed(), reps)
    same(f"batch_prove x2 {'ultra' if ultra else 'turbo'} curve {curve} 2^{log_g}",
         lambda: JP.PlonkKzgSnark.batch_prove(H.test_rng(), [cs, cs], [pk, pk], witness_monts=[wit, wit]).serialize_compressed(),
         max(reps // 5, 5))
    pk.keep_coset_evals()
    same("  the same with resident coset evaluations",
         lambda: JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed(), max(reps // 3, 5))
    pk.keep_lagrange_commit_key()
    same("  ... and round-1 commitments from the witness values (Lagrange-basis key)",
         lambda: JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed(), max(reps // 3, 5))
    pk.drop_lagrange_commit_key()
    del pk

cv = R.BN254
beta = bench.uniform_fr(1, 0x6A660001, bench.BN254_R)[0]
for log_n, reps in ((20, 40), (22, 12)):
    srs = L.Srs.gen_for_testing(0, beta, bench.BN254_G, 1 << log_n)
    sc = bench.uniform_fr(1 << log_n, 9, bench.BN254_R)
    same(f"jfhip_msm host scalars 2^{log_n} (chunked upload)", lambda: bytes(L.msm(srs, sc)[0].tobytes()), reps)
    polys = [sc[: (1 << log_n) - k] for k in range(5)]
    same(f"jfhip_msm_batch 5 x 2^{log_n}", lambda: bytes(L.msm_batch(srs, polys)[0].tobytes()), max(reps // 4, 4))
    short = sc[:3000]
    same(f"short MSM on the 2^{log_n} SRS (prefix table)", lambda: bytes(L.msm(srs, short)[0].tobytes()), 50)
    srs.free()

cv = R.BLS12_381
G = pt_arr(cv, cv.g)
nv = 14
levels = [L.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x77 + i])[0], G, 1 << (nv - 1 - i)) for i in range(nv)]
ev, pt = rand_fr_mont(cv, 1 << nv, 3), rand_fr_mont(cv, nv, 4)
same("mle_open nv = 14", lambda: bytes(L.mle_open(levels, ev, pt)[0].tobytes()), 40)
srs = L.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x99])[0], G, 257)
f = rand_fr_mont(cv, 256, 5)
same("FK23 256 proofs", lambda: bytes(L.kzg_multi_open_rou(srs, f, 8, 256)[0].tobytes()), 10)
print("soak ok")
