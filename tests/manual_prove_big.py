"""Manual GPU check (not collected by pytest; run as `python tests/manual_prove_big.py 24`): PlonkKzgSnark::prove at 2^K gates (BN254 bench circuit) verifies under the restated verifier; timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import hostglue as H, lib as L, plonk as JP
from oracle import plonk_ref as P, pyref as R
K = int(sys.argv[1]) if len(sys.argv) > 1 else 22
BETA = 0x1234567890ABCDEF1234567890ABCDEF
torch.cuda.set_device(0); L.load_library(); L.init(0)
t0 = time.perf_counter()
cs = JP.PlonkCircuit.new_turbo_plonk(0)
a = cs.zero()
for _ in range((1 << K) - 10):
    a = cs.add(a, cs.one())
cs.finalize_for_arithmetization()
wit = H.fr_to_mont(cs.cp, cs.witness)
print("circuit build %.1f s" % (time.perf_counter() - t0), flush=True)
srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, cs.eval_domain_size() + 2, BETA)
pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
rng = H.test_rng()
proof = JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
ok = P.verify(R.BN254, vk.as_dict(), [], proof.as_dict(), trapdoor_beta=BETA)
print("2^%d gates: proof verifies: %s" % (K, ok), flush=True)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
    torch.cuda.synchronize()
    print("prove %.1f ms, peak HBM %.1f GiB" % ((time.perf_counter() - t0) * 1e3, torch.cuda.max_memory_allocated() / 2**30), flush=True)
assert ok
