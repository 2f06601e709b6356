"""Developer probe: cProfile of the host side of PlonkKzgSnark::prove at 2^K gates (BN254 bench circuit)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jellyfish_amd import hostglue as H, lib as L, plonk as JP
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.cuda.set_device(0); L.load_library(); L.init(0)
cs = JP.PlonkCircuit.new_turbo_plonk(0)
a = cs.zero()
for _ in range((1 << K) - 10):
    a = cs.add(a, cs.one())
cs.finalize_for_arithmetization()
wit = H.fr_to_mont(cs.cp, cs.witness)
srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, cs.eval_domain_size() + 2, 0x6A660004)
pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
rng = H.test_rng()
for _ in range(2):
    JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("prove ms", [round(t, 2) for t in ts], flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
