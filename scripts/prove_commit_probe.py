"""Developer probe: per-polynomial MSM time and phases inside one real prove (2^K gates, BN254 bench circuit)."""
import os, sys, time
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
JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
orig = L.msm_batch_device
def timed_batch(srs_, ptrs, lens, out, stream=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    orig(srs_, ptrs, lens, out, stream=stream)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) * 1e3
    L.set_profiling(True)
    singles = []
    for p, ln in zip(ptrs, lens):
        scratch = torch.zeros(128, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L.msm_device(srs_, p, ln, scratch.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        singles.append((round((time.perf_counter() - t0) * 1e3, 2), {k: round(v, 2) for k, v in L.last_phase_ms(stream, 0).items()}))
    L.set_profiling(False)
    print("batch of %d: %.2f ms; singles:" % (len(ptrs), tb), flush=True)
    for s in singles: print("    ", s, flush=True)
L.msm_batch_device = timed_batch
JP.L.msm_batch_device = timed_batch
JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
