"""Real PlonkKzgSnark::prove at 2^K gates (BN254 bench circuit), a few runs with a per-step wall-clock breakdown.
usage: python scripts/plonk_prove_run.py [K=20] [runs=3]   (also the program rocprofv3 is pointed at)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jellyfish_amd import hostglue as H  # noqa: E402
from jellyfish_amd import lib as L  # noqa: E402
from jellyfish_amd import plonk as JP  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
L.load_library()
L.init(0)
t0 = time.perf_counter()
cs = JP.PlonkCircuit.new_turbo_plonk(0)
a = cs.zero()
for _ in range((1 << K) - 10):
    a = cs.add(a, cs.one())
cs.finalize_for_arithmetization()
wit = H.fr_to_mont(cs.cp, cs.witness)
print("circuit build %.2f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, cs.eval_domain_size() + 2, 0x6A660004)
pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
torch.cuda.synchronize()
print("setup + preprocess %.2f s" % (time.perf_counter() - t0), flush=True)
rng = H.test_rng()
JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
if os.environ.get("JF_PK_RESIDENT"):
    pk.keep_coset_evals()
    print("proving key keeps selector / sigma coset evaluations: %.2f GiB" % (pk.coset_evals_bytes() / 2**30))
for _ in range(runs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    JP.PROFILE = {} if os.environ.get("JF_PROVE_PROFILE") else None
    JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
    torch.cuda.synchronize()
    print("prove %.2f ms" % ((time.perf_counter() - t0) * 1e3), JP.PROFILE or "", flush=True)
if os.environ.get("JF_HOST_PROFILE"):
    # host side of one proof under cProfile: where the Python time between device rounds goes
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
