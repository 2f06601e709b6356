"""Two host threads proving at once, each on its own torch / jfhip stream (different curves and domain sizes, so the
process-wide NTT / quotient table caches serve both): every proof must equal the one computed alone."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from jellyfish_amd import hostglue as H
from jellyfish_amd import lib as L
from jellyfish_amd import plonk as JP

torch.cuda.set_device(0)
L.init(0)


def circuit(ultra, curve, log_g):
    cs = JP.PlonkCircuit.new_ultra_plonk(curve, 8) if ultra else JP.PlonkCircuit.new_turbo_plonk(curve)
    a = cs.zero()
    for _ in range((1 << log_g) - 10):
        a = cs.add(a, cs.one())
    cs.finalize_for_arithmetization()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(curve, cs.eval_domain_size() + 2, 0x7EAD + log_g)
    pk, _ = JP.PlonkKzgSnark.preprocess(srs, cs)
    return cs, pk, H.fr_to_mont(cs.cp, cs.witness)


jobs = [circuit(False, 0, 15), circuit(True, 1, 13), circuit(False, 1, 12)]
alone = [JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=w).serialize_compressed() for cs, pk, w in jobs]
torch.cuda.synchronize()
bad = []


def worker(k, reps):
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    cs, pk, w = jobs[k]
    with torch.cuda.stream(s):
        for i in range(reps):
            p = JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=w).serialize_compressed()
            if p != alone[k]:
                bad.append((k, i))
                return


ths = [threading.Thread(target=worker, args=(k, 40)) for k in range(3)]
for t in ths:
    t.start()
for t in ths:
    t.join()
torch.cuda.synchronize()
print("differences:", bad)
sys.exit(1 if bad else 0)
