"""GPU: repeated identical calls return identical bytes.  The library overlaps independent MSMs on lane streams, reuses
grow-only workspaces and keeps per-handle caches (prefix window tables, FK23 transforms); a missing ordering between any
of those would show as a result that changes from call to call (scripts/soak.py is the long form of this test)."""
import numpy as np
import pytest

from jellyfish_amd import hostglue as H
from jellyfish_amd import plonk as JP
from oracle import coracle as C
from oracle import pyref as R
from tests.util import mont_arr, pt_arr, rand_fr_mont

pytestmark = pytest.mark.gpu


def test_repeated_commits_are_identical(hip):
    cv = R.BN254
    n = (1 << 20) + 3
    srs = hip.Srs.gen_for_testing(cv.curve_id, mont_arr(cv.fr, [0x5EED])[0], pt_arr(cv, cv.g), n)
    sc = rand_fr_mont(cv, n, 41)
    polys = [sc[:n - k] for k in range(5)] + [sc[:1000]]
    first_batch = hip.msm_batch(srs, polys)[0].copy()
    first_one = hip.msm(srs, sc)[0].copy()          # host scalars: uploaded in two ranges from 2^20 terms on
    assert (first_batch[0] == first_one).all()
    for _ in range(8):
        assert (hip.msm_batch(srs, polys)[0] == first_batch).all()
        assert (hip.msm(srs, sc)[0] == first_one).all()
    # the short entry against the oracle (prefix window table of the handle)
    assert (first_batch[5] == C.msm(cv.curve_id, srs.read(0, 1000), polys[5])).all()
    srs.free()


@pytest.mark.parametrize("ultra", [False, True], ids=["turbo", "ultra"])
def test_repeated_proofs_are_identical(hip, ultra):
    cs = JP.PlonkCircuit.new_ultra_plonk(1, 8) if ultra else JP.PlonkCircuit.new_turbo_plonk(0)
    a = cs.zero()
    for _ in range((1 << 12) - 10):
        a = cs.add(a, cs.one())
    cs.finalize_for_arithmetization()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(1 if ultra else 0, cs.eval_domain_size() + 2, 0x4E9)
    pk, _ = JP.PlonkKzgSnark.preprocess(srs, cs)
    wit = H.fr_to_mont(cs.cp, cs.witness)
    first = JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed()
    for _ in range(20):
        assert JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed() == first
    pk.keep_coset_evals()
    for _ in range(5):
        assert JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed() == first
