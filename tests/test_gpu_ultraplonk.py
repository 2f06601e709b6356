"""GPU: UltraPlonk (Plookup) PlonkKzgSnark::{preprocess, prove, batch_prove} of jellyfish_amd/plonk.py against the
big-integer restatement oracle/plonk_ref.py (BASELINE config 5's protocol): identical circuits and verifying keys,
byte-identical Proof / BatchProof from the same RNG stream, acceptance by the restated verifier, rejections; at 2^16 and
2^22 gates (config 5's size, BLS12-381) the size-independent property that the proof verifies.
Reference protocol: plonk/src/proof_system/snark.rs tests :852-1064 (prove -> verify for UltraPlonk), :1567-1721."""
import numpy as np
import pytest

from oracle import plonk_ref as P
from oracle import pyref as R
from tests.util import build_ultra_circuit

pytestmark = pytest.mark.gpu

BETA = 0x1234567890ABCDEF1234567890ABCDEF


def _pair(cv, bits, size, extra=0):
    from jellyfish_amd import plonk as JP

    ocs = build_ultra_circuit(P.PlonkCircuit.new_ultra_plonk(cv, bits), size, extra)
    cs = build_ultra_circuit(JP.PlonkCircuit.new_ultra_plonk(cv.curve_id, bits), size, extra)
    return ocs, cs


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_lookup_merge_kernel_matches_oracle(hip, cv):
    """jfhip_plonk_lookup_merge_device against merged_table_value / merged_lookup_wire_value of the restatement"""
    import torch

    from jellyfish_amd import hostglue as H
    from jellyfish_amd import plonk as JP

    ocs, cs = _pair(cv, 3, 10)
    n = cs.eval_domain_size()
    cp = cs.cp
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(cv.curve_id, n + 2, BETA % cv.fr.p)
    pk, _ = JP.PlonkKzgSnark.preprocess(srs, cs)
    tau = 0xABCDEF0123456789ABCDEF % cv.fr.p
    wires = torch.from_numpy(np.ascontiguousarray(H.fr_to_mont(cp, [v for row in ocs.wire_values() for v in row])
                                                  ).view(np.int64).reshape(-1)).cuda()
    out = torch.zeros(2, n * 4, dtype=torch.int64, device="cuda")
    hip.plonk_lookup_merge_device(cv.curve_id, cs.log_n, wires.data_ptr(), pk.plookup_table_vals.data_ptr(),
                                  H.fr_to_mont(cp, [tau])[0], out[0].data_ptr(), out[1].data_ptr())
    torch.cuda.synchronize()
    got = H.fr_from_mont(cp, out.cpu().numpy().view(np.uint64))
    assert got[:n] == ocs.merged_lookup_table(tau)
    assert got[n:] == ocs.merged_lookup_wire_values(tau)
    # the sorted concatenation (host HashMap merge in the reference, constraint_system.rs:1597-1620)
    srt = JP._sorted_lookup_vector(out[0].view(n, 4), out[1].view(n, 4)[:n - 1])
    assert H.fr_from_mont(cp, srt.cpu().numpy().view(np.uint64)) == ocs.lookup_sorted_vec(tau, ocs.merged_lookup_table(tau))


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
@pytest.mark.parametrize("bits,size", [(3, 6), (5, 90)])
def test_ultra_prove_matches_oracle_and_verifies(hip, cv, bits, size):
    from jellyfish_amd import plonk as JP

    ocs, cs = _pair(cv, bits, size)
    n = cs.eval_domain_size()
    assert n == ocs.n
    beta = BETA % cv.fr.p
    osrs = P.universal_setup_for_testing(cv, n + 2, beta)
    opk, ovk = P.preprocess(cv, osrs, ocs)
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(cv.curve_id, n + 2, beta)
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    assert vk.as_dict() == ovk
    assert len(vk.selector_comms) == 14 and len(vk.sigma_comms) == 6

    proof = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    oproof, _, _ = P.prove(cv, P.test_rng(), ocs, opk)
    assert proof.as_dict() == oproof
    assert proof.plookup_proof is not None
    assert proof.serialize_compressed() == P.ser_proof(cv, oproof)
    pub = cs.public_input()
    assert P.verify(cv, vk.as_dict(), pub, proof.as_dict(), trapdoor_beta=beta)
    # wire and sorted-lookup commitments from evaluations on the Lagrange-basis commit key: same bytes
    pk.keep_lagrange_commit_key()
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
    pk.drop_lagrange_commit_key()
    # resident coset evaluations (selectors, sigmas and the four Plookup tables): same bytes
    pk.keep_coset_evals()
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
    pk.coset_evals = None
    # every polynomial transformed like the reference (no zero-polynomial shortcut): same bytes
    JP.SKIP_ZERO_POLYS = False
    try:
        assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
    finally:
        JP.SKIP_ZERO_POLYS = True
    # a caller-owned (non-blocking) jfhip stream instead of torch's current stream: same bytes
    st = hip.Stream()
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk, stream=st).serialize_compressed() == P.ser_proof(cv, oproof)
    st.close()
    # rejections
    bad_pub = list(pub)
    bad_pub[0] = (bad_pub[0] + 1) % cv.fr.p
    assert not P.verify(cv, vk.as_dict(), bad_pub, proof.as_dict(), trapdoor_beta=beta)
    bad = proof.as_dict()
    bad["plookup_proof"] = dict(bad["plookup_proof"], poly_evals=dict(bad["plookup_proof"]["poly_evals"]))
    bad["plookup_proof"]["poly_evals"]["h_1_next_eval"] = (bad["plookup_proof"]["poly_evals"]["h_1_next_eval"] + 1) % cv.fr.p
    assert not P.verify(cv, vk.as_dict(), pub, bad, trapdoor_beta=beta)


def test_ultra_prove_rejects_bad_lookup_and_mismatched_keys(hip):
    """a looked-up tuple outside the table has no sorted vector (constraint_system.rs:1618-1620); a value outside the
    range table likewise; TurboPlonk key with an UltraPlonk circuit is a parameter error (snark.rs:219-224)"""
    from jellyfish_amd import plonk as JP

    cv = R.BN254
    _, cs = _pair(cv, 3, 6)
    n = cs.eval_domain_size()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, n + 2, BETA)
    pk, _ = JP.PlonkKzgSnark.preprocess(srs, cs)
    lookup_val = next(i for i, v in enumerate(cs.witness) if v == 114)    # a looked-up table value (100 + 7 * 2)
    cs.witness[lookup_val] = 115
    with pytest.raises(JP.PlonkError, match="sorted vector"):
        JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    cs.witness[lookup_val] = 114
    ranged = next(i for i, v in enumerate(cs.witness) if v == 6)
    cs.witness[ranged] = 8          # range_size = 8
    with pytest.raises(JP.PlonkError, match="sorted vector"):
        JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    cs.witness[ranged] = 6
    JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    turbo = JP.PlonkCircuit.new_turbo_plonk(0)
    x = turbo.create_public_variable(2)
    turbo.create_public_variable(3)           # same number of public inputs: the type check is what must fire
    while turbo.num_gates() < n - 3:
        x = turbo.add(x, 1)
    turbo.finalize_for_arithmetization()
    tpk, _ = JP.PlonkKzgSnark.preprocess(srs, turbo)
    with pytest.raises(JP.PlonkError, match="Mismatched Plonk types|inconsistent"):
        JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, tpk)


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_ultra_batch_prove_matches_oracle_and_verifies(hip, cv):
    """batch_prove of 2 UltraPlonk instances -> one BatchProof (alpha_base advances by alpha^7, prover.rs:651-655):
    byte-identical to the restated prover, accepted by the restated batch verifier, order and tampering rejected"""
    from jellyfish_amd import plonk as JP

    pairs = [_pair(cv, 3, 6, extra) for extra in (0, 2)]
    ocss, css = [p[0] for p in pairs], [p[1] for p in pairs]
    n = css[0].eval_domain_size()
    assert all(c.eval_domain_size() == n for c in css)
    beta = BETA % cv.fr.p
    osrs = P.universal_setup_for_testing(cv, n + 2, beta)
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(cv.curve_id, n + 2, beta)
    okeys = [P.preprocess(cv, osrs, o) for o in ocss]
    keys = [JP.PlonkKzgSnark.preprocess(srs, c) for c in css]
    pks, vks = [k[0] for k in keys], [k[1].as_dict() for k in keys]
    pubs = [c.public_input() for c in css]
    bp = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, pks, extra_transcript_init_msg=b"ultra")
    obp, _ = P.batch_prove(cv, P.test_rng(), ocss, [k[0] for k in okeys], extra_msg=b"ultra")
    assert bp.as_dict() == obp
    assert bp.serialize_compressed() == P.ser_batch_proof(cv, obp)
    assert P.verify_batch_proof(cv, vks, pubs, bp.as_dict(), extra_msg=b"ultra", trapdoor_beta=beta)
    assert not P.verify_batch_proof(cv, vks, pubs, bp.as_dict(), trapdoor_beta=beta)
    assert not P.verify_batch_proof(cv, vks[::-1], pubs[::-1], bp.as_dict(), extra_msg=b"ultra", trapdoor_beta=beta)
    one = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css[:1], pks[:1])
    single = JP.PlonkKzgSnark.prove(JP.H.test_rng(), css[0], pks[0])
    assert one.to_proof().serialize_compressed() == single.serialize_compressed()


@pytest.mark.parametrize("log_gates", [16, 22])
def test_ultra_bench_circuit_proof_verifies(hip, log_gates):
    """gen_circuit_for_bench(UltraPlonk, range_bit_len = 8) (plonk/benches/bench.rs:29-46) on BLS12-381 at 2^16 and at
    BASELINE config 5's 2^22 gates (quotient domain 2^25, 43 polynomials of 1 GiB each): the proof computed on the device
    verifies under the restated verifier -- the size-independent property; the restated prover cannot reach this size."""
    from jellyfish_amd import plonk as JP

    cv = R.BLS12_381
    cs = JP.PlonkCircuit.new_ultra_plonk(cv.curve_id, 8)
    a = cs.zero()
    for _ in range((1 << log_gates) - 10):
        a = cs.add(a, cs.one())
    cs.finalize_for_arithmetization()
    n = cs.eval_domain_size()
    assert n == 1 << log_gates
    beta = BETA % cv.fr.p
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(cv.curve_id, n + 2, beta)
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    proof = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    assert P.verify(cv, vk.as_dict(), [], proof.as_dict(), trapdoor_beta=beta)
    q = 48
    assert len(proof.serialize_compressed()) == (8 + 6 * q) + q + (8 + 6 * q) + 2 * q + (8 + 6 * 32) + (8 + 5 * 32) + 32 \
        + 1 + (8 + 2 * q) + q + 15 * 32


def _sorted_ref(table, lookups):
    """the reference's HashMap merge (constraint_system.rs:1597-1620) on integers"""
    cnt = {}
    for x in lookups:
        cnt[x] = cnt.get(x, 0) + 1
    out = []
    for x in table:
        out += [x] * (1 + cnt.pop(x, 0))
    return out if not cnt else None


@pytest.mark.parametrize("shape", ["random", "padded", "alternating", "one_value"])
def test_lookup_sorted_device(hip, shape):
    """jfhip_plonk_lookup_sorted_device: duplicates in the table (counted at the first occurrence), long runs, millions of
    equal lookups, a lookup outside the table"""
    import torch

    rng = np.random.default_rng(5)
    n = 5000 if shape == "random" else 1 << 16
    if shape == "random":
        vals = rng.integers(1, 300, size=n)                       # many repeated, non-adjacent values
        lk = rng.choice(vals, size=n - 1)
    elif shape == "padded":
        vals = np.concatenate([np.arange(1, 257), np.zeros(n - 256, dtype=np.int64)])    # a range table and its padding
        lk = np.concatenate([rng.integers(0, 257, size=n // 2), np.zeros(n - 1 - n // 2, dtype=np.int64)])
    elif shape == "alternating":
        vals = np.tile(np.array([7, 9]), n // 2)
        lk = np.tile(np.array([9, 7, 7]), n)[:n - 1]
    else:
        vals = np.arange(n)
        lk = np.full(n - 1, 12345)
    rows = lambda v: torch.from_numpy(np.stack([v * 3 + 1, v ^ 0x55, v * v, v + 2], axis=1).astype(np.int64)).cuda()
    d_t, d_l = rows(vals), rows(lk)
    out = torch.zeros(2 * n - 1, 4, dtype=torch.int64, device="cuda")
    hip.plonk_lookup_sorted_device(d_t.data_ptr(), n, d_l.data_ptr(), n - 1, out.data_ptr())
    want = _sorted_ref(list(vals), list(lk))
    assert torch.equal(out.cpu(), rows(np.array(want)).cpu())
    # one lookup outside the table
    lk2 = lk.copy()
    lk2[len(lk2) // 2] = 10**9
    with pytest.raises(hip.JfHipError, match="sorted vector has wrong length"):
        hip.plonk_lookup_sorted_device(d_t.data_ptr(), n, rows(lk2).data_ptr(), n - 1, out.data_ptr())
