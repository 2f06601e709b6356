"""GPU: PlonkKzgSnark::{preprocess, prove} of jellyfish_amd/plonk.py (every polynomial operation in libjfhip on the
device) against the big-integer restatement oracle/plonk_ref.py: identical verifying keys, byte-identical proofs from the
same RNG stream, and the proofs verify under the restated verifier (plonk/src/proof_system/snark.rs tests :852-1064
protocol: preprocess -> prove -> verify, wrong public input / tampered proof rejected)."""
import pytest

from oracle import plonk_ref as P
from oracle import pyref as R

pytestmark = pytest.mark.gpu

BETA = 0x1234567890ABCDEF1234567890ABCDEF


def build(cs, size):
    """the reference's test circuits are small arithmetic chains (snark.rs:1500-1560 gen_circuit_for_test); this one
    touches every gate kind the product's builder has, public inputs created out of order included"""
    a = cs.create_variable(3)
    b = cs.create_public_variable(5)
    c = cs.mul(a, b)
    d = cs.add(c, a)
    e = cs.create_public_variable(18)
    cs.enforce_equal(d, e)
    f = cs.power5_gate(a)
    cs.enforce_constant(f, 243)
    cs.enforce_bool(cs.sub(f, cs.create_variable(242)))
    x = d
    for i in range(size):
        x = cs.add(x, 1) if i % 3 else cs.mul(x, a)
    cs.create_public_variable(cs.witness[x] + 1)
    cs.finalize_for_arithmetization()
    return cs


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
@pytest.mark.parametrize("size", [9, 100, 500])
def test_prove_matches_oracle_and_verifies(hip, cv, size):
    from jellyfish_amd import plonk as JP

    ocs = build(P.PlonkCircuit(cv), size)
    cs = build(JP.PlonkCircuit(cv.curve_id), size)
    n = cs.eval_domain_size()
    assert n == ocs.n
    beta = BETA % cv.fr.p
    osrs = P.universal_setup_for_testing(cv, n + 2, beta)
    opk, ovk = P.preprocess(cv, osrs, ocs)
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(cv.curve_id, n + 2, beta)
    assert srs.beta_h == osrs["beta_h"]
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    assert vk.selector_comms == ovk["selector_comms"] and vk.sigma_comms == ovk["sigma_comms"] and vk.k == ovk["k"]

    proof = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    oproof, _, _ = P.prove(cv, P.test_rng(), ocs, opk)
    assert proof.as_dict() == oproof
    assert proof.serialize_compressed() == P.ser_proof(cv, oproof)
    # round-1 commitments from the witness values on the Lagrange-basis commit key: same group elements, same bytes
    pk.keep_lagrange_commit_key()
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
    pk.drop_lagrange_commit_key()
    # selector / sigma coset evaluations resident with the proving key: 7 transforms instead of 25, same bytes
    # (per slice of the quotient coset where the prover evaluates on slices, on the whole coset otherwise)
    pk.keep_coset_evals()
    assert (pk.coset_evals_slices is not None) == (n >= 16)
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
    # the reference's own shape: all 8 n points of the coset, with and without resident evaluations
    JP.QUOTIENT_SLICES = False
    try:
        pk.keep_coset_evals()
        assert pk.coset_evals is not None and pk.coset_evals_slices is None
        assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
        pk.coset_evals = None
        assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == P.ser_proof(cv, oproof)
    finally:
        JP.QUOTIENT_SLICES = True
    pub = cs.public_input()
    assert P.verify(cv, vk.as_dict(), pub, proof.as_dict(), trapdoor_beta=beta)
    # a second proof continues the RNG stream: different blinding, still valid; extra transcript message is bound
    rng = JP.H.test_rng()
    JP.PlonkKzgSnark.prove(rng, cs, pk)
    proof2 = JP.PlonkKzgSnark.prove(rng, cs, pk, extra_transcript_init_msg=b"msg")
    assert proof2.wires_poly_comms != proof.wires_poly_comms
    assert P.verify(cv, vk.as_dict(), pub, proof2.as_dict(), extra_msg=b"msg", trapdoor_beta=beta)
    assert not P.verify(cv, vk.as_dict(), pub, proof2.as_dict(), trapdoor_beta=beta)
    bad_pub = list(pub)
    bad_pub[0] = (bad_pub[0] + 1) % cv.fr.p
    assert not P.verify(cv, vk.as_dict(), bad_pub, proof.as_dict(), trapdoor_beta=beta)


def test_prove_rejects_unsatisfied_and_mismatched(hip):
    """an unsatisfied witness gives a quotient of the wrong degree (prover.rs:900-903 WrongQuotientPolyDegree);
    mismatched keys are parameter errors (snark.rs:194-228)"""
    from jellyfish_amd import plonk as JP

    cv = R.BN254
    cs = build(JP.PlonkCircuit(0), 20)
    n = cs.eval_domain_size()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, n + 2, BETA)
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    cs.witness[2] = 4  # a = 3 -> 4 breaks the multiplication gate
    with pytest.raises(JP.PlonkError, match="WrongQuotientPolyDegree"):
        JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    cs.witness[2] = 3
    other = build(JP.PlonkCircuit(0), 200)
    with pytest.raises(JP.PlonkError, match="domain size"):
        JP.PlonkKzgSnark.prove(JP.H.test_rng(), other, pk)
    small = JP.PlonkKzgSnark.universal_setup_for_testing(0, n, BETA)
    with pytest.raises(JP.PlonkError, match="IndexTooLarge"):
        JP.PlonkKzgSnark.preprocess(small, cs)


@pytest.mark.parametrize("log_gates", [16, 20, 22])
def test_bench_circuit_proof_verifies(hip, log_gates):
    """gen_circuit_for_bench shape (plonk/benches/bench.rs:29-46: a chain of additions), 2^16, BASELINE config 4's 2^20
    and 2^22 gates (quotient domain 2^25, 32 GiB of HBM), BN254: the proof computed on the device verifies under the
    restated verifier (size-independent property; the oracle prover would take hours at this size, its verifier is
    O(1)).  tests/manual_prove_big.py runs the same check at 2^24 gates (128 GiB, 0.71 s per proof)."""
    from jellyfish_amd import plonk as JP

    cv = R.BN254
    cs = JP.PlonkCircuit.new_turbo_plonk(0)
    a = cs.zero()
    for _ in range((1 << log_gates) - 10):
        a = cs.add(a, cs.one())
    cs.finalize_for_arithmetization()
    n = cs.eval_domain_size()
    assert n == 1 << log_gates
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, n + 2, BETA)
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    proof = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk)
    assert P.verify(cv, vk.as_dict(), [], proof.as_dict(), trapdoor_beta=BETA)
    assert len(proof.serialize_compressed()) == 769
    # the bench circuit uses 4 of the 13 selectors and has no public input: prove() fills the coset evaluations of the
    # identically-zero polynomials instead of transforming them; transforming all 25 like the reference gives the same bytes
    assert sum(pk.selector_is_zero) == 9
    if log_gates > 20:
        return
    pk.keep_lagrange_commit_key()   # the wires of this circuit hold counters, ones and zeros
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == proof.serialize_compressed()
    pk.drop_lagrange_commit_key()
    JP.SKIP_ZERO_POLYS = False
    try:
        assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == proof.serialize_compressed()
    finally:
        JP.SKIP_ZERO_POLYS = True
    pk.keep_coset_evals()
    assert JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed() == proof.serialize_compressed()


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_batch_prove_matches_oracle_and_verifies(hip, cv):
    """PlonkKzgSnark::batch_prove (snark.rs:61-74 -> 163-430; the reference's test_key_aggregation_and_batch_prove
    :1567-1721 protocol): 3 different circuits of one domain size -> ONE BatchProof, byte-identical to the restated
    prover on the same RNG stream, accepted by the restated batch verifier, rejected for a wrong public input or a
    swapped instance order; one instance gives the bytes of prove() (impl From<Proof> for BatchProof)."""
    from jellyfish_amd import plonk as JP

    sizes = [60, 80, 100]
    ocss = [build(P.PlonkCircuit(cv), s) for s in sizes]
    css = [build(JP.PlonkCircuit(cv.curve_id), s) for s in sizes]
    n = css[0].eval_domain_size()
    assert all(c.eval_domain_size() == n for c in css) and all(o.n == n for o in ocss)
    beta = BETA % cv.fr.p
    osrs = P.universal_setup_for_testing(cv, n + 2, beta)
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(cv.curve_id, n + 2, beta)
    okeys = [P.preprocess(cv, osrs, o) for o in ocss]
    keys = [JP.PlonkKzgSnark.preprocess(srs, c) for c in css]
    pks, vks = [k[0] for k in keys], [k[1].as_dict() for k in keys]
    pubs = [c.public_input() for c in css]

    bp = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, pks, extra_transcript_init_msg=b"batch")
    obp, _ = P.batch_prove(cv, P.test_rng(), ocss, [k[0] for k in okeys], extra_msg=b"batch")
    assert len(bp) == 3
    assert bp.as_dict() == obp
    assert bp.serialize_compressed() == P.ser_batch_proof(cv, obp)
    assert P.verify_batch_proof(cv, vks, pubs, bp.as_dict(), extra_msg=b"batch", trapdoor_beta=beta)
    assert not P.verify_batch_proof(cv, vks, pubs, bp.as_dict(), trapdoor_beta=beta)
    bad = [list(p) for p in pubs]
    bad[1][0] = (bad[1][0] + 1) % cv.fr.p
    assert not P.verify_batch_proof(cv, vks, bad, bp.as_dict(), extra_msg=b"batch", trapdoor_beta=beta)
    assert not P.verify_batch_proof(cv, vks[::-1], pubs[::-1], bp.as_dict(), extra_msg=b"batch", trapdoor_beta=beta)
    # proving-key-resident coset evaluations on some instances only: same bytes
    pks[1].keep_coset_evals()
    again = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, pks, extra_transcript_init_msg=b"batch")
    assert again.serialize_compressed() == bp.serialize_compressed()
    # Lagrange-basis round-1 commitments on some instances only: same bytes
    pks[0].keep_lagrange_commit_key()
    pks[2].keep_lagrange_commit_key()
    again = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, pks, extra_transcript_init_msg=b"batch")
    assert again.serialize_compressed() == bp.serialize_compressed()
    pks[0].drop_lagrange_commit_key()
    pks[2].drop_lagrange_commit_key()
    # one instance: the BatchProof of prove()'s Proof
    one = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css[:1], pks[:1])
    single = JP.PlonkKzgSnark.prove(JP.H.test_rng(), css[0], pks[0])
    assert one.to_proof().serialize_compressed() == single.serialize_compressed()
    assert one.as_dict() == P.batch_proof_from_proof(single.as_dict())
    assert P.verify_batch_proof(cv, vks[:1], pubs[:1], one.as_dict(), trapdoor_beta=beta)
    assert P.verify(cv, vks[0], pubs[0], single.as_dict(), trapdoor_beta=beta)
    with pytest.raises(JP.PlonkError, match="number of circuits"):
        JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, pks[:2])
    with pytest.raises(JP.PlonkError, match="zero number"):
        JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), [], [])


def test_batch_prove_bench_circuits_verifies(hip):
    """4 instances of the bench circuit at 2^16 gates (BN254): the aggregated proof verifies (size-independent property)"""
    from jellyfish_amd import plonk as JP

    cv = R.BN254
    css = []
    for k in range(4):
        cs = JP.PlonkCircuit.new_turbo_plonk(0)
        a = cs.create_variable(k + 2)
        for _ in range((1 << 16) - 10):
            a = cs.add(a, cs.one())
        cs.finalize_for_arithmetization()
        css.append(cs)
    n = css[0].eval_domain_size()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, n + 2, BETA)
    keys = [JP.PlonkKzgSnark.preprocess(srs, c) for c in css]
    bp = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, [k[0] for k in keys])
    assert P.verify_batch_proof(cv, [k[1].as_dict() for k in keys], [[]] * 4, bp.as_dict(), trapdoor_beta=BETA)
