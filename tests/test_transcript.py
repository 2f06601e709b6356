"""Host glue of the prover: Merlin transcript, RNG / field sampling, serialisation (CPU; no GPU compute)."""
import os

import pytest

from oracle import merlin as OM
from oracle import plonk_ref as P
from oracle import pyref as R

MERLIN_KAT = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"

# Coset representatives k_1..k_4 of the BN254 scalar field as Espresso's public PlonkVerifier contract hard-codes them
# (COSET_K1..K4; not a file of /root/reference).  They are the first accepted draws of
# compute_coset_representatives (relation/src/constants.rs:30-84), so they pin the ChaCha20Rng word stream and the
# ark-ff `Fp::rand` restatement (limb order, top-bit shaving, Montgomery interpretation) at the byte level.
BN254_COSET_K = [
    1,
    0x2F8DD1F1A7583C42C4E12A44E110404C73CA6C94813F85835DA4FB7BB1301D4A,
    0x1EE678A0470A75A6EAA8FE837060498BA828A3703B311D0F77F010424AFEB025,
    0x2042A587A90C187B0A087C03E29C968B950B1DB26D5C82D666905A6895790C0A,
    0x2E2B91456103698ADF57B799969DEA1C8F739DA5D8D40DD3EB9222DB7C81E881,
]


def test_merlin_known_answer_oracle():
    t = OM.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == MERLIN_KAT


def test_keccak_matches_sha3():
    import hashlib

    def sha3_256(msg):
        rate, st = 136, bytearray(200)
        m = bytearray(msg) + b"\x06"
        m += b"\x00" * ((-len(m)) % rate)
        m[-1] ^= 0x80
        for i in range(0, len(m), rate):
            for j in range(rate):
                st[j] ^= m[i + j]
            OM.keccak_f1600(st)
        return bytes(st[:32])

    for msg in (b"", b"abc", b"q" * 500):
        assert sha3_256(msg) == hashlib.sha3_256(msg).digest()


def test_merlin_product_matches_oracle():
    """the product's transcript (Keccak in libjfhip host code) against the oracle and the known answer"""
    from jellyfish_amd import lib as L
    from jellyfish_amd.transcript import MerlinTranscript

    L.load_library()
    t = MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == MERLIN_KAT
    a, b = MerlinTranscript(b"PlonkProof"), OM.MerlinTranscript(b"PlonkProof")
    for i in range(40):
        msg = os.urandom(1 + 37 * i % 400)
        a.append_message(b"label%d" % i, msg)
        b.append_message(b"label%d" % i, msg)
        if i % 7 == 0:
            assert a.challenge_bytes(b"c", 64) == b.challenge_bytes(b"c", 64)


def test_coset_representatives_public_constants():
    assert P.coset_representatives(R.BN254, 5, 1 << 10) == BN254_COSET_K
    from jellyfish_amd import hostglue as H

    assert H.compute_coset_representatives(H.BN254, 5, 1 << 10) == BN254_COSET_K
    assert H.compute_coset_representatives(H.BLS12_381, 6, 64) == P.coset_representatives(R.BLS12_381, 6, 64)


def test_hostglue_matches_oracle():
    from jellyfish_amd import hostglue as H

    for cv, cp in ((R.BN254, H.BN254), (R.BLS12_381, H.BLS12_381)):
        assert (cp.q, cp.r, cp.g1, cp.g2) == (cv.fq.p, cv.fr.p, cv.g, P.G2_GEN[cv.name])
        assert P.g2_on_curve(cv, cp.g2)
        a, b = H.test_rng(), P.test_rng()
        xs = [H.fr_rand(cp, a) for _ in range(300)]
        assert xs == [P.fr_rand(cv, b) for _ in range(300)]
        assert H.fr_from_mont(cp, H.fr_to_mont(cp, xs)) == xs
        for k in xs[:3]:
            pt = R.ec_mul(cv, k, cv.g)
            assert H.ser_g1(cp, pt) == P.ser_g1(cv, pt)
            assert (H.g1_from_record(cp, H.g1_to_mont(cp, pt).tobytes() + bytes(4))) == pt
            q2 = P.g2_mul(cv, k, P.G2_GEN[cv.name])
            assert H.g2_mul(cp, k, cp.g2) == q2 and P.g2_on_curve(cv, q2)
            assert H.ser_g2(cp, q2) == P.ser_g2(cv, q2)
        assert H.ser_g1(cp, None) == P.ser_g1(cv, None)
        # group order annihilates the G2 generator
        assert P.g2_add(cv, P.g2_mul(cv, cv.fr.p - 1, cp.g2), cp.g2) is None


def _circuit(new, cv_p):
    cs = new()
    a = cs.create_variable(3)
    b = cs.create_public_variable(5)
    c = cs.mul(a, b)
    d = cs.add(c, a)
    e = cs.create_public_variable(18)
    cs.enforce_equal(d, e)
    f = cs.power5_gate(a)
    cs.enforce_constant(f, 243)
    cs.enforce_bool(cs.sub(f, cs.create_variable(242)))
    for _ in range(9):
        d = cs.add(d, 1)
    cs.finalize_for_arithmetization()
    return cs


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_oracle_prove_verify(cv):
    """the restated prover against the restated verifier (prove -> verify, snark.rs tests :852-1064), and rejections"""
    cs = _circuit(lambda: P.PlonkCircuit(cv), cv.fr.p)
    assert cs.check_circuit_satisfiability(cs.public_input())
    beta = 0x1234567890ABCDEF1234567890ABCDEF % cv.fr.p
    srs = P.universal_setup_for_testing(cv, cs.n + 2, beta)
    pk, vk = P.preprocess(cv, srs, cs)
    proof, _, orc = P.prove(cv, P.test_rng(), cs, pk)
    assert P.verify(cv, vk, cs.public_input(), proof, trapdoor_beta=beta)
    # split quotient recombines: sum_i X^{i(n+2)} t_i has the blinding cancel (prover.rs:922-940)
    bad = dict(proof)
    bad["perm_next_eval"] = (bad["perm_next_eval"] + 1) % cv.fr.p
    assert not P.verify(cv, vk, cs.public_input(), bad, trapdoor_beta=beta)
    assert not P.verify(cv, vk, [5, 19], proof, trapdoor_beta=beta)
    bad = dict(proof)
    bad["opening_proof"] = R.ec_add(cv, proof["opening_proof"], cv.g)
    assert not P.verify(cv, vk, cs.public_input(), bad, trapdoor_beta=beta)
    assert not P.verify(cv, vk, cs.public_input(), proof, extra_msg=b"x", trapdoor_beta=beta)
    assert len(P.ser_proof(cv, proof)) == 8 + 5 * 8 * cv.fq.limbs64 + 8 * cv.fq.limbs64 + 8 + 5 * 8 * cv.fq.limbs64 \
        + 2 * 8 * cv.fq.limbs64 + 8 + 5 * 32 + 8 + 4 * 32 + 32 + 1


def _circuit_k(new, extra):
    cs = new()
    a = cs.create_variable(3)
    b = cs.create_public_variable(5 + extra)
    c = cs.mul(a, b)
    for _ in range(5 + extra):
        c = cs.add(c, a)
    cs.finalize_for_arithmetization()
    return cs


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_oracle_batch_prove_verify(cv):
    """restated batch_prove (snark.rs:163-430) against the restated batch verifier (verifier.rs:70-257): an aggregated
    proof over 3 instances verifies; one instance reproduces prove(); tampering and instance reordering are rejected"""
    css = [_circuit_k(lambda: P.PlonkCircuit(cv), e) for e in range(3)]
    n = css[0].n
    assert all(c.n == n for c in css)
    beta = 0x1234567890ABCDEF1234567890ABCDEF % cv.fr.p
    srs = P.universal_setup_for_testing(cv, n + 2, beta)
    keys = [P.preprocess(cv, srs, c) for c in css]
    pks, vks = [k[0] for k in keys], [k[1] for k in keys]
    pubs = [c.public_input() for c in css]
    bp, _ = P.batch_prove(cv, P.test_rng(), css, pks)
    assert P.verify_batch_proof(cv, vks, pubs, bp, trapdoor_beta=beta)
    assert not P.verify_batch_proof(cv, vks[::-1], pubs[::-1], bp, trapdoor_beta=beta)
    bad = dict(bp)
    bad["poly_evals_vec"] = [dict(e) for e in bp["poly_evals_vec"]]
    bad["poly_evals_vec"][2]["perm_next_eval"] = (bad["poly_evals_vec"][2]["perm_next_eval"] + 1) % cv.fr.p
    assert not P.verify_batch_proof(cv, vks, pubs, bad, trapdoor_beta=beta)
    wrong = [list(p) for p in pubs]
    wrong[0][0] += 1
    assert not P.verify_batch_proof(cv, vks, wrong, bp, trapdoor_beta=beta)
    one, _ = P.batch_prove(cv, P.test_rng(), css[:1], pks[:1])
    proof, _, _ = P.prove(cv, P.test_rng(), css[0], pks[0])
    assert one == P.batch_proof_from_proof(proof)
    k, q = 3, 8 * cv.fq.limbs64
    assert len(P.ser_batch_proof(cv, bp)) == 8 + k * (8 + 5 * q) + 8 + k * q + 8 + k * (8 + 5 * 32 + 8 + 4 * 32 + 32) \
        + 8 + k + 8 + 5 * q + 2 * q


def test_product_circuit_matches_oracle_circuit():
    """the numpy circuit finalisation of the product against the list-based restatement"""
    from jellyfish_amd import plonk as JP

    for cv in (R.BN254, R.BLS12_381):
        a = _circuit(lambda: P.PlonkCircuit(cv), None)
        b = _circuit(lambda: JP.PlonkCircuit(cv.curve_id), None)
        n = a.n
        assert b.eval_domain_size() == n and b.k == a.k and b.public_input() == a.public_input()
        assert [list(map(int, row)) for row in b.wire_variables] == a.wire_variables
        assert [int(x) for x in b.wire_permutation] == [w * n + g for (w, g) in a.wire_permutation]
        assert [list(b.gate_kinds[int(k)]) for k in b.gate_kind] == a.gates
        b.check_circuit_satisfiability(b.public_input())
        with pytest.raises(JP.PlonkError):
            b.check_circuit_satisfiability([5, 19])
        with pytest.raises(JP.PlonkError):
            b.create_variable(1)


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_oracle_ultraplonk_prove_verify(cv):
    """the restated UltraPlonk (Plookup) prover against the restated verifier -- two restatements of different reference
    files (prover.rs / verifier.rs) that must agree: range gates, two key-value tables, single proof and a 2-instance
    BatchProof; tampered Plookup evaluations, a wrong public input and a swapped instance order are rejected"""
    from tests.util import build_ultra_circuit

    css = [build_ultra_circuit(P.PlonkCircuit.new_ultra_plonk(cv, 3), 4, e) for e in (0, 1)]
    n = css[0].n
    assert all(c.n == n for c in css) and all(c.check_circuit_satisfiability(c.public_input()) for c in css)
    beta = 0x1234567890ABCDEF1234567890ABCDEF % cv.fr.p
    srs = P.universal_setup_for_testing(cv, n + 2, beta)
    keys = [P.preprocess(cv, srs, c) for c in css]
    pks, vks = [k[0] for k in keys], [k[1] for k in keys]
    pubs = [c.public_input() for c in css]
    proof, ch, orc = P.prove(cv, P.test_rng(), css[0], pks[0])
    assert ch["tau"] is not None and len(proof["wires_poly_comms"]) == 6 and len(proof["wire_sigma_evals"]) == 5
    assert P.verify(cv, vks[0], pubs[0], proof, trapdoor_beta=beta)
    # h_1 / h_2 interpolate the sorted vector; its halves overlap in one element (constraint_system.rs:1621-1622)
    tau = ch["tau"]
    srt = css[0].lookup_sorted_vec(tau, css[0].merged_lookup_table(tau))
    assert len(srt) == 2 * n - 1 and srt[n - 1] == srt[:n][-1] == srt[n - 1:][0]
    bad = dict(proof, plookup_proof=dict(proof["plookup_proof"], poly_evals=dict(proof["plookup_proof"]["poly_evals"])))
    bad["plookup_proof"]["poly_evals"]["prod_next_eval"] = (bad["plookup_proof"]["poly_evals"]["prod_next_eval"] + 1) % cv.fr.p
    assert not P.verify(cv, vks[0], pubs[0], bad, trapdoor_beta=beta)
    assert not P.verify(cv, vks[0], [pubs[0][0] + 1, pubs[0][1]], proof, trapdoor_beta=beta)
    q = 8 * cv.fq.limbs64
    assert len(P.ser_proof(cv, proof)) == (8 + 6 * q) + q + (8 + 6 * q) + 2 * q + (8 + 6 * 32) + (8 + 5 * 32) + 32 + 1 \
        + (8 + 2 * q) + q + 15 * 32
    bp, _ = P.batch_prove(cv, P.test_rng(), css, pks)
    assert P.verify_batch_proof(cv, vks, pubs, bp, trapdoor_beta=beta)
    assert not P.verify_batch_proof(cv, vks[::-1], pubs[::-1], bp, trapdoor_beta=beta)
    # an unsatisfied lookup has no sorted vector
    broken = build_ultra_circuit(P.PlonkCircuit.new_ultra_plonk(cv, 3), 4)
    broken.witness[broken.witness.index(114)] = 115
    assert not broken.check_circuit_satisfiability(broken.public_input())
    with pytest.raises(ValueError, match="sorted vector"):
        P.prove(cv, P.test_rng(), broken, pks[0])


@pytest.mark.parametrize("cv", [R.BN254, R.BLS12_381], ids=lambda c: c.name)
def test_ultra_circuit_finalisation_matches_oracle(cv):
    """pad / io gates to the front / lookup gates to the rear / permutation (constraint_system.rs:831-884, 1167-1187)"""
    from jellyfish_amd import plonk as JP
    from tests.util import build_ultra_circuit

    ocs = build_ultra_circuit(P.PlonkCircuit.new_ultra_plonk(cv, 3), 10)
    cs = build_ultra_circuit(JP.PlonkCircuit.new_ultra_plonk(cv.curve_id, 3), 10)
    n = ocs.n
    assert cs.eval_domain_size() == n and cs.k == ocs.k and cs.public_input() == ocs.public_input()
    assert [list(map(int, row)) for row in cs.wire_variables] == ocs.wire_variables
    assert [int(x) for x in cs.wire_permutation] == [w * n + g for (w, g) in ocs.wire_permutation]
    kinds = [cs.gate_kinds[int(k)] for k in cs.gate_kind]
    assert [list(k[:13]) for k in kinds] == ocs.gates
    assert [tuple(k[13:]) for k in kinds] == ocs.lk
    assert ocs.check_circuit_satisfiability(ocs.public_input())
    cs.check_circuit_satisfiability(cs.public_input())
