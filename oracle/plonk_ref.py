"""Big-integer restatement of jellyfish's TurboPlonk prover, preprocessor and verifier -- the checker for
jellyfish_amd/plonk.py.

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py).  Follows, function by function:
  * circuit builder / finalisation     relation/src/constraint_system.rs:406-470, 652-790, 831-884, 943-981,
                                       1116-1187; gates relation/src/gates/arithmetic.rs
  * coset representatives k            relation/src/constants.rs:30-84
  * arithmetisation                    relation/src/constraint_system.rs:1359-1456
  * preprocess                         plonk/src/proof_system/snark.rs:492-580
  * prover rounds 1-5                  plonk/src/proof_system/snark.rs:163-430, prover.rs:77-413, 457-488, 491-661,
                                       894-1015
  * transcript                         plonk/src/transcript/mod.rs:46-186, standard.rs:14-39 (Merlin: oracle/merlin.py)
  * verifier                           plonk/src/proof_system/verifier.rs:70-257, 262-339, 358-432, 439-535, 541-713,
                                       718-788, 883-912
External pieces restated from their published algorithms (crates absent from /root/reference, "ark, unverified
here"): rand_chacha 0.3 ChaCha{12,20}Rng word stream, ark-ff 0.5 `Fp::rand` (limbs from next_u64, top bits shaved,
rejection; the draw is the Montgomery representation), ark-serialize 0.5 compressed encodings (BN254: little-endian x
with flag bits in the last byte; BLS12-381: zcash big-endian format).  No reference fixture holds proof bytes, so at
the byte level versus arkworks the status is "parity unpinned"; what pins this file is that its proofs VERIFY under its
restatement of the reference's verifier (a wrong prover or verifier restatement fails that check), the satisfied-circuit
identities, and Merlin's published known-answer vector.

The final pairing check e(A, [beta]h) = e(B, h) (verifier.rs:201-257) is evaluated here either by a real optimal-ate-free
Tate-style pairing on BN254 (oracle/pairing.py, if present) or, equivalently for a test SRS whose trapdoor beta is known,
as the G1 equality [beta]A == B.
"""
from __future__ import annotations

from . import pyref as R
from .merlin import MerlinTranscript

GATE_WIDTH = 4
N_SELECTORS_TURBO = 13  # q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc

# ------------------------------------------------------------------------------------------------
# RNG: rand_chacha block stream + ark-ff field sampling
# ------------------------------------------------------------------------------------------------


class ChaChaRng:
    """rand_chacha ChaCha{8,12,20}Rng: 32-byte key, 64-bit block counter (words 12, 13), 64-bit stream id 0; words are
    consumed in order; next_u64 = lo | hi << 32 of two consecutive words (rand_core BlockRng with a 64-word buffer)."""

    def __init__(self, seed: bytes, rounds: int = 20):
        assert len(seed) == 32
        self.key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
        self.rounds = rounds
        self.counter = 0
        self.buf: list[int] = []
        self.idx = 0

    def _block(self, counter: int) -> list[int]:
        M = 0xFFFFFFFF
        s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + self.key + [counter & M, (counter >> 32) & M, 0, 0]
        x = list(s)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & M
            x[d] ^= x[a]
            x[d] = ((x[d] << 16) | (x[d] >> 16)) & M
            x[c] = (x[c] + x[d]) & M
            x[b] ^= x[c]
            x[b] = ((x[b] << 12) | (x[b] >> 20)) & M
            x[a] = (x[a] + x[b]) & M
            x[d] ^= x[a]
            x[d] = ((x[d] << 8) | (x[d] >> 24)) & M
            x[c] = (x[c] + x[d]) & M
            x[b] ^= x[c]
            x[b] = ((x[b] << 7) | (x[b] >> 25)) & M

        for _ in range(self.rounds // 2):
            qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15)
            qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14)
        return [(x[i] + s[i]) & M for i in range(16)]

    def _refill(self):
        self.buf = []
        for _ in range(4):
            self.buf += self._block(self.counter)
            self.counter += 1
        self.idx = 0

    def next_u32(self) -> int:
        if self.idx >= len(self.buf):
            self._refill()
        v = self.buf[self.idx]
        self.idx += 1
        return v

    def next_u64(self) -> int:
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)


def test_rng() -> ChaChaRng:
    """jf_utils::test_rng (utilities/src/lib.rs:88-96): StdRng (= ChaCha12Rng in rand 0.8) from a fixed seed."""
    seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
    return ChaChaRng(seed, rounds=12)


def fr_rand(cv: R.Curve, rng) -> int:
    """ark-ff `Fp::rand`: N u64 limbs, shave the unused top bits, reject if >= p; the accepted limbs ARE the
    Montgomery representation.  Returns the canonical value."""
    f = cv.fr
    shave = 64 * f.limbs64 - f.p.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(f.limbs64)]
        limbs[-1] &= (1 << (64 - shave)) - 1
        v = R.Field.from_limbs(limbs)
        if v < f.p:
            return f.from_mont(v)


def coset_representatives(cv: R.Curve, num_wire_types: int, n: int) -> list[int]:
    """compute_coset_representatives (relation/src/constants.rs:30-84)"""
    r = cv.fr.p
    rng = ChaChaRng(bytes(32), rounds=20)
    ks, pows = [], []
    for i in range(num_wire_types):
        if i == 0:
            ks.append(1)
            pows.append(1)
            continue
        while True:
            nxt = fr_rand(cv, rng)
            pw = pow(nxt, n, r)
            if all(pow(q, -1, r) * pw % r != 1 for q in pows):
                break
        ks.append(nxt)
        pows.append(pw)
    return ks


# ------------------------------------------------------------------------------------------------
# G2 (for the "SRS G2 element" transcript message and the pairing check) and ark-serialize encodings
# ------------------------------------------------------------------------------------------------

G2_GEN = {
    "bn254": ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
               11559732032986387107991004021392285783925812861821192530917403151452391805634),
              (8495653923123431417604973247489272438418190587263600148770280649306958101930,
               4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    "bls12_381": ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
                   0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
                  (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
                   0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)),
}


def _f2mul(p, a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def _f2inv(p, a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, p)
    return (a[0] * d % p, -a[1] * d % p)


def g2_b(cv: R.Curve):
    p = cv.fq.p
    if cv.name == "bn254":
        return _f2mul(p, (3, 0), _f2inv(p, (9, 1)))  # D-twist: 3 / (9 + u)
    return (4, 4)  # M-twist: 4 (1 + u)


def g2_on_curve(cv: R.Curve, P) -> bool:
    if P is None:
        return True
    p = cv.fq.p
    x, y = P
    x3 = _f2mul(p, _f2mul(p, x, x), x)
    b = g2_b(cv)
    y2 = _f2mul(p, y, y)
    return y2 == ((x3[0] + b[0]) % p, (x3[1] + b[1]) % p)


def g2_add(cv: R.Curve, P, Q):
    p = cv.fq.p
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if ((y1[0] + y2[0]) % p, (y1[1] + y2[1]) % p) == (0, 0):
            return None
        num = _f2mul(p, (3, 0), _f2mul(p, x1, x1))
        lam = _f2mul(p, num, _f2inv(p, ((2 * y1[0]) % p, (2 * y1[1]) % p)))
    else:
        lam = _f2mul(p, ((y2[0] - y1[0]) % p, (y2[1] - y1[1]) % p), _f2inv(p, ((x2[0] - x1[0]) % p, (x2[1] - x1[1]) % p)))
    l2 = _f2mul(p, lam, lam)
    x3 = ((l2[0] - x1[0] - x2[0]) % p, (l2[1] - x1[1] - x2[1]) % p)
    t = _f2mul(p, lam, ((x1[0] - x3[0]) % p, (x1[1] - x3[1]) % p))
    return (x3, ((t[0] - y1[0]) % p, (t[1] - y1[1]) % p))


def g2_mul(cv: R.Curve, k: int, P):
    k %= cv.fr.p
    acc = None
    while k:
        if k & 1:
            acc = g2_add(cv, acc, P)
        P = g2_add(cv, P, P)
        k >>= 1
    return acc


def ser_fr(cv: R.Curve, x: int) -> bytes:
    return (x % cv.fr.p).to_bytes(32, "little")


def ser_g1(cv: R.Curve, P) -> bytes:
    """ark-serialize compressed G1Affine."""
    p = cv.fq.p
    nb = 8 * cv.fq.limbs64
    if cv.name == "bn254":
        if P is None:
            b = bytearray(nb)
            b[-1] |= 0x40
            return bytes(b)
        b = bytearray(P[0].to_bytes(nb, "little"))
        if P[1] > (p - P[1]) % p:
            b[-1] |= 0x80
        return bytes(b)
    if P is None:
        b = bytearray(nb)
        b[0] |= 0xC0
        return bytes(b)
    b = bytearray(P[0].to_bytes(nb, "big"))
    b[0] |= 0x80
    if P[1] > (p - P[1]) % p:
        b[0] |= 0x20
    return bytes(b)


def _f2_gt(a, b) -> bool:
    """ark QuadExtField ordering: c1 first, then c0"""
    return (a[1], a[0]) > (b[1], b[0])


def ser_g2(cv: R.Curve, P) -> bytes:
    p = cv.fq.p
    nb = 8 * cv.fq.limbs64
    assert P is not None
    x, y = P
    neg_y = ((-y[0]) % p, (-y[1]) % p)
    if cv.name == "bn254":
        b = bytearray(x[0].to_bytes(nb, "little") + x[1].to_bytes(nb, "little"))
        if _f2_gt(y, neg_y):
            b[-1] |= 0x80
        return bytes(b)
    b = bytearray(x[1].to_bytes(nb, "big") + x[0].to_bytes(nb, "big"))
    b[0] |= 0x80
    if _f2_gt(y, neg_y):
        b[0] |= 0x20
    return bytes(b)


def ser_proof(cv: R.Curve, proof: dict) -> bytes:
    """CanonicalSerialize (compressed) of `Proof` (plonk/src/proof_system/structs.rs:54-76), TurboPlonk"""
    out = bytearray()

    def vec_g1(v):
        out.extend(len(v).to_bytes(8, "little"))
        for P in v:
            out.extend(ser_g1(cv, P))

    def vec_fr(v):
        out.extend(len(v).to_bytes(8, "little"))
        for x in v:
            out.extend(ser_fr(cv, x))

    vec_g1(proof["wires_poly_comms"])
    out.extend(ser_g1(cv, proof["prod_perm_poly_comm"]))
    vec_g1(proof["split_quot_poly_comms"])
    out.extend(ser_g1(cv, proof["opening_proof"]))
    out.extend(ser_g1(cv, proof["shifted_opening_proof"]))
    vec_fr(proof["wires_evals"])
    vec_fr(proof["wire_sigma_evals"])
    out.extend(ser_fr(cv, proof["perm_next_eval"]))
    out.append(0)  # plookup_proof: None
    return bytes(out)


def ser_batch_proof(cv: R.Curve, bp: dict) -> bytes:
    """CanonicalSerialize (compressed) of `BatchProof` (plonk/src/proof_system/structs.rs:227-249), TurboPlonk
    instances (plookup_proofs_vec = [None; k])"""
    out = bytearray()

    def vec_g1(v):
        out.extend(len(v).to_bytes(8, "little"))
        for P in v:
            out.extend(ser_g1(cv, P))

    def vec_fr(v):
        out.extend(len(v).to_bytes(8, "little"))
        for x in v:
            out.extend(ser_fr(cv, x))

    k = len(bp["prod_perm_poly_comms_vec"])
    out.extend(k.to_bytes(8, "little"))
    for comms in bp["wires_poly_comms_vec"]:
        vec_g1(comms)
    vec_g1(bp["prod_perm_poly_comms_vec"])
    out.extend(k.to_bytes(8, "little"))
    for ev in bp["poly_evals_vec"]:
        vec_fr(ev["wires_evals"])
        vec_fr(ev["wire_sigma_evals"])
        out.extend(ser_fr(cv, ev["perm_next_eval"]))
    out.extend(k.to_bytes(8, "little"))
    out.extend(b"\x00" * k)  # plookup_proofs_vec: None each
    vec_g1(bp["split_quot_poly_comms"])
    out.extend(ser_g1(cv, bp["opening_proof"]))
    out.extend(ser_g1(cv, bp["shifted_opening_proof"]))
    return bytes(out)


def batch_proof_from_proof(proof: dict) -> dict:
    """impl From<Proof> for BatchProof (structs.rs:275-287)"""
    return {"wires_poly_comms_vec": [proof["wires_poly_comms"]], "prod_perm_poly_comms_vec": [proof["prod_perm_poly_comm"]],
            "poly_evals_vec": [{"wires_evals": proof["wires_evals"], "wire_sigma_evals": proof["wire_sigma_evals"],
                                "perm_next_eval": proof["perm_next_eval"]}],
            "split_quot_poly_comms": proof["split_quot_poly_comms"], "opening_proof": proof["opening_proof"],
            "shifted_opening_proof": proof["shifted_opening_proof"]}


# ------------------------------------------------------------------------------------------------
# circuit
# ------------------------------------------------------------------------------------------------


def _gate(q_lc=(0, 0, 0, 0), q_mul=(0, 0), q_hash=(0, 0, 0, 0), q_o=0, q_c=0, q_ecc=0):
    return list(q_lc) + list(q_mul) + list(q_hash) + [q_o, q_c, q_ecc]


class PlonkCircuit:
    """TurboPlonk subset of `PlonkCircuit` (constraint_system.rs:347-470, 558-790)"""

    def __init__(self, cv: R.Curve):
        self.cv = cv
        self.r = cv.fr.p
        self.num_wire_types = GATE_WIDTH + 1
        self.witness = [0, 1]
        self.gates: list[list[int]] = []           # 13 selector values per gate
        self.is_io: list[bool] = []
        self.wire_variables = [[] for _ in range(GATE_WIDTH + 1)]
        self.pub_input_gate_ids: list[int] = []
        self.n = 0                                  # eval domain size once finalised
        self.enforce_constant(0, 0)
        self.enforce_constant(1, 1)

    # -- builder ---------------------------------------------------------------------------------
    def _insert_gate(self, wire_vars, sel, io=False):
        assert self.n == 0, "ModifyFinalizedCircuit"
        for w, v in zip(self.wire_variables, wire_vars):
            w.append(v)
        self.gates.append(sel)
        self.is_io.append(io)

    def num_gates(self):
        return len(self.gates)

    def create_variable(self, val: int) -> int:
        assert self.n == 0, "ModifyFinalizedCircuit"
        self.witness.append(val % self.r)
        return len(self.witness) - 1

    def create_public_variable(self, val: int) -> int:
        v = self.create_variable(val)
        self.pub_input_gate_ids.append(self.num_gates())
        self._insert_gate([0, 0, 0, 0, v], _gate(q_o=1), io=True)
        return v

    def enforce_constant(self, var: int, c: int):
        self._insert_gate([0, 0, 0, 0, var], _gate(q_c=c % self.r, q_o=1))

    def add(self, a: int, b: int) -> int:
        c = self.create_variable(self.witness[a] + self.witness[b])
        self._insert_gate([a, b, 0, 0, c], _gate(q_lc=(1, 1, 0, 0), q_o=1))
        return c

    def sub(self, a: int, b: int) -> int:
        c = self.create_variable(self.witness[a] - self.witness[b])
        self._insert_gate([a, b, 0, 0, c], _gate(q_lc=(1, self.r - 1, 0, 0), q_o=1))
        return c

    def mul(self, a: int, b: int) -> int:
        c = self.create_variable(self.witness[a] * self.witness[b])
        self._insert_gate([a, b, 0, 0, c], _gate(q_mul=(1, 0), q_o=1))
        return c

    def enforce_bool(self, a: int):
        self._insert_gate([a, a, 0, 0, a], _gate(q_mul=(1, 0), q_o=1))

    def enforce_equal(self, a: int, b: int):
        self._insert_gate([a, b, 0, 0, 0], _gate(q_lc=(1, self.r - 1, 0, 0), q_o=1))

    def power5_gate(self, a: int) -> int:
        """FifthRootGate shape (q_hash[0] = 1): c = a^5"""
        c = self.create_variable(pow(self.witness[a], 5, self.r))
        self._insert_gate([a, 0, 0, 0, c], _gate(q_hash=(1, 0, 0, 0), q_o=1))
        return c

    def public_input(self):
        return [self.witness[self.wire_variables[GATE_WIDTH][g]] for g in self.pub_input_gate_ids]

    # -- finalisation ----------------------------------------------------------------------------
    def finalize_for_arithmetization(self):
        if self.n:
            return
        n = 1
        while n < self.num_gates():
            n <<= 1
        self.n = n
        self.log_n = n.bit_length() - 1
        # pad (constraint_system.rs:874-884): PaddingGate = all-zero selectors, wires = zero variable
        while len(self.gates) < n:
            self.gates.append(_gate())
            self.is_io.append(False)
        for w in self.wire_variables:
            w.extend([0] * (n - len(w)))
        # rearrange_gates (:831-844): io gates to the front
        for gate_id in range(len(self.pub_input_gate_ids)):
            io_gate_id = self.pub_input_gate_ids[gate_id]
            if io_gate_id > gate_id:
                self.gates[gate_id], self.gates[io_gate_id] = self.gates[io_gate_id], self.gates[gate_id]
                self.is_io[gate_id], self.is_io[io_gate_id] = self.is_io[io_gate_id], self.is_io[gate_id]
                for w in self.wire_variables:
                    w[gate_id], w[io_gate_id] = w[io_gate_id], w[gate_id]
                self.pub_input_gate_ids[gate_id] = gate_id
        # compute_wire_permutation (:943-981)
        W = self.num_wire_types
        var_wires = [[] for _ in range(len(self.witness))]
        for wid in range(W):
            for gid, var in enumerate(self.wire_variables[wid]):
                var_wires[var].append((wid, gid))
        self.wire_permutation = [(0, 0)] * (W * n)
        for wires in var_wires:
            for a, b in zip(wires, wires[1:] + wires[:1]):
                self.wire_permutation[a[0] * n + a[1]] = b
        # compute_extended_id_permutation (:1116-1132)
        self.k = coset_representatives(self.cv, W, n)
        g = R.domain_root(self.cv, self.log_n)
        elems = [1] * n
        for j in range(1, n):
            elems[j] = elems[j - 1] * g % self.r
        self.extended_id_permutation = [ki * e % self.r for ki in self.k for e in elems]

    def check_circuit_satisfiability(self, pub_input) -> bool:
        r = self.r
        pis = {g: v for g, v in zip(self.pub_input_gate_ids, pub_input)}
        for gid, sel in enumerate(self.gates):
            w = [self.witness[self.wire_variables[i][gid]] for i in range(5)]
            pi = pis.get(gid, 0)
            lhs = (pi + sum(sel[i] * w[i] for i in range(4)) + sel[4] * w[0] * w[1] + sel[5] * w[2] * w[3]
                   + sum(sel[6 + i] * pow(w[i], 5, r) for i in range(4)) + sel[11]
                   + sel[12] * w[0] * w[1] * w[2] * w[3] * w[4]) % r
            if lhs != sel[10] * w[4] % r:
                return False
        return True

    # -- arithmetisation (:1359-1456) --------------------------------------------------------------
    def selector_values(self):
        return [[g[s] for g in self.gates] for s in range(N_SELECTORS_TURBO)]

    def extended_permutation(self):
        n = self.n
        return [self.extended_id_permutation[w * n + g] for (w, g) in self.wire_permutation]

    def wire_values(self):
        return [[self.witness[v] for v in self.wire_variables[i]] for i in range(self.num_wire_types)]

    def pub_input_values(self):
        v = [0] * self.n
        for g in self.pub_input_gate_ids:
            v[g] = self.witness[self.wire_variables[GATE_WIDTH][g]]
        return v

    def prod_permutation_values(self, beta, gamma):
        r, n, W = self.r, self.n, self.num_wire_types
        prod = [1]
        for j in range(n - 1):
            a = b = 1
            for i in range(W):
                wv = self.witness[self.wire_variables[i][j]]
                tmp = wv + gamma
                a = a * (tmp + beta * self.extended_id_permutation[i * n + j]) % r
                pi, pj = self.wire_permutation[i * n + j]
                b = b * (tmp + beta * self.extended_id_permutation[pi * n + pj]) % r
            prod.append(prod[-1] * a % r * pow(b, -1, r) % r)
        return prod


# ------------------------------------------------------------------------------------------------
# SRS, preprocess
# ------------------------------------------------------------------------------------------------


def universal_setup_for_testing(cv: R.Curve, max_degree: int, beta: int, g=None, h=None):
    """[beta^i]g, i <= max_degree; h, [beta]h  (snark.rs:447-488; g, h, beta are inputs here instead of RNG draws)"""
    g = g or cv.g
    h = h or G2_GEN[cv.name]
    return {"powers_of_g": R.srs_powers(cv, beta, g, max_degree + 1), "h": h, "beta_h": g2_mul(cv, beta, h), "g": g,
            "beta": beta}


def _commit(cv, srs, coeffs):
    return R.commit_via_beta(cv, srs["beta"], srs["g"], coeffs)


def preprocess(cv: R.Curve, srs: dict, cs: PlonkCircuit):
    n = cs.n
    assert len(srs["powers_of_g"]) >= n + 3, "IndexTooLarge"
    selectors = [R.ifft(cv, s, cs.log_n) for s in cs.selector_values()]
    ext = cs.extended_permutation()
    sigmas = [R.ifft(cv, ext[i * n:(i + 1) * n], cs.log_n) for i in range(cs.num_wire_types)]
    vk = {"domain_size": n, "num_inputs": len(cs.pub_input_gate_ids),
          "selector_comms": [_commit(cv, srs, p) for p in selectors],
          "sigma_comms": [_commit(cv, srs, p) for p in sigmas], "k": list(cs.k),
          "g": srs["g"], "h": srs["h"], "beta_h": srs["beta_h"]}
    pk = {"selectors": selectors, "sigmas": sigmas, "vk": vk, "srs": srs}
    return pk, vk


# ------------------------------------------------------------------------------------------------
# transcript
# ------------------------------------------------------------------------------------------------


class StandardTranscript:
    def __init__(self, cv: R.Curve, label: bytes = b"PlonkProof"):
        self.cv = cv
        self.t = MerlinTranscript(label)

    def append_message(self, label: bytes, msg: bytes):
        self.t.append_message(label, msg)

    def append_vk_and_pub_input(self, vk: dict, pub_input):
        cv = self.cv
        self.append_message(b"field size in bits", cv.fr.p.bit_length().to_bytes(4, "little"))
        self.append_message(b"domain size", vk["domain_size"].to_bytes(8, "little"))
        self.append_message(b"input size", vk["num_inputs"].to_bytes(8, "little"))
        self.append_message(b"SRS G2 element", ser_g2(cv, vk["beta_h"]))
        for k in vk["k"]:
            self.append_message(b"wire subsets separators", ser_fr(cv, k))
        for c in vk["selector_comms"]:
            self.append_message(b"selector commitments", ser_g1(cv, c))
        for c in vk["sigma_comms"]:
            self.append_message(b"sigma commitments", ser_g1(cv, c))
        for x in pub_input:
            self.append_message(b"public input", ser_fr(cv, x))

    def append_commitments(self, label: bytes, comms):
        for c in comms:
            self.append_message(label, ser_g1(self.cv, c))

    def append_field_elems(self, label: bytes, xs):
        for x in xs:
            self.append_message(label, ser_fr(self.cv, x))

    def get_challenge(self, label: bytes) -> int:
        return int.from_bytes(self.t.challenge_bytes(label, 64), "little") % self.cv.fr.p


# ------------------------------------------------------------------------------------------------
# prover
# ------------------------------------------------------------------------------------------------


def _mask(cv, rng, coeffs, n, hiding_bound):
    """mask_polynomial (prover.rs:457-466): poly + rand(hiding_bound)(X) * (X^n - 1)"""
    r = cv.fr.p
    b = [fr_rand(cv, rng) for _ in range(hiding_bound + 1)]
    out = list(coeffs) + [0] * (n + hiding_bound + 1 - len(coeffs))
    for i, bi in enumerate(b):
        out[i] = (out[i] - bi) % r
        out[n + i] = (out[n + i] + bi) % r
    return out


def _lincomb(r, terms, length):
    out = [0] * length
    for c, poly in terms:
        c %= r
        for i, x in enumerate(poly):
            out[i] = (out[i] + c * x) % r
    return out


def prove(cv: R.Curve, rng, cs: PlonkCircuit, pk: dict, extra_msg: bytes | None = None):
    """PlonkKzgSnark::prove for one TurboPlonk instance (snark.rs:587-613 -> 163-430).
    Returns (proof dict, challenges dict, oracles dict)."""
    r = cv.fr.p
    n, log_n, W = cs.n, cs.log_n, cs.num_wire_types
    srs, vk = pk["srs"], pk["vk"]
    m = 1
    while m < n * ((W * (n + 1) + 2) // n + 1):
        m <<= 1
    log_m = m.bit_length() - 1
    gen = cv.fr_generator
    tr = StandardTranscript(cv)
    if extra_msg is not None:
        tr.append_message(b"extra info", extra_msg)
    tr.append_vk_and_pub_input(vk, cs.public_input())
    # Round 1
    wire_polys = [_mask(cv, rng, R.ifft(cv, v, log_n), n, 1) for v in cs.wire_values()]
    wires_poly_comms = [_commit(cv, srs, p) for p in wire_polys]
    pub_inp_poly = R.ifft(cv, cs.pub_input_values(), log_n)
    tr.append_commitments(b"witness_poly_comms", wires_poly_comms)
    # Round 2
    beta = tr.get_challenge(b"beta")
    gamma = tr.get_challenge(b"gamma")
    prod_perm_poly = _mask(cv, rng, R.ifft(cv, cs.prod_permutation_values(beta, gamma), log_n), n, 2)
    prod_perm_poly_comm = _commit(cv, srs, prod_perm_poly)
    tr.append_commitments(b"perm_poly_comms", [prod_perm_poly_comm])
    # Round 3
    alpha = tr.get_challenge(b"alpha")
    sel_c = [R.fft(cv, p, log_m, gen) for p in pk["selectors"]]
    sig_c = [R.fft(cv, p, log_m, gen) for p in pk["sigmas"]]
    wir_c = [R.fft(cv, p, log_m, gen) for p in wire_polys]
    z_c = R.fft(cv, prod_perm_poly, log_m, gen)
    pi_c = R.fft(cv, pub_inp_poly, log_m, gen)
    quot_evals = R.quotient_evals_turbo(cv, log_n, log_m, sel_c, sig_c, wir_c, z_c, pi_c, cs.k, alpha, beta, gamma)
    quot = R.ifft(cv, quot_evals, log_m, gen)
    while quot and quot[-1] == 0:
        quot.pop()
    expected_degree = W * (n + 1) + 2
    if len(quot) - 1 != expected_degree:
        raise ValueError("WrongQuotientPolyDegree(%d, %d)" % (len(quot) - 1, expected_degree))
    split = []
    for i in range(W):
        end = (i + 1) * (n + 2) if i < W - 1 else len(quot)
        split.append(quot[i * (n + 2):end])
    last = 0
    for i in range(W - 1):
        now = fr_rand(cv, rng)
        split[i][0] = (split[i][0] - last) % r
        split[i].append(now)
        last = now
    split[W - 1][0] = (split[W - 1][0] - last) % r
    split_quot_poly_comms = [_commit(cv, srs, p) for p in split]
    tr.append_commitments(b"quot_poly_comms", split_quot_poly_comms)
    # Round 4
    zeta = tr.get_challenge(b"zeta")
    g = R.domain_root(cv, log_n)
    wires_evals = [R.poly_eval(cv, p, zeta) for p in wire_polys]
    wire_sigma_evals = [R.poly_eval(cv, p, zeta) for p in pk["sigmas"][:W - 1]]
    perm_next_eval = R.poly_eval(cv, prod_perm_poly, zeta * g % r)
    tr.append_field_elems(b"wire_evals", wires_evals)
    tr.append_field_elems(b"wire_sigma_evals", wire_sigma_evals)
    tr.append_field_elems(b"perm_next_eval", [perm_next_eval])
    # linearisation polynomial (prover.rs:299-352, 946-1015)
    vanish_eval = (pow(zeta, n, r) - 1) % r
    zeta_n2 = (vanish_eval + 1) * zeta % r * zeta % r
    terms = []
    coeff = 1
    for i, t in enumerate(split):
        if i:
            coeff = coeff * zeta_n2 % r
        terms.append((-vanish_eval * coeff, t))
    w = wires_evals
    sel = pk["selectors"]
    q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r), pow(w[2], 5, r),
                 pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
    terms += [(s, p) for s, p in zip(q_scalars, sel)]
    lagrange_1 = vanish_eval * pow(n * (zeta - 1), -1, r) % r
    c_z = alpha
    for j in range(W):
        c_z = c_z * (w[j] + beta * vk["k"][j] % r * zeta + gamma) % r
    c_z = (c_z + alpha * alpha % r * lagrange_1) % r
    terms.append((c_z, prod_perm_poly))
    c_s = alpha * beta % r * perm_next_eval % r
    for j in range(W - 1):
        c_s = c_s * (w[j] + beta * wire_sigma_evals[j] + gamma) % r
    terms.append((-c_s, pk["sigmas"][W - 1]))
    lin_poly = _lincomb(r, terms, n + 3)
    # Round 5
    v = tr.get_challenge(b"v")
    polys = [lin_poly] + wire_polys + pk["sigmas"][:W - 1]
    batch, c = [], 1
    for p in polys:
        batch.append((c, p))
        c = c * v % r
    batch_poly = _lincomb(r, batch, n + 3)
    q, _ = R.witness_poly(cv, batch_poly, zeta)
    opening_proof = _commit(cv, srs, q)
    q, _ = R.witness_poly(cv, prod_perm_poly, zeta * g % r)
    shifted_opening_proof = _commit(cv, srs, q)
    proof = {"wires_poly_comms": wires_poly_comms, "prod_perm_poly_comm": prod_perm_poly_comm,
             "split_quot_poly_comms": split_quot_poly_comms, "opening_proof": opening_proof,
             "shifted_opening_proof": shifted_opening_proof, "wires_evals": wires_evals,
             "wire_sigma_evals": wire_sigma_evals, "perm_next_eval": perm_next_eval}
    challenges = {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v}
    oracles = {"wire_polys": wire_polys, "prod_perm_poly": prod_perm_poly, "pub_inp_poly": pub_inp_poly,
               "split_quot_polys": split, "lin_poly": lin_poly}
    return proof, challenges, oracles


def batch_prove(cv: R.Curve, rng, circuits, pks, extra_msg: bytes | None = None):
    """PlonkKzgSnark::batch_prove for k TurboPlonk instances -> one BatchProof (snark.rs:61-74 -> 163-430): per round
    the instances run in order on the ONE rng and the ONE transcript; the quotient is sum_i alpha^(3i) t_i
    (prover.rs:519-656), the opening polynomials fold all instances (prover.rs:356-413)."""
    r = cv.fr.p
    if not circuits:
        raise ValueError("zero number of circuits/proving keys")
    if len(circuits) != len(pks):
        raise ValueError("the number of circuits %d != the number of proving keys %d" % (len(circuits), len(pks)))
    n, log_n, W = circuits[0].n, circuits[0].log_n, circuits[0].num_wire_types
    for cs, pk in zip(circuits, pks):
        if cs.n != n:
            raise ValueError("circuit domain size %d != expected domain size %d" % (cs.n, n))
        if pk["vk"]["domain_size"] != n:
            raise ValueError("proving key domain size %d != expected domain size %d" % (pk["vk"]["domain_size"], n))
    srs = pks[0]["srs"]
    m = 1
    while m < n * ((W * (n + 1) + 2) // n + 1):
        m <<= 1
    log_m = m.bit_length() - 1
    gen = cv.fr_generator
    g = R.domain_root(cv, log_n)
    tr = StandardTranscript(cv)
    if extra_msg is not None:
        tr.append_message(b"extra info", extra_msg)
    for cs, pk in zip(circuits, pks):
        tr.append_vk_and_pub_input(pk["vk"], cs.public_input())
    # Round 1
    wire_polys_v, pub_polys_v, wires_comms_v = [], [], []
    for cs, pk in zip(circuits, pks):
        wp = [_mask(cv, rng, R.ifft(cv, v, log_n), n, 1) for v in cs.wire_values()]
        comms = [_commit(cv, pk["srs"], p) for p in wp]
        wire_polys_v.append(wp)
        pub_polys_v.append(R.ifft(cv, cs.pub_input_values(), log_n))
        tr.append_commitments(b"witness_poly_comms", comms)
        wires_comms_v.append(comms)
    # Round 2
    beta = tr.get_challenge(b"beta")
    gamma = tr.get_challenge(b"gamma")
    perm_polys_v, perm_comms_v = [], []
    for cs, pk in zip(circuits, pks):
        z = _mask(cv, rng, R.ifft(cv, cs.prod_permutation_values(beta, gamma), log_n), n, 2)
        c = _commit(cv, pk["srs"], z)
        tr.append_commitments(b"perm_poly_comms", [c])
        perm_polys_v.append(z)
        perm_comms_v.append(c)
    # Round 3
    alpha = tr.get_challenge(b"alpha")
    alpha_3 = pow(alpha, 3, r)
    quot_evals = [0] * m
    alpha_base = 1
    for i, (cs, pk) in enumerate(zip(circuits, pks)):
        sel_c = [R.fft(cv, p, log_m, gen) for p in pk["selectors"]]
        sig_c = [R.fft(cv, p, log_m, gen) for p in pk["sigmas"]]
        wir_c = [R.fft(cv, p, log_m, gen) for p in wire_polys_v[i]]
        z_c = R.fft(cv, perm_polys_v[i], log_m, gen)
        pi_c = R.fft(cv, pub_polys_v[i], log_m, gen)
        ev = R.quotient_evals_turbo(cv, log_n, log_m, sel_c, sig_c, wir_c, z_c, pi_c, cs.k, alpha, beta, gamma, alpha_base)
        quot_evals = [(a + b) % r for a, b in zip(quot_evals, ev)]
        alpha_base = alpha_base * alpha_3 % r
    quot = R.ifft(cv, quot_evals, log_m, gen)
    while quot and quot[-1] == 0:
        quot.pop()
    expected_degree = W * (n + 1) + 2
    if len(quot) - 1 != expected_degree:
        raise ValueError("WrongQuotientPolyDegree(%d, %d)" % (len(quot) - 1, expected_degree))
    split = []
    for i in range(W):
        end = (i + 1) * (n + 2) if i < W - 1 else len(quot)
        split.append(quot[i * (n + 2):end])
    last = 0
    for i in range(W - 1):
        now = fr_rand(cv, rng)
        split[i][0] = (split[i][0] - last) % r
        split[i].append(now)
        last = now
    split[W - 1][0] = (split[W - 1][0] - last) % r
    split_quot_poly_comms = [_commit(cv, srs, p) for p in split]
    tr.append_commitments(b"quot_poly_comms", split_quot_poly_comms)
    # Round 4
    zeta = tr.get_challenge(b"zeta")
    evals_v = []
    for i, pk in enumerate(pks):
        ev = {"wires_evals": [R.poly_eval(cv, p, zeta) for p in wire_polys_v[i]],
              "wire_sigma_evals": [R.poly_eval(cv, p, zeta) for p in pk["sigmas"][:W - 1]],
              "perm_next_eval": R.poly_eval(cv, perm_polys_v[i], zeta * g % r)}
        tr.append_field_elems(b"wire_evals", ev["wires_evals"])
        tr.append_field_elems(b"wire_sigma_evals", ev["wire_sigma_evals"])
        tr.append_field_elems(b"perm_next_eval", [ev["perm_next_eval"]])
        evals_v.append(ev)
    vanish_eval = (pow(zeta, n, r) - 1) % r
    zeta_n2 = (vanish_eval + 1) * zeta % r * zeta % r
    lagrange_1 = vanish_eval * pow(n * (zeta - 1), -1, r) % r
    terms = []
    coeff = 1
    for i, t in enumerate(split):
        if i:
            coeff = coeff * zeta_n2 % r
        terms.append((-vanish_eval * coeff, t))
    alpha_base = 1
    for i, pk in enumerate(pks):
        w, sig_e, znext = evals_v[i]["wires_evals"], evals_v[i]["wire_sigma_evals"], evals_v[i]["perm_next_eval"]
        q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r), pow(w[2], 5, r),
                     pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
        terms += [(alpha_base * s, p) for s, p in zip(q_scalars, pk["selectors"])]
        c_z = alpha
        for j in range(W):
            c_z = c_z * (w[j] + beta * pk["vk"]["k"][j] % r * zeta + gamma) % r
        c_z = (c_z + alpha * alpha % r * lagrange_1) % r
        terms.append((alpha_base * c_z, perm_polys_v[i]))
        c_s = alpha * beta % r * znext % r
        for j in range(W - 1):
            c_s = c_s * (w[j] + beta * sig_e[j] + gamma) % r
        terms.append((-c_s * alpha_base, pk["sigmas"][W - 1]))
        alpha_base = alpha_base * alpha_3 % r
    lin_poly = _lincomb(r, terms, n + 3)
    # Round 5
    v = tr.get_challenge(b"v")
    polys = [lin_poly]
    for i, pk in enumerate(pks):
        polys += wire_polys_v[i] + pk["sigmas"][:W - 1]
    batch, c = [], 1
    for p in polys:
        batch.append((c, p))
        c = c * v % r
    q, _ = R.witness_poly(cv, _lincomb(r, batch, n + 3), zeta)
    opening_proof = _commit(cv, srs, q)
    batch, c = [], 1
    for z in perm_polys_v:
        batch.append((c, z))
        c = c * v % r
    q, _ = R.witness_poly(cv, _lincomb(r, batch, n + 3), zeta * g % r)
    shifted_opening_proof = _commit(cv, srs, q)
    bp = {"wires_poly_comms_vec": wires_comms_v, "prod_perm_poly_comms_vec": perm_comms_v, "poly_evals_vec": evals_v,
          "split_quot_poly_comms": split_quot_poly_comms, "opening_proof": opening_proof,
          "shifted_opening_proof": shifted_opening_proof}
    challenges = {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v}
    return bp, challenges


# ------------------------------------------------------------------------------------------------
# verifier
# ------------------------------------------------------------------------------------------------


def verify(cv: R.Curve, vk: dict, pub_input, proof: dict, extra_msg: bytes | None = None, trapdoor_beta: int | None = None,
           pairing_check=None) -> bool:
    """PlonkKzgSnark::verify for one TurboPlonk proof (verifier.rs).  The final check e(A, [beta]h) == e(B, h) is
    evaluated by `pairing_check(A, beta_h, B, h)` if given, else as [trapdoor_beta]A == B."""
    r = cv.fr.p
    n = vk["domain_size"]
    log_n = n.bit_length() - 1
    W = len(proof["wires_poly_comms"])
    if len(pub_input) != vk["num_inputs"]:
        return False
    for P in (proof["wires_poly_comms"] + proof["split_quot_poly_comms"]
              + [proof["prod_perm_poly_comm"], proof["opening_proof"], proof["shifted_opening_proof"]]):
        if not R.on_curve(cv, P):
            return False
    # compute_challenges (:262-339)
    tr = StandardTranscript(cv)
    if extra_msg is not None:
        tr.append_message(b"extra info", extra_msg)
    tr.append_vk_and_pub_input(vk, pub_input)
    tr.append_commitments(b"witness_poly_comms", proof["wires_poly_comms"])
    beta = tr.get_challenge(b"beta")
    gamma = tr.get_challenge(b"gamma")
    tr.append_commitments(b"perm_poly_comms", [proof["prod_perm_poly_comm"]])
    alpha = tr.get_challenge(b"alpha")
    tr.append_commitments(b"quot_poly_comms", proof["split_quot_poly_comms"])
    zeta = tr.get_challenge(b"zeta")
    tr.append_field_elems(b"wire_evals", proof["wires_evals"])
    tr.append_field_elems(b"wire_sigma_evals", proof["wire_sigma_evals"])
    tr.append_field_elems(b"perm_next_eval", [proof["perm_next_eval"]])
    v = tr.get_challenge(b"v")
    tr.append_commitments(b"open_proof", [proof["opening_proof"]])
    tr.append_commitments(b"shifted_open_proof", [proof["shifted_opening_proof"]])
    u = tr.get_challenge(b"u")

    g = R.domain_root(cv, log_n)
    alpha_2 = alpha * alpha % r
    vanish_eval = (pow(zeta, n, r) - 1) % r
    lagrange_1 = vanish_eval * pow(n * (zeta - 1), -1, r) % r
    w = proof["wires_evals"]
    sig = proof["wire_sigma_evals"]
    znext = proof["perm_next_eval"]
    # evaluate_pi_poly (:883-912): sum_i L_i(zeta) pub_input[i], L_i(z) = Z_H(z) g^i / (n (z - g^i))
    pi_eval = 0
    gi = 1
    for x in pub_input:
        pi_eval = (pi_eval + vanish_eval * gi % r * pow(n * (zeta - gi), -1, r) % r * x) % r
        gi = gi * g % r
    # compute_lin_poly_constant_term (:358-432)
    tmp = (pi_eval - alpha_2 * lagrange_1) % r
    acc = alpha * znext % r * (gamma + w[W - 1]) % r
    for we, se in zip(w[:W - 1], sig):
        acc = acc * (gamma + we + beta * se) % r
    lin_poly_constant = (tmp - acc) % r
    # linearization_scalars_and_bases (:541-713)
    sb = []
    coeff = alpha_2 * lagrange_1 % r
    acc = alpha
    for we, k in zip(w, vk["k"]):
        acc = acc * (beta * k % r * zeta + gamma + we) % r
    sb.append(((coeff + acc) % r, proof["prod_perm_poly_comm"]))
    acc = alpha * beta % r * znext % r
    for we, se in zip(w[:W - 1], sig):
        acc = acc * (beta * se + gamma + we) % r
    sb.append((-acc % r, vk["sigma_comms"][-1]))
    q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r), pow(w[2], 5, r),
                 pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
    sb += [(s % r, c) for s, c in zip(q_scalars, vk["selector_comms"])]
    zeta_n2 = (1 + vanish_eval) * zeta % r * zeta % r
    coeff = -vanish_eval % r
    sb.append((coeff, proof["split_quot_poly_comms"][0]))
    for c in proof["split_quot_poly_comms"][1:]:
        coeff = coeff * zeta_n2 % r
        sb.append((coeff, c))
    # aggregate_poly_commitments (:439-535) / aggregate_evaluations (:718-788)
    v_base, uv_base = v, u
    result = -lin_poly_constant % r
    for c, e in zip(proof["wires_poly_comms"] + vk["sigma_comms"][:W - 1], w + sig):
        sb.append((v_base, c))
        result = (result + e * v_base) % r
        v_base = v_base * v % r
    sb.append((uv_base, proof["prod_perm_poly_comm"]))
    result = (result + znext * uv_base) % r
    # batch_verify_opening_proofs (:201-257), single instance (r = 1)
    A = R.msm_naive(cv, [proof["opening_proof"], proof["shifted_opening_proof"]], [1, u])
    next_eval_point = zeta * g % r
    sb.append((zeta, proof["opening_proof"]))
    sb.append((u * next_eval_point % r, proof["shifted_opening_proof"]))
    sb.append((-result % r, vk["g"]))
    B = R.msm_naive(cv, [c for _, c in sb], [s for s, _ in sb])
    if pairing_check is not None:
        return pairing_check(A, vk["beta_h"], B, vk["h"])
    assert trapdoor_beta is not None
    return R.ec_mul(cv, trapdoor_beta, A) == B


def verify_batch_proof(cv: R.Curve, vks, pub_inputs, bp: dict, extra_msg: bytes | None = None,
                       trapdoor_beta: int | None = None, pairing_check=None) -> bool:
    """Verifier::prepare_pcs_info + batch_verify_opening_proofs for ONE BatchProof over k TurboPlonk instances
    (verifier.rs:70-257, 262-339, 358-432, 439-535, 541-713, 718-788).  Final pairing check as in verify()."""
    r = cv.fr.p
    k_inst = len(bp["prod_perm_poly_comms_vec"])
    if not (len(vks) == k_inst == len(pub_inputs)) or k_inst == 0:
        return False
    n = vks[0]["domain_size"]
    log_n = n.bit_length() - 1
    for vk, pi in zip(vks, pub_inputs):
        if len(pi) != vk["num_inputs"] or vk["domain_size"] != n:
            return False
    pts = bp["split_quot_poly_comms"] + bp["prod_perm_poly_comms_vec"] + [bp["opening_proof"], bp["shifted_opening_proof"]]
    for comms in bp["wires_poly_comms_vec"]:
        pts = pts + comms
    if not all(R.on_curve(cv, P) for P in pts):
        return False
    tr = StandardTranscript(cv)
    if extra_msg is not None:
        tr.append_message(b"extra info", extra_msg)
    for vk, pi in zip(vks, pub_inputs):
        tr.append_vk_and_pub_input(vk, pi)
    for comms in bp["wires_poly_comms_vec"]:
        tr.append_commitments(b"witness_poly_comms", comms)
    beta = tr.get_challenge(b"beta")
    gamma = tr.get_challenge(b"gamma")
    tr.append_commitments(b"perm_poly_comms", bp["prod_perm_poly_comms_vec"])
    alpha = tr.get_challenge(b"alpha")
    tr.append_commitments(b"quot_poly_comms", bp["split_quot_poly_comms"])
    zeta = tr.get_challenge(b"zeta")
    for ev in bp["poly_evals_vec"]:
        tr.append_field_elems(b"wire_evals", ev["wires_evals"])
        tr.append_field_elems(b"wire_sigma_evals", ev["wire_sigma_evals"])
        tr.append_field_elems(b"perm_next_eval", [ev["perm_next_eval"]])
    v = tr.get_challenge(b"v")
    tr.append_commitments(b"open_proof", [bp["opening_proof"]])
    tr.append_commitments(b"shifted_open_proof", [bp["shifted_opening_proof"]])
    u = tr.get_challenge(b"u")

    g = R.domain_root(cv, log_n)
    alpha_2 = alpha * alpha % r
    alpha_3 = alpha_2 * alpha % r
    alpha_bases = [pow(alpha_3, i, r) for i in range(k_inst)]
    vanish_eval = (pow(zeta, n, r) - 1) % r
    lagrange_1 = vanish_eval * pow(n * (zeta - 1), -1, r) % r
    lin_poly_constant = 0
    sb = []
    for i, (vk, pi) in enumerate(zip(vks, pub_inputs)):
        ev = bp["poly_evals_vec"][i]
        w, sig, znext = ev["wires_evals"], ev["wire_sigma_evals"], ev["perm_next_eval"]
        W = len(w)
        ab = alpha_bases[i]
        pi_eval = 0
        gi = 1
        for x in pi:
            pi_eval = (pi_eval + vanish_eval * gi % r * pow(n * (zeta - gi), -1, r) % r * x) % r
            gi = gi * g % r
        tmp = (pi_eval - alpha_2 * lagrange_1) % r
        acc = alpha * znext % r * (gamma + w[W - 1]) % r
        for we, se in zip(w[:W - 1], sig):
            acc = acc * (gamma + we + beta * se) % r
        lin_poly_constant = (lin_poly_constant + ab * (tmp - acc)) % r
        coeff = alpha_2 * lagrange_1 % r
        acc = alpha
        for we, kk in zip(w, vk["k"]):
            acc = acc * (beta * kk % r * zeta + gamma + we) % r
        sb.append(((coeff + acc) * ab % r, bp["prod_perm_poly_comms_vec"][i]))
        acc = alpha * beta % r * znext % r
        for we, se in zip(w[:W - 1], sig):
            acc = acc * (beta * se + gamma + we) % r
        sb.append((-acc * ab % r, vk["sigma_comms"][-1]))
        q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r),
                     pow(w[2], 5, r), pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
        sb += [(s * ab % r, c) for s, c in zip(q_scalars, vk["selector_comms"])]
    zeta_n2 = (1 + vanish_eval) * zeta % r * zeta % r
    coeff = -vanish_eval % r
    sb.append((coeff, bp["split_quot_poly_comms"][0]))
    for c in bp["split_quot_poly_comms"][1:]:
        coeff = coeff * zeta_n2 % r
        sb.append((coeff, c))
    v_base, uv_base = v, u
    result = -lin_poly_constant % r
    for i, vk in enumerate(vks):
        ev = bp["poly_evals_vec"][i]
        W = len(bp["wires_poly_comms_vec"][i])
        for c, e in zip(bp["wires_poly_comms_vec"][i] + vk["sigma_comms"][:W - 1], ev["wires_evals"] + ev["wire_sigma_evals"]):
            sb.append((v_base, c))
            result = (result + e * v_base) % r
            v_base = v_base * v % r
        sb.append((uv_base, bp["prod_perm_poly_comms_vec"][i]))
        result = (result + ev["perm_next_eval"] * uv_base) % r
        uv_base = uv_base * v % r
    A = R.msm_naive(cv, [bp["opening_proof"], bp["shifted_opening_proof"]], [1, u])
    sb.append((zeta, bp["opening_proof"]))
    sb.append((u * (zeta * g % r) % r, bp["shifted_opening_proof"]))
    sb.append((-result % r, vks[0]["g"]))
    B = R.msm_naive(cv, [c for _, c in sb], [s for s, _ in sb])
    if pairing_check is not None:
        return pairing_check(A, vks[0]["beta_h"], B, vks[0]["h"])
    assert trapdoor_beta is not None
    return R.ec_mul(cv, trapdoor_beta, A) == B
