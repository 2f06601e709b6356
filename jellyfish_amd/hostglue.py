"""Host-side glue of the prover that the reference also runs on the CPU: field/curve constants, the RNG the
reference draws blinding factors from, ark-serialize encodings for the transcript and the proof, and the one G2
scalar multiplication of the test SRS.  Nothing here is on the hot path (a few dozen elements per proof).

Reference: relation/src/constants.rs:30-84 (coset representatives), utilities/src/lib.rs:88-96 (test_rng),
plonk/src/transcript/mod.rs:46-186 (what is serialised into the transcript), plonk/src/proof_system/structs.rs:54-76
(Proof), snark.rs:447-488 (universal_setup_for_testing).  rand_chacha / ark-ff sampling / ark-serialize are external
crates, restated from their published algorithms.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class CurveParams:
    name: str
    curve_id: int
    q: int          # base field modulus
    r: int          # scalar field modulus
    fq_limbs64: int
    b: int
    g1: tuple
    g2: tuple
    fr_generator: int
    two_adicity: int
    two_adic_root: int

    @property
    def fr_R(self) -> int:
        return 1 << 256

    @property
    def fq_R(self) -> int:
        return 1 << (64 * self.fq_limbs64)


BN254 = CurveParams(
    "bn254", 0,
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    21888242871839275222246405745257275088548364400416034343698204186575808495617, 4, 3, (1, 2),
    ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    5, 28, 19103219067921713944291392827692070036145651957329286315305642004821462161904)

BLS12_381 = CurveParams(
    "bls12_381", 1,
    0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 6, 4,
    (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
     0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
    ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
      0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
     (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
      0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)),
    7, 32, 10238227357739495823651030575849232062558860180284477541189508159991286009131)

CURVES = {0: BN254, 1: BLS12_381, "bn254": BN254, "bls12_381": BLS12_381}

_M64 = (1 << 64) - 1

# ---- limb conversions (the boundary's format: Montgomery, little-endian u64 limbs) ----------------


def fr_to_mont(cp: CurveParams, xs) -> np.ndarray:
    """canonical ints -> (len, 4) uint64 Montgomery limbs"""
    R, r = cp.fr_R, cp.r
    buf = b"".join(((x % r) * R % r).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def fr_from_mont(cp: CurveParams, a) -> list[int]:
    """(len, 4) uint64 Montgomery limbs -> canonical ints"""
    raw = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).tobytes()
    rinv = pow(cp.fr_R, -1, cp.r)
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") * rinv % cp.r for i in range(len(raw) // 32)]


def g1_from_record(cp: CurveParams, rec: bytes):
    """packed affine x||y (Montgomery) followed by a u32 infinity flag -> canonical (x, y) or None"""
    nb = 8 * cp.fq_limbs64
    if int.from_bytes(rec[2 * nb:2 * nb + 4], "little"):
        return None
    rinv = pow(cp.fq_R, -1, cp.q)
    return (int.from_bytes(rec[:nb], "little") * rinv % cp.q, int.from_bytes(rec[nb:2 * nb], "little") * rinv % cp.q)


def g1_to_record(cp: CurveParams, P) -> bytes:
    """inverse of g1_from_record: canonical (x, y) or None -> packed Montgomery x||y + u32 infinity flag"""
    nb = 8 * cp.fq_limbs64
    if P is None:
        return bytes(2 * nb) + (1).to_bytes(4, "little")
    return ((P[0] * cp.fq_R % cp.q).to_bytes(nb, "little") + (P[1] * cp.fq_R % cp.q).to_bytes(nb, "little")
            + bytes(4))


def g1_to_mont(cp: CurveParams, P) -> np.ndarray:
    nb = 8 * cp.fq_limbs64
    buf = (P[0] * cp.fq_R % cp.q).to_bytes(nb, "little") + (P[1] * cp.fq_R % cp.q).to_bytes(nb, "little")
    return np.frombuffer(buf, dtype=np.uint64).copy()


# ---- RNG -----------------------------------------------------------------------------------------


class ChaChaRng:
    """rand_chacha ChaCha{12,20}Rng word stream (key = seed, 64-bit block counter, stream 0); next_u64 joins two
    consecutive words, low word first."""

    def __init__(self, seed: bytes, rounds: int = 20):
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        self._key = np.frombuffer(seed, dtype="<u4").astype(np.uint64)
        self._rounds = rounds
        self._counter = 0
        self._words = np.zeros(0, dtype=np.uint32)
        self._i = 0

    def _refill(self):
        # 64 blocks at once, column-vectorised over the blocks: the cost of a refill is the ~600 numpy calls, not the
        # width, and prove() sits on this stream (17+ field draws per proof; 4 blocks per refill cost 0.4 ms per draw)
        nblk = 64
        ctr = np.arange(self._counter, self._counter + nblk, dtype=np.uint64)
        self._counter += nblk
        M = np.uint64(0xFFFFFFFF)
        init = np.zeros((16, nblk), dtype=np.uint64)
        init[0], init[1], init[2], init[3] = 0x61707865, 0x3320646E, 0x79622D32, 0x6B206574
        for i in range(8):
            init[4 + i] = self._key[i]
        init[12] = ctr & M
        init[13] = (ctr >> np.uint64(32)) & M
        x = init.copy()

        def rotl(v, n):
            return ((v << np.uint64(n)) | (v >> np.uint64(32 - n))) & M

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & M
            x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & M
            x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & M
            x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & M
            x[b] = rotl(x[b] ^ x[c], 7)

        for _ in range(self._rounds // 2):
            qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15)
            qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14)
        out = (x + init) & M
        self._words = out.T.reshape(-1).astype(np.uint32)
        self._i = 0

    def next_u32(self) -> int:
        if self._i >= self._words.size:
            self._refill()
        v = int(self._words[self._i])
        self._i += 1
        return v

    def next_u64(self) -> int:
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


def test_rng() -> ChaChaRng:
    """jf_utils::test_rng (utilities/src/lib.rs:88-96): StdRng = ChaCha12Rng with the reference's fixed seed"""
    return ChaChaRng(bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16), rounds=12)


_FR_RINV: dict = {}


def _fr_rinv(cp: CurveParams) -> int:
    if cp.r not in _FR_RINV:
        _FR_RINV[cp.r] = pow(cp.fr_R, -1, cp.r)
    return _FR_RINV[cp.r]


def fr_rand(cp: CurveParams, rng) -> int:
    """ark-ff Fp::rand: four u64 limbs, unused top bits cleared, rejected if >= r; the draw is the Montgomery
    representation.  Returns the canonical value."""
    top_bits = cp.r.bit_length() - 192
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << top_bits) - 1
        v = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        if v < cp.r:
            return v * _fr_rinv(cp) % cp.r


def compute_coset_representatives(cp: CurveParams, num_wire_types: int, coset_size: int) -> list[int]:
    """relation/src/constants.rs:30-84: k_0 = 1, then ChaCha20(seed 0) draws kept when they open a new coset of H"""
    rng = ChaChaRng(bytes(32), rounds=20)
    ks, pows = [1], [1]
    while len(ks) < num_wire_types:
        cand = fr_rand(cp, rng)
        pw = pow(cand, coset_size, cp.r)
        if all(pow(q, -1, cp.r) * pw % cp.r != 1 for q in pows):
            ks.append(cand)
            pows.append(pw)
    return ks[:num_wire_types]


# ---- ark-serialize (compressed) ----------------------------------------------------------------------


def ser_fr(cp: CurveParams, x: int) -> bytes:
    return (x % cp.r).to_bytes(32, "little")


def ser_g1(cp: CurveParams, P) -> bytes:
    nb = 8 * cp.fq_limbs64
    little = cp.curve_id == 0  # BN254: ark default (LE, flags in the last byte); BLS12-381: zcash format (BE, first byte)
    if P is None:
        b = bytearray(nb)
        b[-1 if little else 0] |= 0x40 if little else 0xC0
        return bytes(b)
    y_neg = P[1] > (cp.q - P[1]) % cp.q
    b = bytearray(P[0].to_bytes(nb, "little" if little else "big"))
    if little:
        b[-1] |= 0x80 if y_neg else 0
    else:
        b[0] |= 0x80 | (0x20 if y_neg else 0)
    return bytes(b)


def ser_g2(cp: CurveParams, P) -> bytes:
    nb = 8 * cp.fq_limbs64
    (x0, x1), (y0, y1) = P
    n0, n1 = (-y0) % cp.q, (-y1) % cp.q
    y_neg = (y1, y0) > (n1, n0)  # ark's Fp2 ordering: c1 first
    if cp.curve_id == 0:
        b = bytearray(x0.to_bytes(nb, "little") + x1.to_bytes(nb, "little"))
        b[-1] |= 0x80 if y_neg else 0
    else:
        b = bytearray(x1.to_bytes(nb, "big") + x0.to_bytes(nb, "big"))
        b[0] |= 0x80 | (0x20 if y_neg else 0)
    return bytes(b)


# ---- G2 (test SRS only: [beta]h) ---------------------------------------------------------------------


def _f2mul(q, a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _f2inv(q, a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, q)
    return (a[0] * d % q, -a[1] * d % q)


def _g2_add(q, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if ((y1[0] + y2[0]) % q, (y1[1] + y2[1]) % q) == (0, 0):
            return None
        lam = _f2mul(q, _f2mul(q, (3, 0), _f2mul(q, x1, x1)), _f2inv(q, (2 * y1[0] % q, 2 * y1[1] % q)))
    else:
        lam = _f2mul(q, ((y2[0] - y1[0]) % q, (y2[1] - y1[1]) % q), _f2inv(q, ((x2[0] - x1[0]) % q, (x2[1] - x1[1]) % q)))
    l2 = _f2mul(q, lam, lam)
    x3 = ((l2[0] - x1[0] - x2[0]) % q, (l2[1] - x1[1] - x2[1]) % q)
    t = _f2mul(q, lam, ((x1[0] - x3[0]) % q, (x1[1] - x3[1]) % q))
    return (x3, ((t[0] - y1[0]) % q, (t[1] - y1[1]) % q))


def g2_mul(cp: CurveParams, k: int, P):
    k %= cp.r
    acc = None
    while k:
        if k & 1:
            acc = _g2_add(cp.q, acc, P)
        P = _g2_add(cp.q, P, P)
        k >>= 1
    return acc
