"""Shared helpers for the tests: conversions between Python integers and the boundary's
Montgomery limb arrays, deterministic inputs, golden fixture access."""
from __future__ import annotations

import json
import os

import numpy as np

from oracle import pyref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELD_ID = {"bn254_fr": 0, "bn254_fq": 1, "bls12_381_fr": 2, "bls12_381_fq": 3}


def golden(name: str):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def h2i(s) -> int:
    return int(s, 16)


def mont_arr(f: R.Field, xs) -> np.ndarray:
    """integers -> (len, limbs64) uint64 Montgomery limbs"""
    return np.array([f.mont_limbs(x) for x in xs], dtype=np.uint64).reshape(len(xs), f.limbs64)


def from_mont_arr(f: R.Field, a) -> list[int]:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, f.limbs64)
    return [f.from_mont(R.Field.from_limbs(row)) for row in a]


def pt_arr(cv: R.Curve, P) -> np.ndarray:
    n = cv.fq.limbs64
    if P is None:
        return np.zeros(2 * n, dtype=np.uint64)
    return np.array(cv.fq.mont_limbs(P[0]) + cv.fq.mont_limbs(P[1]), dtype=np.uint64)


def pt_from(cv: R.Curve, a):
    n = cv.fq.limbs64
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    x = R.Field.from_limbs(a[:n])
    y = R.Field.from_limbs(a[n:2 * n])
    if x == 0 and y == 0:
        return None
    return (cv.fq.from_mont(x), cv.fq.from_mont(y))


def gpt(p):
    """golden point ([hexx, hexy] or None) -> tuple"""
    return None if p is None else (h2i(p[0]), h2i(p[1]))


def seeded_scalars(cv: R.Curve, seed: int, n: int) -> list[int]:
    rng = R.SplitMix64(seed)
    return [rng.field(cv.fr) for _ in range(n)]


def rand_fr_mont(cv: R.Curve, n: int, seed: int) -> np.ndarray:
    """n uniformly random Fr elements as Montgomery limbs (any canonical value < r is a valid
    Montgomery representative, so uniform limbs below r are uniform field elements)."""
    rng = np.random.default_rng(seed)
    bits = cv.fr.p.bit_length()
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << (bits - 192)) - 1)
    p_limbs = np.array(cv.fr.limbs(cv.fr.p), dtype=np.uint64)
    while True:
        # lexicographic compare (top limb first) against p
        ge = np.zeros(n, dtype=bool)
        decided = np.zeros(n, dtype=bool)
        for l in (3, 2, 1, 0):
            gt = (out[:, l] > p_limbs[l]) & ~decided
            lt = (out[:, l] < p_limbs[l]) & ~decided
            ge |= gt
            decided |= gt | lt
        ge |= ~decided
        bad = np.nonzero(ge)[0]
        if bad.size == 0:
            return out
        fresh = rng.integers(0, 1 << 64, size=(bad.size, 4), dtype=np.uint64)
        fresh[:, 3] &= np.uint64((1 << (bits - 192)) - 1)
        out[bad] = fresh


def ref_kat_cases():
    """The reference-held iFFT known answers (tests/golden/ntt_ref_kat.json, transcribed from
    pcs/src/multilinear_kzg/util.rs:428-563): -> list of (cite, log_n, evals, coeffs) as integers mod r."""
    g = golden("ntt_ref_kat")
    r = int(g["fr_modulus"])
    assert r == R.BLS12_381.fr.p

    def val(x):
        if isinstance(x, list):
            return int(x[0]) * pow(int(x[1]), -1, r) % r
        return int(x) % r

    return [(c["cite"], c["log_n"], [val(x) for x in c["evals"]], [val(x) for x in c["coeffs"]]) for c in g["cases"]], g


def build_ultra_circuit(cs, size: int = 4, extra: int = 0):
    """An UltraPlonk test circuit for both builders (oracle and product share the method names): arithmetic gates,
    public inputs created out of order, range-checked variables and two key-value tables, one with fewer lookups than
    rows and one with more (relation/src/gadgets/ultraplonk/lookup_table.rs:91-131 test shapes)."""
    a = cs.create_variable(3)
    b = cs.create_public_variable(5 + extra)
    c = cs.mul(a, b)
    d = cs.add(c, a)
    for v in (a, b):
        cs.add_range_check_variable(v)
    cs.add_range_check_variable(cs.create_variable(6))
    tv = [(cs.create_variable(10 + i), cs.create_variable(100 + 7 * i)) for i in range(4)]
    lk = [(cs.create_variable(key), cs.create_variable(10 + key), cs.create_variable(100 + 7 * key)) for key in (2, 0, 2)]
    cs.create_table_and_lookup_variables(lk, tv)
    tv2 = [(cs.create_variable(55), cs.create_variable(66))]
    lk2 = [(cs.create_variable(0), cs.create_variable(55), cs.create_variable(66)) for _ in range(3)]
    cs.create_table_and_lookup_variables(lk2, tv2)
    for i in range(size + extra):
        d = cs.add(d, 1) if i % 3 else cs.mul(d, a)
    cs.create_public_variable(cs.witness[d])
    cs.finalize_for_arithmetization()
    return cs
