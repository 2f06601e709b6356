"""Generate tests/golden/*.json from the big-integer oracle (oracle/pyref.py).

TEST INFRASTRUCTURE ONLY.  Run in the authoring container:  python -m oracle.gen_golden
The reference holds no literal vectors for this path, so these are
computed from first principles with Python integers: definition-level MSM
(double-and-add, no windows), recursive radix-2 FFT cross-checked by Horner evaluation,
schoolbook synthetic division.  MSM / NTT / division results are mathematically unique,
so they are valid expectations for any correct implementation, arkworks included.
Public known answers (curve generators, group orders, 2*G of alt_bn128 from EIP-196,
roots of unity) pin the constants themselves.
"""
from __future__ import annotations

import hashlib
import json
import os
import struct

from . import pyref as R

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def hx(x: int) -> str:
    return hex(x)


def pt(P):
    return None if P is None else [hx(P[0]), hx(P[1])]


def mont_bytes(f: R.Field, xs) -> bytes:
    out = bytearray()
    for x in xs:
        for l in f.mont_limbs(x):
            out += struct.pack("<Q", l)
    return bytes(out)


def scalar_sets(cv: R.Curve, n: int, rng: R.SplitMix64):
    r = cv.fr.p
    sets = {"random": [rng.field(cv.fr) for _ in range(n)]}
    if n <= 64:
        sets["zeros"] = [0] * n
        sets["ones"] = [1] * n
        sets["r_minus_1"] = [r - 1] * n
        single = [0] * n
        single[n // 2] = rng.field(cv.fr)
        sets["single"] = single
        lz = [rng.field(cv.fr) for _ in range(n)]
        for i in range(min(n - 1, max(1, n // 3))):
            lz[i] = 0
        sets["leading_zeros"] = lz
        # window-boundary stress: all-ones bit patterns and powers of two
        sets["patterns"] = [((1 << (1 + (37 * i) % 253)) - 1) % r if i % 2 else (1 << ((53 * i) % 254)) % r
                            for i in range(n)]
    return sets


def gen_constants():
    out = {}
    for cv in R.CURVES.values():
        G2 = R.ec_add(cv, cv.g, cv.g)
        out[cv.name] = {
            "curve_id": cv.curve_id,
            "fq_p": hx(cv.fq.p), "fr_p": hx(cv.fr.p), "b": cv.b,
            "g": pt(cv.g), "g2": pt(G2),
            "fr_generator": cv.fr_generator, "two_adicity": cv.two_adicity,
            "two_adic_root": hx(cv.two_adic_root),
            "fq_R": hx(cv.fq.R % cv.fq.p), "fq_R2": hx(cv.fq.R ** 2 % cv.fq.p), "fq_inv32": hx(cv.fq.inv32()),
            "fr_R": hx(cv.fr.R % cv.fr.p), "fr_R2": hx(cv.fr.R ** 2 % cv.fr.p), "fr_inv32": hx(cv.fr.inv32()),
        }
        assert R.on_curve(cv, cv.g) and R.ec_mul(cv, cv.fr.p, cv.g) is None
        assert pow(cv.two_adic_root, 1 << cv.two_adicity, cv.fr.p) == 1
        assert pow(cv.two_adic_root, 1 << (cv.two_adicity - 1), cv.fr.p) == cv.fr.p - 1
        assert pow(cv.fr_generator, (cv.fr.p - 1) >> cv.two_adicity, cv.fr.p) == cv.two_adic_root
    # public known answer: alt_bn128 (1,2)+(1,2), EIP-196 / go-ethereum bn256Add vector
    assert out["bn254"]["g2"] == [
        "0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3",
        "0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4",
    ]
    return out


def gen_field():
    out = {}
    for cv in R.CURVES.values():
        for f in (cv.fr, cv.fq):
            rng = R.SplitMix64(0xF1E1D + len(out))
            p = f.p
            a = [0, 1, p - 1, f.R % p, 2, p - 2] + [rng.field(f) for _ in range(10)]
            b = [p - 1, p - 1, p - 1, f.R % p, (p + 1) // 2, 3] + [rng.field(f) for _ in range(10)]
            out[f.name] = {
                "a": [hx(x) for x in a], "b": [hx(x) for x in b],
                "add": [hx((x + y) % p) for x, y in zip(a, b)],
                "sub": [hx((x - y) % p) for x, y in zip(a, b)],
                "mul": [hx(x * y % p) for x, y in zip(a, b)],
                "neg": [hx(-x % p) for x in a],
                "inv": [hx(pow(x, -1, p) if x else 0) for x in a],
            }
    return out


def gen_msm():
    out = {}
    for cv in R.CURVES.values():
        rng = R.SplitMix64(0x6A660001 + cv.curve_id)
        beta = rng.field(cv.fr)
        gk = rng.field(cv.fr)
        g = R.ec_mul(cv, gk, cv.g)
        cases = []
        nmax = 256
        srs = R.srs_powers(cv, beta, g, nmax + 8)
        for n in (1, 2, 3, 31, 32, 33, 64, 256):
            for name, sc in scalar_sets(cv, n, rng).items():
                exp = R.msm_naive(cv, srs[:n], sc)
                assert exp == R.commit_via_beta(cv, beta, g, sc)
                cases.append({"n": n, "set": name, "offset": 0, "scalars": [hx(s) for s in sc], "expect": pt(exp)})
        # SRS sub-range (commit skips low-order zero coefficients: mod.rs:108-116,1044-1053)
        sc = [rng.field(cv.fr) for _ in range(40)]
        exp = R.msm_naive(cv, srs[5:45], sc)
        cases.append({"n": 40, "set": "offset5", "offset": 5, "scalars": [hx(s) for s in sc], "expect": pt(exp)})
        # larger sizes: closed form [p(beta)] g only (the definition-level loop is too slow in Python)
        for n in (1 << 10, (1 << 12) + 3):
            sc_seed = 0xABC000 + n
            r2 = R.SplitMix64(sc_seed)
            sc = [r2.field(cv.fr) for _ in range(n)]
            cases.append({"n": n, "set": "seeded", "offset": 0, "scalar_seed": sc_seed,
                          "expect": pt(R.commit_via_beta(cv, beta, g, sc))})
        # arbitrary (non-SRS) bases with repeats, negations and identities
        pts = [R.ec_mul(cv, rng.field(cv.fr), cv.g) for _ in range(6)]
        bases = [pts[0], pts[0], R.ec_neg(cv, pts[0]), None, pts[1], pts[2], pts[1], pts[3], None, pts[4], pts[5],
                 R.ec_neg(cv, pts[5])]
        bsets = []
        for sname, sc in (
            ("cancel", [5, 7, 12, 9, 1, 2, cv.fr.p - 1, 3, 4, 0, 6, 6]),
            ("double", [3, 3, 0, 1, 8, 0, 8, 0, 0, 0, 1, 0]),
            ("random", [rng.field(cv.fr) for _ in range(12)]),
            ("same_digit", [0x10001] * 12),
        ):
            bsets.append({"set": sname, "scalars": [hx(s) for s in sc], "expect": pt(R.msm_naive(cv, bases, sc))})
        out[cv.name] = {
            "beta": hx(beta), "g": pt(g),
            "srs_spot": {str(i): pt(srs[i]) for i in (0, 1, 2, 17, 255, 263)},
            "cases": cases,
            "arbitrary_bases": [pt(P) for P in bases],
            "arbitrary_cases": bsets,
        }
    return out


def gen_ntt():
    out = {}
    for cv in R.CURVES.values():
        cases = []
        for log_n in (1, 2, 3, 4, 5, 8, 10, 12):
            n = 1 << log_n
            for offset in (1, cv.fr_generator):
                for in_len in sorted({n, max(1, n // 8 + 3 if n >= 8 else n)}):
                    if in_len > n:
                        continue
                    seed = 0x77000 + 64 * log_n + offset + (0 if in_len == n else 1000)
                    rng = R.SplitMix64(seed)
                    coeffs = [rng.field(cv.fr) for _ in range(in_len)]
                    ev = R.fft(cv, coeffs, log_n, offset)
                    spots = [0, 1, n // 2, n - 1] if n >= 4 else list(range(n))
                    assert [ev[i] for i in spots] == R.fft_direct(cv, coeffs, log_n, offset, spots)
                    assert R.ifft(cv, ev, log_n, offset)[:in_len] == coeffs
                    case = {"log_n": log_n, "offset": offset, "in_len": in_len, "seed": seed,
                            "fwd_sha256": hashlib.sha256(mont_bytes(cv.fr, ev)).hexdigest(),
                            "spots": {str(i): hx(ev[i]) for i in spots}}
                    if n <= 16:
                        case["coeffs"] = [hx(c) for c in coeffs]
                        case["evals"] = [hx(e) for e in ev]
                    # the inverse direction on independent data: ifft(random evals)
                    rng2 = R.SplitMix64(seed ^ 0x5555)
                    evals_in = [rng2.field(cv.fr) for _ in range(n)]
                    co = R.ifft(cv, evals_in, log_n, offset)
                    case["inv_sha256"] = hashlib.sha256(mont_bytes(cv.fr, co)).hexdigest()
                    case["inv_spots"] = {str(i): hx(co[i]) for i in spots}
                    cases.append(case)
        out[cv.name] = {"cases": cases, "roots": {str(k): hx(R.domain_root(cv, k)) for k in (1, 2, 3, 10, 20, 25)}}
    return out


def gen_open():
    out = {}
    for cv in R.CURVES.values():
        rng = R.SplitMix64(0x09E20 + cv.curve_id)
        beta = rng.field(cv.fr)
        cases = []
        for n in (1, 2, 5, 33, 300):
            coeffs = [rng.field(cv.fr) for _ in range(n)]
            z = rng.field(cv.fr)
            q, ev = R.witness_poly(cv, coeffs, z)
            assert ev == R.poly_eval(cv, coeffs, z)
            # (X - z) q(X) + ev == f(X) at a random point
            t = rng.field(cv.fr)
            assert ((t - z) * R.poly_eval(cv, q, t) + ev - R.poly_eval(cv, coeffs, t)) % cv.fr.p == 0
            cases.append({"n": n, "coeffs": [hx(c) for c in coeffs], "z": hx(z), "eval": hx(ev),
                          "witness": [hx(c) for c in q],
                          "proof": pt(R.commit_via_beta(cv, beta, cv.g, q)),
                          "commitment": pt(R.commit_via_beta(cv, beta, cv.g, coeffs))})
        out[cv.name] = {"beta": hx(beta), "g": pt(cv.g), "cases": cases}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("constants", gen_constants), ("field", gen_field), ("msm", gen_msm), ("ntt", gen_ntt),
                     ("open", gen_open)):
        data = fn()
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(data, f, indent=0, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(OUT, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
