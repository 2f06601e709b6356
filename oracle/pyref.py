"""Big-integer CPU oracle for the jellyfish PLONK proving hot path (MSM + NTT).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
(``jellyfish_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Parity status: the reference (EspressoSystems/jellyfish) delegates the arithmetic on
this path to arkworks 0.5 (``ark-ec`` ``VariableBaseMSM::msm_bigint``, ``ark-poly``
``Radix2EvaluationDomain``; ``Cargo.toml:30-42`` in the reference, semver ``"0.5"``,
no lockfile, not vendored) and holds NO literal known-answer vectors for commit /
FFT.  The results on this path are mathematically unique
(group element / field element), so this restatement is pinned by
  * public curve constants (generators on curve, group orders, roots of unity),
  * the algebraic identities the reference's own tests assert
    (``pcs/src/univariate_kzg/mod.rs:1089-1234`` commit/open;
    ``relation/src/constraint_system.rs:2262-2268`` iFFT re-evaluation),
  * MSM(powers_of_g, coeffs) == [p(beta)] g for SRS = [beta^i] g
    (definition at ``pcs/src/univariate_kzg/srs.rs:175-225``).
At the byte level versus arkworks itself: "parity unpinned" (no Rust toolchain here).

Conventions restated from the reference call sites:
  * field elements cross the boundary as little-endian u64 limbs in Montgomery form,
    R = 2^(64*limbs)  (``pcs/src/univariate_kzg/mod.rs:970-979``: ``p.x.0.0``);
  * commit(ck, poly) = sum_i coeff_i * powers_of_g[i], affine
    (``pcs/src/univariate_kzg/mod.rs:90-125``);
  * open(ck, poly, z) = commit((poly - poly(z)) / (X - z)), poly(z)
    (``pcs/src/univariate_kzg/mod.rs:144-180``);
  * fft(c)[i] = sum_j c_j (h w^i)^j, natural order, zero padded; ifft is the inverse
    (``plonk/src/proof_system/prover.rs:524-573,658-660`` call sites; h = Fr::GENERATOR).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Field:
    name: str
    p: int
    limbs64: int  # u64 limbs of the Montgomery representation

    @property
    def R(self) -> int:
        return 1 << (64 * self.limbs64)

    def to_mont(self, x: int) -> int:
        return (x % self.p) * self.R % self.p

    def from_mont(self, xm: int) -> int:
        return xm * pow(self.R, -1, self.p) % self.p

    def limbs(self, x: int) -> list[int]:
        """little-endian u64 limbs of a raw integer (no Montgomery conversion)."""
        return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(self.limbs64)]

    def mont_limbs(self, x: int) -> list[int]:
        return self.limbs(self.to_mont(x))

    @staticmethod
    def from_limbs(limbs) -> int:
        v = 0
        for i, l in enumerate(limbs):
            v |= int(l) << (64 * i)
        return v

    def inv32(self) -> int:
        """-p^{-1} mod 2^32 (the constant the 32-bit-limb CIOS reduction uses)."""
        return (-pow(self.p, -1, 1 << 32)) % (1 << 32)


@dataclass(frozen=True)
class Curve:
    name: str
    curve_id: int
    fq: Field
    fr: Field
    b: int  # y^2 = x^3 + b
    gx: int
    gy: int
    fr_generator: int  # Fr::GENERATOR (coset offset used by the prover)
    two_adicity: int
    two_adic_root: int  # Fr::TWO_ADIC_ROOT_OF_UNITY

    @property
    def g(self):
        return (self.gx, self.gy)


BN254 = Curve(
    name="bn254",
    curve_id=0,
    fq=Field("bn254_fq", 21888242871839275222246405745257275088696311157297823662689037894645226208583, 4),
    fr=Field("bn254_fr", 21888242871839275222246405745257275088548364400416034343698204186575808495617, 4),
    b=3,
    gx=1,
    gy=2,
    fr_generator=5,
    two_adicity=28,
    two_adic_root=19103219067921713944291392827692070036145651957329286315305642004821462161904,
)

BLS12_381 = Curve(
    name="bls12_381",
    curve_id=1,
    fq=Field(
        "bls12_381_fq",
        0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        6,
    ),
    fr=Field("bls12_381_fr", 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 4),
    b=4,
    gx=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    gy=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    fr_generator=7,
    two_adicity=32,
    two_adic_root=10238227357739495823651030575849232062558860180284477541189508159991286009131,
)

CURVES = {c.name: c for c in (BN254, BLS12_381)}

# ----------------------------------------------------------------------------------
# short-Weierstrass arithmetic, a = 0.  Affine points are (x, y) tuples; None = identity.
# ----------------------------------------------------------------------------------


def on_curve(cv: Curve, P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - cv.b) % cv.fq.p == 0


def ec_neg(cv: Curve, P):
    if P is None:
        return None
    return (P[0], (-P[1]) % cv.fq.p)


def ec_add(cv: Curve, P, Q):
    p = cv.fq.p
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


# Jacobian (X, Y, Z) used internally so long scalar multiplications avoid inversions.
def _jac_double(p, P):
    X, Y, Z = P
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def _jac_add(p, P, Q):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == 0:
        return Q
    if Z2 == 0:
        return P
    Z1Z1 = Z1 * Z1 % p
    Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p
    U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 * Z2Z2 % p
    S2 = Y2 * Z1 * Z1Z1 % p
    if U1 == U2:
        if S1 == S2:
            return _jac_double(p, P)
        return (1, 1, 0)
    H = (U2 - U1) % p
    Rr = (S2 - S1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = U1 * HH % p
    X3 = (Rr * Rr - HHH - 2 * V) % p
    Y3 = (Rr * (V - X3) - S1 * HHH) % p
    Z3 = Z1 * Z2 * H % p
    return (X3, Y3, Z3)


def _to_jac(P):
    return (1, 1, 0) if P is None else (P[0], P[1], 1)


def _from_jac(p, P):
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, -1, p)
    zi2 = zi * zi % p
    return (X * zi2 % p, Y * zi2 * zi % p)


def ec_mul(cv: Curve, k: int, P):
    p = cv.fq.p
    k %= cv.fr.p
    acc = (1, 1, 0)
    base = _to_jac(P)
    while k:
        if k & 1:
            acc = _jac_add(p, acc, base)
        base = _jac_double(p, base)
        k >>= 1
    return _from_jac(p, acc)


def msm_naive(cv: Curve, bases, scalars):
    """sum_i scalars[i] * bases[i]  -- the definition (mod.rs:114-118), no Pippenger."""
    p = cv.fq.p
    acc = (1, 1, 0)
    for P, k in zip(bases, scalars):
        k %= cv.fr.p
        if k == 0 or P is None:
            continue
        t = (1, 1, 0)
        b = _to_jac(P)
        while k:
            if k & 1:
                t = _jac_add(p, t, b)
            b = _jac_double(p, b)
            k >>= 1
        acc = _jac_add(p, acc, t)
    return _from_jac(p, acc)


def srs_powers(cv: Curve, beta: int, g, n: int):
    """[beta^i] g, i < n  (gen_srs_for_testing, pcs/src/univariate_kzg/srs.rs:175-225)."""
    out = []
    s = 1
    for _ in range(n):
        out.append(ec_mul(cv, s, g))
        s = s * beta % cv.fr.p
    return out


# ----------------------------------------------------------------------------------
# Multilinear KZG (pcs/src/multilinear_kzg): test SRS, commit, open
# ----------------------------------------------------------------------------------


def mle_eq_levels(cv: Curve, t):
    """scalars of MultilinearProverParam::powers_of_g (tests::gen_srs_for_testing, pcs/src/multilinear_kzg/srs.rs:
    299-358): level i holds eq_(t[i..])(x) for x in {0,1}^(nv - i), x's bit j standing for variable i + j; the last
    level is [1] (the generator itself).  eq factor of one variable: t x + (1 - t)(1 - x) (eq_extension, :280-297)."""
    r = cv.fr.p
    nv = len(t)
    levels = []
    for i in range(nv):
        vals = [1]
        for tj in t[i:]:                              # variable i first = lowest bit
            vals = [v * f % r for f in ((1 - tj) % r, tj % r) for v in vals]
        levels.append(vals)
    levels.append([1])
    return levels


def mle_srs_for_testing(cv: Curve, g, t):
    """powers_of_g[i].evals = g * eq level i"""
    return [[ec_mul(cv, e, g) for e in lvl] for lvl in mle_eq_levels(cv, t)]


def mle_evaluate(cv: Curve, evals, point) -> int:
    """DenseMultilinearExtension::evaluate: fix variable 0 (lowest index bit) first"""
    r = cv.fr.p
    f = list(evals)
    for z in point:
        f = [(f[2 * b] + (f[2 * b + 1] - f[2 * b]) * z) % r for b in range(len(f) // 2)]
    return f[0]


def mle_commit(cv: Curve, powers_of_g, num_vars: int, evals):
    """MultilinearKzgPCS::commit (pcs/src/multilinear_kzg/mod.rs:123-149)"""
    nv = len(evals).bit_length() - 1
    return msm_naive(cv, powers_of_g[num_vars - nv], evals)


def mle_open(cv: Curve, powers_of_g, num_vars: int, evals, point):
    """open_internal (pcs/src/multilinear_kzg/mod.rs:288-358) -> (proofs, evaluation)"""
    r = cv.fr.p
    nv = len(point)
    assert len(evals) == 1 << nv
    ignored = num_vars - nv + 1
    f = [e % r for e in evals]
    proofs = []
    for i, z in enumerate(point):
        k = nv - 1 - i
        q = [(f[2 * b + 1] - f[2 * b]) % r for b in range(1 << k)]
        f = [(f[2 * b] + q[b] * z) % r for b in range(1 << k)]
        proofs.append(msm_naive(cv, powers_of_g[ignored + i], q))
    return proofs, mle_evaluate(cv, evals, point)


# ----------------------------------------------------------------------------------
# KZG on top (commit / open), restating pcs/src/univariate_kzg/mod.rs:90-180
# ----------------------------------------------------------------------------------


def poly_eval(cv: Curve, coeffs, z: int) -> int:
    r = cv.fr.p
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * z + c) % r
    return acc


def commit_via_beta(cv: Curve, beta: int, g, coeffs):
    """Commitment against SRS [beta^i]g without touching the SRS: [p(beta)] g."""
    return ec_mul(cv, poly_eval(cv, coeffs, beta), g)


def poly_mul(cv: Curve, a, b):
    r = cv.fr.p
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % r
    return out


def poly_divide(cv: Curve, num, den):
    """quotient of the naive long division ark-poly's `&DensePolynomial / &DensePolynomial` runs (remainder discarded);
    den's leading coefficient must be non-zero"""
    r = cv.fr.p
    num = list(num)
    while num and num[-1] % r == 0:
        num.pop()
    if len(num) < len(den):
        return []
    q = [0] * (len(num) - len(den) + 1)
    lead_inv = pow(den[-1], -1, r)
    rem = num
    for k in range(len(q) - 1, -1, -1):
        c = rem[k + len(den) - 1] * lead_inv % r
        q[k] = c
        if c:
            for j, d in enumerate(den):
                rem[k + j] = (rem[k + j] - c * d) % r
    return q


def multi_point_open_witness(cv: Curve, coeffs, points):
    """UnivariatePCS::multi_point_open (pcs/src/univariate_kzg/mod.rs:416-468) up to the MSM: evaluations at every point
    (:433-436), the divisor prod_i (X - z_i) folded left to right (:443-447), the witness polynomial = quotient of the
    long division (:454).  -> (witness coefficients, evals)"""
    if not points:
        raise ValueError("no point to evaluate and open")   # :421-425
    r = cv.fr.p
    evals = [poly_eval(cv, coeffs, z) for z in points]
    divisor = [(-points[0]) % r, 1]
    for z in points[1:]:
        divisor = poly_mul(cv, divisor, [(-z) % r, 1])
    return poly_divide(cv, coeffs, divisor), evals


def witness_poly(cv: Curve, coeffs, z: int):
    """(f(X) - f(z)) / (X - z): synthetic division; returns (quotient, remainder=f(z))."""
    r = cv.fr.p
    n = len(coeffs)
    if n == 0:
        return [], 0
    q = [0] * (n - 1)
    carry = 0
    for i in range(n - 1, 0, -1):
        carry = (coeffs[i] + carry * z) % r
        q[i - 1] = carry
    rem = (coeffs[0] + carry * z) % r
    return q, rem


# ----------------------------------------------------------------------------------
# Radix-2 evaluation domain, arkworks conventions
# ----------------------------------------------------------------------------------


def domain_root(cv: Curve, log_n: int) -> int:
    """group_gen of Radix2EvaluationDomain of size 2^log_n."""
    assert log_n <= cv.two_adicity
    return pow(cv.two_adic_root, 1 << (cv.two_adicity - log_n), cv.fr.p)


def _fft_rec(vals, w, r):
    n = len(vals)
    if n == 1:
        return vals
    even = _fft_rec(vals[0::2], w * w % r, r)
    odd = _fft_rec(vals[1::2], w * w % r, r)
    out = [0] * n
    t = 1
    h = n // 2
    for i in range(h):
        o = odd[i] * t % r
        out[i] = (even[i] + o) % r
        out[i + h] = (even[i] - o) % r
        t = t * w % r
    return out


def fft(cv: Curve, coeffs, log_n: int, offset: int = 1):
    """evals[i] = sum_j c_j (offset * w^i)^j ; coeffs zero-extended to 2^log_n."""
    r = cv.fr.p
    n = 1 << log_n
    assert len(coeffs) <= n
    c = [x % r for x in coeffs] + [0] * (n - len(coeffs))
    if offset != 1:
        t = 1
        for j in range(n):
            c[j] = c[j] * t % r
            t = t * offset % r
    return _fft_rec(c, domain_root(cv, log_n), r)


def ifft(cv: Curve, evals, log_n: int, offset: int = 1):
    """inverse of :func:`fft` (evals must have exactly 2^log_n entries... or fewer,
    zero-extended, as ark-poly does)."""
    r = cv.fr.p
    n = 1 << log_n
    assert len(evals) <= n
    e = [x % r for x in evals] + [0] * (n - len(evals))
    winv = pow(domain_root(cv, log_n), -1, r)
    c = _fft_rec(e, winv, r)
    ninv = pow(n, -1, r)
    oinv = pow(offset, -1, r)
    t = ninv
    for j in range(n):
        c[j] = c[j] * t % r
        t = t * oinv % r
    return c


def fft_direct(cv: Curve, coeffs, log_n: int, offset: int, idxs):
    """Horner evaluation at offset*w^i for i in idxs -- independent spot check."""
    r = cv.fr.p
    w = domain_root(cv, log_n)
    return [poly_eval(cv, coeffs, offset * pow(w, i, r) % r) for i in idxs]


# ----------------------------------------------------------------------------------
# deterministic inputs (splitmix64), shared with the C restatement and the tests
# ----------------------------------------------------------------------------------


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def field(self, f: Field) -> int:
        """uniform in [0,p) by rejection on the top-masked limb draw."""
        bits = f.p.bit_length()
        while True:
            v = 0
            for i in range(f.limbs64):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < f.p:
                return v


# ----------------------------------------------------------------------------------
# TurboPlonk quotient evaluations, restating
# plonk/src/proof_system/prover.rs:587-655 (+ :665-699 gate part, :709-752 copy constraints)
# ----------------------------------------------------------------------------------


def quotient_evals_turbo(cv: Curve, log_n: int, log_m: int, sel, sig, wires, zperm, pi, k, alpha, beta, gamma,
                         alpha_base=1):
    """sel: 13 lists, sig/wires: 5 lists, zperm/pi: lists -- all coset evaluations (offset
    Fr::GENERATOR) over the size-2^log_m domain, as integers.  Returns alpha_base * t_i."""
    r = cv.fr.p
    n, m = 1 << log_n, 1 << log_m
    ratio = m // n
    wm = domain_root(cv, log_m)
    g = cv.fr_generator
    zh_inv = [pow((pow(g * pow(wm, i, r) % r, n, r) - 1) % r, -1, r) for i in range(ratio)]
    out = []
    for i in range(m):
        x = g * pow(wm, i, r) % r
        w = [wires[j][i] for j in range(5)]
        q_lc = [sel[j][i] for j in range(4)]
        q_mul = [sel[4][i], sel[5][i]]
        q_hash = [sel[6 + j][i] for j in range(4)]
        q_o, q_c, q_ecc = sel[10][i], sel[11][i], sel[12][i]
        t_circ = (q_c + pi[i] + sum(q_lc[j] * w[j] for j in range(4)) + q_mul[0] * w[0] * w[1]
                  + q_mul[1] * w[2] * w[3] + q_ecc * w[0] * w[1] * w[2] * w[3] * w[4]
                  + sum(q_hash[j] * pow(w[j], 5, r) for j in range(4)) - q_o * w[4]) % r
        a = zperm[i]
        for j in range(5):
            a = a * (w[j] + k[j] * x * beta + gamma) % r
        b = zperm[(i + ratio) % m]
        for j in range(5):
            b = b * (w[j] + sig[j][i] * beta + gamma) % r
        t_perm_1 = alpha * (a - b) % r
        t_perm_2 = alpha * alpha * (zperm[i] - 1) % r * pow(n * (x - 1) % r, -1, r) % r
        t = ((t_circ + t_perm_1) * zh_inv[i % ratio] + t_perm_2) % r
        out.append(alpha_base * t % r)
    return out


# ----------------------------------------------------------------------------------
# UltraPlonk (Plookup) quotient evaluations: the same loop with 6 wire types, 14 selectors
# (the last is q_lookup) and compute_quotient_plookup_contribution,
# plonk/src/proof_system/prover.rs:766-882 with eval_merged_table / eval_merged_lookup_witness
# (plonk/src/proof_system/structs.rs:899-931) and first_and_last_lagrange_coeffs
# (plonk/src/lagrange.rs:89-115).
# ----------------------------------------------------------------------------------

PLOOKUP_POLYS = ("range_table", "key_table", "table_dom_sep", "q_dom_sep", "h1", "h2", "prod_lookup")


def quotient_evals_ultra(cv: Curve, log_n: int, log_m: int, sel, sig, wires, zperm, pi, lookup, k, alpha, beta, gamma,
                         tau, alpha_base=1):
    """sel: 14 lists (q_lookup last), sig/wires: 6 lists, lookup: 7 lists in PLOOKUP_POLYS order."""
    r = cv.fr.p
    n, m = 1 << log_n, 1 << log_m
    ratio = m // n
    wm = domain_root(cv, log_m)
    wn_inv = pow(domain_root(cv, log_n), -1, r)
    g = cv.fr_generator
    zh_inv = [pow((pow(g * pow(wm, i, r) % r, n, r) - 1) % r, -1, r) for i in range(ratio)]
    rng_t, key_t, tds, qds, h1, h2, pl = lookup
    q_lookup = sel[13]
    out = []
    for i in range(m):
        nx = (i + ratio) % m
        x = g * pow(wm, i, r) % r
        w = [wires[j][i] for j in range(6)]
        wn = [wires[j][nx] for j in range(6)]
        t_circ = (sel[11][i] + pi[i] + sum(sel[j][i] * w[j] for j in range(4)) + sel[4][i] * w[0] * w[1]
                  + sel[5][i] * w[2] * w[3] + sel[12][i] * w[0] * w[1] * w[2] * w[3] * w[4]
                  + sum(sel[6 + j][i] * pow(w[j], 5, r) for j in range(4)) - sel[10][i] * w[4]) % r
        a = zperm[i]
        for j in range(6):
            a = a * (w[j] + k[j] * x * beta + gamma) % r
        b = zperm[nx]
        for j in range(6):
            b = b * (w[j] + sig[j][i] * beta + gamma) % r
        t_perm_1 = alpha * (a - b) % r
        t_perm_2 = alpha * alpha * (zperm[i] - 1) % r * pow(n * (x - 1) % r, -1, r) % r
        # Plookup part
        vanish = (pow(x, n, r) - 1) % r
        l1 = vanish * pow(n * (x - 1) % r, -1, r) % r
        ln = vanish * wn_inv % r * pow(n * (x - wn_inv) % r, -1, r) % r
        l1_div = l1 * pow(vanish, -1, r) % r
        ln_div = ln * pow(vanish, -1, r) % r
        mt = lambda rg, ky, ql, w3, w4, td: (rg + ql * tau * (td + tau * (ky + tau * (w3 + tau * w4)))) % r
        mt_x = mt(rng_t[i], key_t[i], q_lookup[i], w[3], w[4], tds[i])
        mt_xw = mt(rng_t[nx], key_t[nx], q_lookup[nx], wn[3], wn[4], tds[nx])
        ml_x = (w[5] + q_lookup[i] * tau * (qds[i] + tau * (w[0] + tau * (w[1] + tau * w[2])))) % r
        ap = pow(alpha, 3, r)
        res2 = ap * (h1[i] - h2[nx]) % r * ln_div % r
        ap = ap * alpha % r
        res2 += ap * (pl[i] - 1) % r * l1_div % r
        ap = ap * alpha % r
        res2 += ap * (pl[i] - 1) % r * ln_div % r
        ap = ap * alpha % r
        b1 = (1 + beta) % r
        gb = b1 * gamma % r
        p3 = (x - wn_inv) * (pl[i] * b1 % r * (gamma + ml_x) % r * (gb + mt_x + beta * mt_xw) % r
                             - pl[nx] * (gb + h1[i] + beta * h1[nx]) % r * (gb + h2[i] + beta * h2[nx])) % r
        res1 = ap * p3 % r
        t = ((t_circ + t_perm_1 + res1) * zh_inv[i % ratio] + t_perm_2 + res2) % r
        out.append(alpha_base * t % r)
    return out


# ----------------------------------------------------------------------------------
# Grand products: values on H before the reference's final ifft
# ----------------------------------------------------------------------------------


def perm_product(cv: Curve, wires, idp, sig, beta, gamma):
    """compute_prod_permutation_polynomial, relation/src/constraint_system.rs:1394-1420.
    wires / idp / sig: [nw][n] integers (witness values, extended_id_permutation, its permuted values)."""
    r = cv.fr.p
    n = len(wires[0])
    out = [1]
    for j in range(n - 1):
        a = b = 1
        for i in range(len(wires)):
            t = (wires[i][j] + gamma) % r
            a = a * (t + beta * idp[i][j]) % r
            b = b * (t + beta * sig[i][j]) % r
        out.append(out[-1] * a % r * pow(b, -1, r) % r)
    return out


def lookup_product(cv: Curve, f, t, s, beta, gamma):
    """compute_lookup_prod_polynomial, relation/src/constraint_system.rs:1516-1575.
    f: merged lookup witness [n], t: merged table [n], s: sorted vector [2n-1]."""
    r = cv.fr.p
    n = len(t)
    b1 = (1 + beta) % r
    gb = gamma * b1 % r
    out = [1]
    for j in range(n - 2):
        a = b1 * (gamma + f[j]) % r * (gb + t[j] + beta * t[j + 1]) % r
        b = (gb + s[j] + beta * s[j + 1]) % r * (gb + s[n - 1 + j] + beta * s[n + j]) % r
        out.append(out[-1] * a % r * pow(b, -1, r) % r)
    out.append(1)
    return out
