"""GPU parity: field and group arithmetic of the HIP kernels vs the oracle, through the C ABI.
Bar: bit-exact (integer arithmetic)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import FIELD_ID, golden, gpt, h2i, mont_arr, pt_arr, pt_from

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]
FIELDS = {x.name: x for cv in CURVES for x in (cv.fr, cv.fq)}


@pytest.mark.parametrize("fname", list(FIELD_ID))
def test_field_ops_golden_and_random(hip, fname):
    f = FIELDS[fname]
    fid = FIELD_ID[fname]
    g = golden("field")[fname]
    a = mont_arr(f, [h2i(x) for x in g["a"]])
    b = mont_arr(f, [h2i(x) for x in g["b"]])
    for op in ("add", "sub", "mul"):
        assert (hip.field_op(fid, op, a, b) == mont_arr(f, [h2i(x) for x in g[op]])).all(), op
    for op in ("neg", "inv"):
        assert (hip.field_op(fid, op, a) == mont_arr(f, [h2i(x) for x in g[op]])).all(), op
    # random bulk vs the C oracle
    rng = R.SplitMix64(99 + fid)
    n = 4096
    xs = [rng.field(f) for _ in range(n)]
    ys = [rng.field(f) for _ in range(n)]
    A, B = mont_arr(f, xs), mont_arr(f, ys)
    for op in ("add", "sub", "mul"):
        assert (hip.field_op(fid, op, A, B) == C.fp_binop(fid, op, A, B)).all(), op
    assert (hip.field_op(fid, "neg", A) == C.fp_unop(fid, "neg", A)).all()
    assert (hip.field_op(fid, "inv", A[:256]) == C.fp_unop(fid, "inv", A[:256])).all()
    # the Euclid-type inversions the kernels use (normalisation of every MSM result, batch inversions): edge values too
    E = mont_arr(f, [0, 1, 2, f.p - 1, f.p - 2, (f.p - 1) // 2, (f.p + 1) // 2, 1 << 200, (1 << 253) - 1] + xs[:2039])
    want = C.fp_unop(fid, "inv", E)
    assert (hip.field_op(fid, "inv_safegcd", E) == want).all()
    assert (hip.field_op(fid, "inv_binary", E) == want).all()
    assert (hip.field_op(fid, "from_mont", A) == C.fp_unop(fid, "from_mont", A)).all()
    assert (hip.field_op(fid, "to_mont", A) == C.fp_unop(fid, "to_mont", A)).all()
    # outputs fully reduced
    out = hip.field_op(fid, "mul", A, B)
    vals = [R.Field.from_limbs(row) for row in out]
    assert max(vals) < f.p


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_group_ops(hip, cv):
    k = golden("constants")[cv.name]
    G = pt_arr(cv, cv.g)
    assert pt_from(cv, hip.g1_add(cv.curve_id, G, G)) == gpt(k["g2"])
    neg = pt_arr(cv, R.ec_neg(cv, cv.g))
    assert pt_from(cv, hip.g1_add(cv.curve_id, G, neg)) is None
    zero = np.zeros_like(G)
    assert (hip.g1_add(cv.curve_id, G, zero) == G).all()
    assert (hip.g1_add(cv.curve_id, zero, G) == G).all()
    rng = R.SplitMix64(5)
    for _ in range(3):
        s = rng.field(cv.fr)
        sm = mont_arr(cv.fr, [s])[0]
        assert (hip.g1_mul(cv.curve_id, G, sm) == C.ec_mul(cv.curve_id, G, sm)).all()
    # order * G = identity
    rm1 = mont_arr(cv.fr, [cv.fr.p - 1])[0]
    assert pt_from(cv, hip.g1_mul(cv.curve_id, G, rm1)) == R.ec_neg(cv, cv.g)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_group_ops_29bit_limbs(hip, cv):
    """The MSM tail's group law on lazily reduced 29-bit limbs (csrc/msm.hip p29_add / p29_double) against the oracle:
    the rare branches (P + P, P - P, identity on either side) and 256-step double-and-add chains whose every output is
    fed back in, which is where a lazy-reduction bound that is too optimistic would show."""
    cid = cv.curve_id
    G = pt_arr(cv, cv.g)
    zero = np.zeros_like(G)
    neg = pt_arr(cv, R.ec_neg(cv, cv.g))
    assert (hip.g1_add(cid, G, G, limbs29=True) == hip.g1_add(cid, G, G)).all()
    assert pt_from(cv, hip.g1_add(cid, G, neg, limbs29=True)) is None
    assert (hip.g1_add(cid, G, zero, limbs29=True) == G).all()
    assert (hip.g1_add(cid, zero, G, limbs29=True) == G).all()
    rng = R.SplitMix64(77)
    scal = [1, 2, 3, cv.fr.p - 1, cv.fr.p - 2, (1 << 200) - 1] + [rng.field(cv.fr) for _ in range(6)]
    pts = []
    for s in scal:
        sm = mont_arr(cv.fr, [s])[0]
        got = hip.g1_mul(cid, G, sm, limbs29=True)
        assert (got == C.ec_mul(cid, G, sm)).all(), hex(s)
        assert (hip.g1_mul(cid, G, sm, limbs29="naf") == got).all(), hex(s)
        pts.append(got)
    for a, b in zip(pts, pts[1:]):
        assert (hip.g1_add(cid, a, b, limbs29=True) == C.ec_add(cid, a, b)).all()
    # a chain on a random base point rather than the generator
    sm = mont_arr(cv.fr, [scal[-1]])[0]
    assert (hip.g1_mul(cid, pts[-2], sm, limbs29=True) == C.ec_mul(cid, pts[-2], sm)).all()
    assert (hip.g1_mul(cid, pts[-2], sm, limbs29="naf") == C.ec_mul(cid, pts[-2], sm)).all()
    zero = mont_arr(cv.fr, [0])[0]
    assert pt_from(cv, hip.g1_mul(cid, G, zero, limbs29="naf")) is None


@pytest.mark.parametrize("fname", list(FIELD_ID))
def test_unsaturated_29bit_arithmetic(hip, fname):
    """csrc/fp29.hpp (29-bit limbs, lazy reduction, R' = 2^(29 L)) agrees with the saturated arithmetic / the oracle
    after conversion back: products, squares, differences and a chain that exercises the lazy bounds; edge values
    0, 1, p - 1 included."""
    f = FIELDS[fname]
    fid = FIELD_ID[fname]
    p = f.p
    rng = R.SplitMix64(4242 + fid)
    n = 4096
    xs = [0, 1, p - 1, p - 1, 0, 2, p - 2] + [rng.field(f) for _ in range(n - 7)]
    ys = [0, p - 1, p - 1, 1, 5, p - 1, p - 2] + [rng.field(f) for _ in range(n - 7)]
    A, B = mont_arr(f, xs), mont_arr(f, ys)
    assert (hip.field_op(fid, "mul29", A, B) == C.fp_binop(fid, "mul", A, B)).all()
    assert (hip.field_op(fid, "sqr29", A, B) == C.fp_binop(fid, "mul", A, A)).all()
    assert (hip.field_op(fid, "sub29", A, B) == C.fp_binop(fid, "sub", A, B)).all()
    want = [(((x * y - 2 * x) * (x * x - y)) * y - x * y) % p for x, y in zip(xs, ys)]
    assert (hip.field_op(fid, "chain29", A, B) == mont_arr(f, want)).all()
    want = [(x * y + (x * x - y) * y) % p for x, y in zip(xs, ys)]
    assert (hip.field_op(fid, "dot29", A, B) == mont_arr(f, want)).all()
