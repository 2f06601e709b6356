"""GPU parity for the multilinear-KZG MSM sites (pcs/src/multilinear_kzg/mod.rs:123-149 commit, :288-358 open_internal)
through the C ABI: jfhip_msm against a level of the prover parameter, jfhip_mle_open, jfhip_mle_fold_device.
The checker is the oracle's restatement (oracle/pyref.py mle_*, pinned by the scheme's identities in tests/test_oracle.py)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import from_mont_arr, mont_arr, pt_arr, pt_from, rand_fr_mont

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


def _load_levels(hip, cv, powers_of_g):
    return [hip.Srs.load(cv.curve_id, np.stack([pt_arr(cv, P) for P in lvl])) for lvl in powers_of_g]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_mle_commit_open_structured_srs(hip, cv):
    """the reference's test SRS shape (eq-basis of a random point t, srs.rs:299-358), num_vars = 5; polynomials with 5, 3
    and 1 variables (the `ignored` levels of commit / open); compared with the big-integer restatement"""
    num_vars = 5
    rng = R.SplitMix64(4100 + cv.curve_id)
    t = [rng.field(cv.fr) for _ in range(num_vars)]
    powers = R.mle_srs_for_testing(cv, cv.g, t)
    levels = _load_levels(hip, cv, powers)
    for nv in (5, 3, 1):
        evals = [rng.field(cv.fr) for _ in range(1 << nv)]
        if nv == 3:
            evals[2] = 0                                   # a zero entry
        point = [rng.field(cv.fr) for _ in range(nv)]
        ev_m, pt_m = mont_arr(cv.fr, evals), mont_arr(cv.fr, point)
        com, inf = hip.msm(levels[num_vars - nv], ev_m)
        assert pt_from(cv, com) == R.mle_commit(cv, powers, num_vars, evals) and not inf
        ignored = num_vars - nv + 1
        proofs, infs, val = hip.mle_open(levels[ignored:ignored + nv], ev_m, pt_m)
        want_proofs, want_val = R.mle_open(cv, powers, num_vars, evals, point)
        assert from_mont_arr(cv.fr, val)[0] == want_val
        assert [pt_from(cv, p) for p in proofs] == want_proofs
        assert not infs.any()
    # the zero polynomial: every proof is the identity, the evaluation 0
    proofs, infs, val = hip.mle_open(levels[1:6], np.zeros((32, 4), dtype=np.uint64), mont_arr(cv.fr, [3, 4, 5, 6, 7]))
    assert infs.all() and not proofs.any() and not val.any()
    # a constant polynomial (no variables): evaluation only
    proofs, infs, val = hip.mle_open([], mont_arr(cv.fr, [99]), np.zeros((0, 4), dtype=np.uint64))
    assert proofs.shape[0] == 0 and from_mont_arr(cv.fr, val)[0] == 99
    # a level too short for its round is refused
    with pytest.raises(hip.JfHipError):
        hip.mle_open(levels[2:6] + [levels[5]], mont_arr(cv.fr, list(range(32))), mont_arr(cv.fr, [1, 2, 3, 4, 5]))
    for lv in levels:
        lv.free()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_mle_open_vs_c_oracle(hip, cv):
    """14 variables; the levels are arbitrary point sets (ranges of a device-built [beta^i] g), which is all the MSM
    sites see; folds restated with the C oracle's field ops, proofs with its MSM"""
    nv = 14
    fid = 0 if cv.curve_id == 0 else 2
    beta = mont_arr(cv.fr, [0x5EED + cv.curve_id])[0]
    big = hip.Srs.gen_for_testing(cv.curve_id, beta, pt_arr(cv, cv.g), 1 << nv)
    pts = big.read(0, 1 << nv)
    levels, level_pts, at = [], [], 0
    for i in range(nv):
        half = 1 << (nv - 1 - i)
        level_pts.append(pts[at:at + half])
        levels.append(hip.Srs.load(cv.curve_id, level_pts[-1]))
        at += half
    evals = rand_fr_mont(cv, 1 << nv, 31)
    point = rand_fr_mont(cv, nv, 32)
    proofs, infs, val = hip.mle_open(levels, evals, point)
    f = evals
    for i in range(nv):
        half = 1 << (nv - 1 - i)
        q = C.fp_binop(fid, "sub", f[1::2], f[0::2])
        f = C.fp_binop(fid, "add", f[0::2], C.fp_binop(fid, "mul", q, np.broadcast_to(point[i], (half, 4)).copy()))
        assert (proofs[i] == C.msm(cv.curve_id, level_pts[i], q)).all(), i
    assert (val == f[0]).all() and not infs.any()
    for lv in levels:
        lv.free()
    big.free()


def test_mle_fold_device(hip):
    import torch

    cv = R.BLS12_381
    half = 1000
    f = rand_fr_mont(cv, 2 * half, 5)
    z = rand_fr_mont(cv, 1, 6)[0]
    d_f = torch.from_numpy(f.view(np.int64)).cuda()
    d_q = torch.zeros(half * 4, dtype=torch.int64, device="cuda")
    d_r = torch.zeros(half * 4, dtype=torch.int64, device="cuda")
    hip.mle_fold_device(cv.curve_id, d_f.data_ptr(), half, z, d_q.data_ptr(), d_r.data_ptr())
    torch.cuda.synchronize()
    q = C.fp_binop(2, "sub", f[1::2], f[0::2])
    r = C.fp_binop(2, "add", f[0::2], C.fp_binop(2, "mul", q, np.broadcast_to(z, (half, 4)).copy()))
    assert (d_q.cpu().numpy().view(np.uint64).reshape(half, 4) == q).all()
    assert (d_r.cpu().numpy().view(np.uint64).reshape(half, 4) == r).all()
