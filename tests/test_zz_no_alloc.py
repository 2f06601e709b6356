"""The entry point that allocates nothing (jfhip_poly_divide_linear_noalloc_device, include/jfhip.h) against the
existing jfhip_poly_divide_linear_device through the C ABI: bit-exact, on two calls in a row, the second with a smaller
input (the static workspace keeps the first call's contents).  The oracle checks both as well.  Collected last on
purpose (file name): the round's end runs pytest -x."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests.util import rand_fr_mont

pytestmark = pytest.mark.gpu
CURVES = [R.BN254, R.BLS12_381]


def _divide(hip, cv, d_in, n, z, st, noalloc):
    import torch

    out = torch.full((n * 4,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    fn = hip.poly_divide_linear_noalloc_device if noalloc else hip.poly_divide_linear_device
    fn(cv.curve_id, d_in.data_ptr(), n, z, out.data_ptr(), stream=st)
    st.sync()
    return out.cpu().numpy().view(np.uint64).reshape(n, 4)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_no_alloc_divide_matches_existing_entry_point(hip, cv):
    import torch

    st = hip.Stream()
    # first a three-level size, then a smaller two-level one, then a single chunk
    for k, n in enumerate([300001, 4097, 37]):
        coeffs = rand_fr_mont(cv, n, 4100 + k)
        z = rand_fr_mont(cv, 1, 4200 + k)[0]
        d_in = torch.from_numpy(coeffs.view(np.int64).reshape(-1)).cuda()
        got = _divide(hip, cv, d_in, n, z, st, noalloc=True)
        ref = _divide(hip, cv, d_in, n, z, st, noalloc=False)
        assert (got == ref).all(), (cv.name, n)
        q, ev = C.open_witness(cv.curve_id, coeffs, z)
        assert (got[0] == ev).all() and (got[1:] == q).all(), (cv.name, n)
    st.close()


def test_no_alloc_divide_limits(hip):
    import torch

    st = hip.Stream()
    buf = torch.zeros(8, dtype=torch.int64, device="cuda")
    z = rand_fr_mont(R.BN254, 1, 1)[0]
    with pytest.raises(hip.JfHipError, match="exceed the fixed workspace"):
        hip.poly_divide_linear_noalloc_device(0, buf.data_ptr(), (1 << 23) + 1, z, buf.data_ptr(), stream=st)
    with pytest.raises(hip.JfHipError, match="stream"):
        hip.poly_divide_linear_noalloc_device(0, buf.data_ptr(), 1, z, buf.data_ptr(), stream=None)
    # n = 0 writes nothing and is fine
    hip.poly_divide_linear_noalloc_device(0, None, 0, z, None, stream=st)
    st.close()


def test_no_alloc_divide_at_the_limit(hip):
    """n = 2^23: the whole static workspace is in use; checked against the existing entry point and f(z) by Horner chunks"""
    import torch

    cv = R.BN254
    n = 1 << 23
    st = hip.Stream()
    g = torch.Generator(device="cpu").manual_seed(7)
    raw = torch.randint(0, 1 << 62, (n * 4,), dtype=torch.int64, generator=g)
    raw[3::4] &= (1 << 59) - 1  # < 2^251 < r: valid Montgomery residues
    d_in = raw.cuda()
    z = rand_fr_mont(cv, 1, 9)[0]
    got = _divide(hip, cv, d_in, n, z, st, noalloc=True)
    ref = _divide(hip, cv, d_in, n, z, st, noalloc=False)
    assert (got == ref).all()
    # a smaller call afterwards still matches
    got2 = _divide(hip, cv, d_in, 1000, z, st, noalloc=True)
    ref2 = _divide(hip, cv, d_in, 1000, z, st, noalloc=False)
    assert (got2 == ref2).all()
    st.close()


def test_no_alloc_divide_leaves_device_memory_unchanged_and_orders_streams(hip):
    """the property the entry point exists for: after one warm-up call, a run of calls on fresh streams changes neither
    the free device memory nor the streams' (empty) workspaces; and calls that overlap on two streams of one GPU, which
    share the static array, are ordered by the library: both results stay exact"""
    import torch

    cv = R.BN254
    n = 200000
    coeffs = rand_fr_mont(cv, n, 4300)
    zs = rand_fr_mont(cv, 2, 4301)
    d_in = torch.from_numpy(coeffs.view(np.int64).reshape(-1)).cuda()
    outs = [torch.zeros(n * 4, dtype=torch.int64, device="cuda") for _ in range(2)]
    sts = [hip.Stream(), hip.Stream()]
    hip.poly_divide_linear_noalloc_device(cv.curve_id, d_in.data_ptr(), n, zs[0], outs[0].data_ptr(), stream=sts[0])
    sts[0].sync()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(6):
        for k in range(2):   # enqueued back to back on two streams: they overlap unless the library orders them
            hip.poly_divide_linear_noalloc_device(cv.curve_id, d_in.data_ptr(), n, zs[k], outs[k].data_ptr(),
                                                  stream=sts[k])
    for st in sts:
        st.sync()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    for k in range(2):
        q, ev = C.open_witness(cv.curve_id, coeffs, zs[k])
        got = outs[k].cpu().numpy().view(np.uint64).reshape(n, 4)
        assert (got[0] == ev).all() and (got[1:] == q).all(), k
    for st in sts:
        st.close()
