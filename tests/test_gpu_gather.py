"""GPU: jfhip_witness_gather_device (the witness[var] gather of Arithmetization::compute_wire_polynomials,
relation/src/constraint_system.rs:1422-1444) against numpy indexing, bit for bit; ragged sizes, repeated and boundary
indices; an index beyond the table is reported with check=True instead of being read."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,count", [(1, 1), (2, 7), (1000, 12345), (1 << 16, (1 << 18) + 3)])
def test_witness_gather_matches_numpy(hip, rows, count):
    import torch

    rng = np.random.default_rng(rows + count)
    table = rng.integers(0, 1 << 63, size=(rows, 4), dtype=np.int64)
    idx = rng.integers(0, rows, size=count, dtype=np.int64)
    idx[0], idx[-1] = rows - 1, 0
    d_table = torch.from_numpy(table).cuda()
    d_idx = torch.from_numpy(idx.astype(np.int32)).cuda()
    out = torch.zeros(count, 4, dtype=torch.int64, device="cuda")
    st = hip.Stream()
    hip.witness_gather_device(d_table.data_ptr(), rows, d_idx.data_ptr(), count, out.data_ptr(), check=True, stream=st)
    assert (out.cpu().numpy() == table[idx]).all()
    # asynchronous form, then an index one past the table
    out.zero_()
    hip.witness_gather_device(d_table.data_ptr(), rows, d_idx.data_ptr(), count, out.data_ptr(), stream=st)
    st.sync()
    assert (out.cpu().numpy() == table[idx]).all()
    d_idx[count // 2] = rows
    with pytest.raises(hip.JfHipError, match="wire variable index"):
        hip.witness_gather_device(d_table.data_ptr(), rows, d_idx.data_ptr(), count, out.data_ptr(), check=True, stream=st)
    hip.witness_gather_device(d_table.data_ptr(), rows, d_idx.data_ptr(), 0, out.data_ptr(), check=True, stream=st)
    st.close()
