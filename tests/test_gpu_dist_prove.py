"""GPU: the sharded-commit TurboPlonk sequence with TWO ranks sharing cuda:0 (gloo rendezvous, so no
second GPU is needed) gives the same 13 commitments as the single-device sequence."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, log_n, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jellyfish_amd import lib as L
        from jellyfish_amd import prover_path as P
        from oracle import pyref as R
        from tests.util import mont_arr, pt_arr, rand_fr_mont

        torch.cuda.set_device(0)
        L.load_library()
        L.init(0)
        cv = R.BN254
        n = 1 << log_n
        beta = mont_arr(cv.fr, [13579])[0]
        srs = L.Srs.gen_for_testing(0, beta, pt_arr(cv, cv.g), n + 8)
        rand = lambda count, seed: rand_fr_mont(cv, count, 6000 + seed)
        ref = P.TurboPlonkHotPath(0, log_n, srs, rand).prove().cpu().numpy().copy()
        got = P.TurboPlonkHotPath(0, log_n, srs, rand, sharded=True).prove().cpu().numpy().copy()
        rec = ref.size // 13
        ok = all((ref[j * rec:j * rec + 68] == got[j * rec:j * rec + 68]).all() for j in range(13))
        q.put((rank, bool(ok)))
    except Exception as e:  # report instead of hanging the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_prove_matches_single():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, True), (1, True)], res
