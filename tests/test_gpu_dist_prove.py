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


def _batch_worker(rank, world, port, log_n, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jellyfish_amd import dist as jdist
        from jellyfish_amd import lib as L
        from jellyfish_amd import prover_path as P
        from oracle import pyref as R
        from tests.util import mont_arr, pt_arr, rand_fr_mont

        torch.cuda.set_device(0)
        L.load_library()
        L.init(0)
        cv = R.BLS12_381
        n = 1 << log_n
        beta = mont_arr(cv.fr, [97531])[0]
        srs = L.Srs.gen_for_testing(1, beta, pt_arr(cv, cv.g), n + 8)
        rand = lambda count, seed: rand_fr_mont(cv, count, 30000 + 1000 * rank + seed)   # a different circuit per rank
        hp = P.UltraPlonkHotPath(1, log_n, srs, rand)
        res = jdist.batch_prove_one_per_rank(hp, L)
        hp.stream.sync()
        own = hp.commits.cpu().numpy().copy()
        q.put((rank, own, res["instance"].cpu().numpy().copy(), res["aggregated"].cpu().numpy().copy(), hp.rec))
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_batch_prove_one_circuit_per_rank_aggregates_by_group_sum():
    """BASELINE config 5 shape at world size 2 (both ranks on cuda:0): per-instance commitments are gathered,
    the split-quotient and opening commitments are the group sums of the per-rank contributions."""
    import torch.multiprocessing as mp

    from oracle import coracle as C

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
    assert all(len(r) == 5 for r in res), res
    rec = res[0][4]
    own = [r[1].reshape(18, rec) for r in res]
    for r in res:
        inst, agg = r[2], r[3]
        assert inst.shape == (2, 10, rec) and agg.shape == (8, rec)
        for k in range(2):
            assert (inst[k] == own[k][:10]).all()
        for j in range(8):
            a = own[0][10 + j][:96].view(np.uint64)
            b = own[1][10 + j][:96].view(np.uint64)
            assert (agg[j][:96].view(np.uint64) == C.ec_add(1, a, b)).all(), j


def _real_worker(rank, world, port, q, size=300):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jellyfish_amd import lib as L
        from jellyfish_amd import plonk as JP
        from tests.test_gpu_plonk import BETA, build

        torch.cuda.set_device(0)
        L.load_library()
        L.init(0)
        cs = build(JP.PlonkCircuit(0), size)
        srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, cs.eval_domain_size() + 2, BETA)
        pk, _ = JP.PlonkKzgSnark.preprocess(srs, cs)
        ref = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk).serialize_compressed()
        got = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk, sharded=True).serialize_compressed()
        # the proving key keeps this rank's slice of the selector / sigma coset evaluations
        pk.keep_coset_evals(rank=rank, world=world)
        kept = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk, sharded=True).serialize_compressed()
        # round-1 commitments from the witness values, the evaluation ranges partitioned over the ranks
        pk.keep_lagrange_commit_key()
        lag = JP.PlonkKzgSnark.prove(JP.H.test_rng(), cs, pk, sharded=True).serialize_compressed()
        q.put((rank, ref == got and ref == kept and ref == lag))
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, 300), (4, 40), (8, 100)])
def test_sharded_real_prove_is_byte_identical(world, size):
    """PlonkKzgSnark.prove(sharded=True) with 2 / 4 / 8 ranks sharing cuda:0 (BASELINE config 4 layout): commitment point
    ranges split over the ranks (one all_gather per commit round) and the quotient coset split into one slice per rank
    (slice-local coset NTTs and quotient evaluation, one all_gather of the slices' inverse transforms;
    world 8 = m / n exercises the fold of the n + 3 coefficients mod X^n - c).  Proof bytes equal the single-device
    proof, with and without the proving key's resident coset evaluations."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, q, size)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, True) for r in range(world)], res


def _real_batch_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jellyfish_amd import lib as L
        from jellyfish_amd import plonk as JP
        from tests.test_gpu_plonk import BETA, build

        torch.cuda.set_device(0)
        L.load_library()
        L.init(0)
        sizes = [60, 80, 100, 70][:world]
        css = [build(JP.PlonkCircuit(0), s) for s in sizes]
        srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, css[0].eval_domain_size() + 2, BETA)
        keys = [JP.PlonkKzgSnark.preprocess(srs, c) for c in css]
        ref = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, [k[0] for k in keys], b"cfg5").serialize_compressed()
        # the other multi-GPU layout of the same call: every rank holds every circuit, commitments' point ranges and the
        # quotient coset are split over the ranks (batch_prove(sharded=True))
        shard = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, [k[0] for k in keys], b"cfg5", sharded=True)
        assert shard.serialize_compressed() == ref, "sharded batch_prove differs"
        got = JP.PlonkKzgSnark.batch_prove_one_per_rank(JP.H.test_rng(), css[rank], keys[rank][0], b"cfg5")
        keys[rank][0].keep_coset_evals()
        kept = JP.PlonkKzgSnark.batch_prove_one_per_rank(JP.H.test_rng(), css[rank], keys[rank][0], b"cfg5")
        q.put((rank, ref == got.serialize_compressed() and ref == kept.serialize_compressed() and len(got) == world))
    except Exception as e:
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_real_batch_prove_one_circuit_per_rank_is_byte_identical(world):
    """PlonkKzgSnark.batch_prove_one_per_rank (BASELINE config 5's layout: one circuit per GPU, here the ranks share
    cuda:0): every rank returns the BatchProof that batch_prove of all the circuits gives on one GPU from the same RNG
    state -- per-instance commitments / evaluations gathered, aggregated commitments as group sums of per-rank shares."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_batch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, True) for r in range(world)], res


def _ultra_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jellyfish_amd import lib as L
        from jellyfish_amd import plonk as JP
        from tests.test_gpu_plonk import BETA
        from tests.util import build_ultra_circuit

        torch.cuda.set_device(0)
        L.load_library()
        L.init(0)
        # one instance more than ranks: rank 0 proves instances 0 and `world`
        css = [build_ultra_circuit(JP.PlonkCircuit.new_ultra_plonk(1, 4), 6, e) for e in range(world + 1)]   # b = 5 + e < 16
        beta = BETA % css[0].r
        srs = JP.PlonkKzgSnark.universal_setup_for_testing(1, css[0].eval_domain_size() + 2, beta)
        keys = [JP.PlonkKzgSnark.preprocess(srs, c) for c in css]
        pks = [k[0] for k in keys]
        ref = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css[:world], pks[:world], b"cfg5").serialize_compressed()
        # (a) BASELINE config 5's layout: one UltraPlonk circuit per rank
        got = JP.PlonkKzgSnark.batch_prove_one_per_rank(JP.H.test_rng(), css[rank], pks[rank], b"cfg5")
        ok = got.serialize_compressed() == ref and got.plookup_proofs_vec is not None and len(got) == world
        # (b) point-range sharded commitments + coset-sliced quotient of the same batch
        shard = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css[:world], pks[:world], b"cfg5", sharded=True)
        ok = ok and shard.serialize_compressed() == ref
        # (c) uneven placement: world + 1 instances, the last one on rank 0 as well
        ref2 = JP.PlonkKzgSnark.batch_prove(JP.H.test_rng(), css, pks).serialize_compressed()
        owners = list(range(world)) + [0]
        mine_c = [c if owners[i] == rank else None for i, c in enumerate(css)]
        mine_k = [k if owners[i] == rank else None for i, k in enumerate(pks)]
        placed = JP.PlonkKzgSnark.batch_prove_placed(JP.H.test_rng(), mine_c, mine_k, owners)
        ok = ok and placed.serialize_compressed() == ref2
        # (d) a parameter error on ONE rank is raised on every rank (no rank left waiting in a collective)
        bad_k = list(mine_k)
        if rank == 1:
            bad_k[1] = None
        try:
            JP.PlonkKzgSnark.batch_prove_placed(JP.H.test_rng(), mine_c, bad_k, owners)
            ok = False
        except JP.PlonkError as e:
            ok = ok and "no circuit / proving key" in str(e)
        q.put((rank, ok))
    except Exception as e:
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ultraplonk_batch_prove_across_ranks_is_byte_identical(world):
    """BASELINE config 5 (UltraPlonk, BLS12-381, batch_prove with one circuit per GPU) with the ranks sharing cuda:0:
    one-per-rank, sharded and unevenly placed batches all give the bytes of the one-GPU batch_prove; world 8 = m / n
    runs the coset-slice fold of the Plookup polynomials (n + 3 coefficients mod X^n - c)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ultra_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, True) for r in range(world)], res
