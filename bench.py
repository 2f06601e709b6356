#!/usr/bin/env python3
"""bench.py -- jellyfish PLONK proving hot path on MI355X: MSM throughput (Mpts/s).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.
A "step" is one BN254 G1 MSM (the arithmetic of UnivariateKzgPCS::commit, reference
pcs/src/univariate_kzg/mod.rs:90-125) over 2^24 points with the SRS and the scalars
already resident in HBM.  At N > 1 the SAME 2^24-point MSM is partitioned over the ranks by point
range (2^24 / N points per GPU, strong scaling, the same group element as at N = 1) and the only
exchange is an RCCL all_gather of one group element per GPU; the weak-scaling
figure (2^24 points per GPU) is kept as the extra key `weak_scaling`.

Extra objects on the same line:
  roofline      dominant kernel (bucket accumulation), INT32 VALU multiply-add rate vs the
                v_mad_u64_u32 peak measured live on the same device (SURVEY.md 8d names INT32
                VALU, not HBM or MFMA, as the bound of this kernel); hbm_view gives the same
                launch against the 8 TB/s HBM peak.
  cpu_baseline  the CPU restatement (oracle/, kind "port") timed on the host cores on a bounded
                sample of the same workload.  The reference's own rayon/arkworks prover cannot be
                built here (no Rust toolchain), see DESIGN.md.
  ntt           BASELINE config 3 (8 x 2^22 BLS12-381 Fr coset NTT, forward + inverse) measured
                in the same run with its HBM roofline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 24
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

# BN254 G1 generator (1, 2), Montgomery limbs
BN254_G = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f,
                    0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def kernel_source_sha() -> str:
    """fingerprint of the kernel sources: the committed PMC summaries (profiles/pmc_*latest.json) carry the one they were
    taken on, and a bench line only presents their counters as current when it matches"""
    import hashlib

    h = hashlib.sha256()
    for f in ("msm.hip", "ntt.hip", "fp.hpp", "fp29.hpp", "curve.hpp", "g29.hpp"):
        with open(os.path.join(ROOT, "jellyfish_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def uniform_fr(n: int, seed: int, modulus: int) -> np.ndarray:
    """n uniform field elements (rejection sampling), as 4 x u64 limbs.  Any value < r is a valid
    Montgomery representative, so these are uniform scalars in the boundary's format."""
    rng = np.random.default_rng(seed)
    bits = modulus.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    p = np.array([(modulus >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= top_mask
    while True:
        ge = np.zeros(n, dtype=bool)
        decided = np.zeros(n, dtype=bool)
        for l in (3, 2, 1, 0):
            gt = (out[:, l] > p[l]) & ~decided
            lt = (out[:, l] < p[l]) & ~decided
            ge |= gt
            decided |= gt | lt
        ge |= ~decided
        bad = np.nonzero(ge)[0]
        if bad.size == 0:
            return out
        fresh = rng.integers(0, 1 << 64, size=(bad.size, 4), dtype=np.uint64)
        fresh[:, 3] &= top_mask
        out[bad] = fresh


def max_over_ranks(torch, dist, dt: float) -> float:
    """MAX over ranks of a wall time (device tensor for RCCL, host tensor for the gloo rehearsal)"""
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def prove_section(L, torch, dist, world, stream, beta):
    """Device-side operation sequence of one TurboPlonk proof, 2^20 gates, BN254 (BASELINE config 4 shape); host
    transcript / RNG / grand product are inputs (jellyfish_amd/prover_path.py).  world > 1: MSM point ranges sharded
    over the ranks (SRS replicated, one all_gather of partial sums per commit round); time = max over ranks."""
    import hashlib

    from jellyfish_amd import prover_path as PP

    log_g = 20
    srs_p = L.Srs.gen_for_testing(0, beta, BN254_G, (1 << log_g) + 8)
    hp = PP.TurboPlonkHotPath(0, log_g, srs_p, lambda cnt, seed: uniform_fr(cnt, 7000 + seed, BN254_R), stream,
                              sharded=world > 1)
    hp.prove()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        commits = hp.prove()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dt = max_over_ranks(torch, dist, dt)
        ts.append(dt * 1e3)
    sec = {
        "workload": "TurboPlonk BN254 2^20 gates, one instance: 7 iNTT(2^20) + 25 coset NTT(2^23, zero-extended) "
                    "+ permutation grand product + quotient evaluation + coset iNTT(2^23) + 13 commitments (~2^20 points) "
                    "+ 10 evaluations + 2 linear combinations + 2 divisions, all device resident",
        "n_gpus": world,
        "sharding": "single GPU" if world == 1 else "commitment point ranges over %d ranks (SRS replicated), "
                    "transforms and quotient replicated" % world,
        "ms": round(float(np.median(ts)), 3), "runs_ms": [round(t, 3) for t in ts],
        "excluded": "host transcript and RNG (challenges / blinding are inputs); witness upload",
        "note": "round 1's kernel sequence, kept as a fixed reference point: every polynomial transformed on all 8n points "
                "of the quotient coset.  plonk_prove (the real prover) evaluates the quotient on 6n points and skips "
                "identically-zero polynomials, which is why it is faster than this section",
        "commitments_sha": hashlib.sha256(bytes(commits.cpu().numpy().tobytes())).hexdigest()[:16]}
    srs_p.free()
    del hp
    torch.cuda.empty_cache()
    return sec


def real_prove_section(L, torch, dist, world, log_g=20):
    """BASELINE config 4: PlonkKzgSnark::prove of the reference bench circuit (plonk/benches/bench.rs:29-46, a chain of
    2^20 - 10 additions), BN254, StandardTranscript (Merlin), blinding from test_rng -- jellyfish_amd/plonk.py.  The
    timed region is prove() as the reference bench times it (bench.rs:59-66): witness upload, every device step, the
    host transcript and RNG, 13 commitment read-backs.  world > 1: commitment point ranges sharded over the ranks."""
    import hashlib

    from jellyfish_amd import hostglue as H
    from jellyfish_amd import plonk as JP

    t0 = time.perf_counter()
    cs = JP.PlonkCircuit.new_turbo_plonk(0)
    a = cs.zero()
    for _ in range((1 << log_g) - 10):
        a = cs.add(a, cs.one())
    cs.finalize_for_arithmetization()
    n = cs.eval_domain_size()
    wit = H.fr_to_mont(cs.cp, cs.witness)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, n + 2, 0x6A660004)
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    rng = H.test_rng()
    proof = JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit, sharded=world > 1)
    first = proof.serialize_compressed()

    def timed(reps=5):
        ts = []
        for _ in range(reps):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit, sharded=world > 1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                dt = max_over_ranks(torch, dist, dt)
            ts.append(dt * 1e3)
        return ts

    ts = timed()
    # the same with every polynomial transformed, as the reference does (prover.rs:534-549): by default prove() fills the
    # coset evaluations of identically-zero polynomials (selectors of gate kinds the circuit never uses, an empty public
    # input) instead of transforming them -- same values, same proof bytes
    zero_sel = int(sum(pk.selector_is_zero))
    JP.SKIP_ZERO_POLYS = False
    try:
        ts_all = timed(3)
    finally:
        JP.SKIP_ZERO_POLYS = True
    # third mode (one GPU): the round-1 commitments as MSMs over the witness VALUES on the Lagrange-basis form of the commit
    # key (ProvingKey.keep_lagrange_commit_key: one group DFT of the SRS at set-up, a second window table) -- same group
    # elements, same proof bytes; what it saves depends on how small the witness values are (this circuit: counters,
    # ones and zeros), a dense witness gains nothing
    ts_lag, lag_same, lag_setup_s = None, None, None
    if world == 1:
        t0 = time.perf_counter()
        pk.keep_lagrange_commit_key()
        torch.cuda.synchronize()
        lag_setup_s = time.perf_counter() - t0
        lag_same = JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk, witness_mont=wit).serialize_compressed() == \
            JP.PlonkKzgSnark.prove(H.test_rng(), cs, pk.drop_lagrange_commit_key(), witness_mont=wit).serialize_compressed()
        pk.keep_lagrange_commit_key()
        JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
        ts_lag = timed()
        pk.drop_lagrange_commit_key()
    # second mode: the proving key keeps its 18 selector / sigma coset evaluations (this rank's slice) in HBM
    rank = dist.get_rank() if world > 1 else 0
    pk.keep_coset_evals(rank=rank, world=world)
    kept_bytes = pk.coset_evals_bytes()
    JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit, sharded=world > 1)
    ts_kept = timed()
    ts_both = None
    if world == 1:      # both options together
        pk.keep_lagrange_commit_key()
        JP.PlonkKzgSnark.prove(rng, cs, pk, witness_mont=wit)
        ts_both = timed()
        pk.drop_lagrange_commit_key()
    m = pk.quot_domain_size()
    coset_sharded = JP.coset_sharding_applies(n, m, world)
    sec = {"workload": "PlonkKzgSnark::prove, TurboPlonk BN254, gen_circuit_for_bench(2^%d gates), StandardTranscript, "
                       "test SRS of degree n + 2" % log_g,
           "n_gpus": world, "domain_size": n,
           "sharding": "single GPU" if world == 1 else
                       ("commitment point ranges over %d ranks (SRS replicated, one all_gather of partial sums per commit "
                        "round); " % world)
                       + ("quotient coset split into %d slices (slice-local coset NTTs + quotient evaluation, one "
                          "all_gather of %d x %d MiB slice polynomials); size-n iNTTs, grand product, evaluations and "
                          "opening polynomials on every rank" % (world, world, (m // world * 32) >> 20)
                          if coset_sharded else "transforms and quotient on every rank"),
           "ms": round(float(np.median(ts)), 3), "runs_ms": [round(t, 3) for t in ts],
           "ms_all_25_coset_ntts": round(float(np.median(ts_all)), 3),
           "zero_polynomials": "%d of the 13 selector polynomials of this circuit are identically zero (it has addition "
                               "and constant gates only) and the public input is empty: 'ms' fills those %d rows of coset "
                               "evaluations with zeros instead of transforming them (same proof bytes, "
                               "tests/test_gpu_plonk.py); 'ms_all_25_coset_ntts' transforms all of them like the reference"
                               % (zero_sel, zero_sel + 1),
           "ms_pk_coset_evals_resident": round(float(np.median(ts_kept)), 3),
           "runs_ms_pk_coset_evals_resident": [round(t, 3) for t in ts_kept],
           "pk_coset_evals_resident": "second timing: ProvingKey.keep_coset_evals() holds the 13 selector + 5 sigma coset "
                                      "evaluations in HBM (%.2f GiB per rank), so a proof runs 7 coset NTTs instead of "
                                      "the reference's 25 (prover.rs:534-544 recomputes them per proof); same proof "
                                      "bytes.  'ms' recomputes all 25 like the reference" % (kept_bytes / 2**30),
           "ms_pk_coset_evals_resident_and_lagrange_commits": round(float(np.median(ts_both)), 3) if ts_both else None,
           "ms_lagrange_round1_commits": round(float(np.median(ts_lag)), 3) if ts_lag else None,
           "lagrange_round1_commits": "third timing (one GPU): ProvingKey.keep_lagrange_commit_key() -- the 5 wire commitments "
                                      "are MSMs over the witness values + blinders on [L_i(beta)]g (jfhip_srs_lagrange, %s s "
                                      "at set-up) instead of over the iFFT's coefficients; proof bytes equal: %s.  The "
                                      "gain is this circuit's (wires hold counters, ones, zeros); 'ms' commits coefficient "
                                      "vectors like the reference (prover.rs:77-91)"
                                      % ("%.2f" % lag_setup_s if lag_setup_s is not None else "-", lag_same),
           "ns_per_gate": round(float(np.median(ts)) * 1e6 / (1 << log_g), 2),
           "includes": "witness upload (32 B x num_vars), all device work, host Merlin transcript + RNG, read-back of "
                       "13 commitments and 10 evaluations",
           "setup_not_timed": {"circuit_build_s": round(build_s, 2), "srs_and_preprocess_s": round(setup_s, 2)},
           "proof_bytes": len(first), "first_proof_sha": hashlib.sha256(first).hexdigest()[:16],
           "parity": "proof bytes equal oracle/plonk_ref.py at small sizes and the 2^20 proof verifies "
                     "(tests/test_gpu_plonk.py)"}
    srs.powers_of_g.free()
    del pk
    torch.cuda.empty_cache()
    return sec


def real_batch_prove_section(L, torch, dist, world, rank, log_g=20):
    """PlonkKzgSnark::batch_prove (snark.rs:61-74) of TurboPlonk BN254 bench circuits of 2^log_g gates with the real
    transcript and RNG: N = 1 proves 2 instances on the one GPU; N > 1 proves N instances, one circuit per GPU
    (batch_prove_one_per_rank: commitments and evaluations gathered, aggregated commitments as group sums)."""
    import hashlib

    from jellyfish_amd import hostglue as H
    from jellyfish_amd import plonk as JP

    k_inst = 2 if world == 1 else world
    mine = range(k_inst) if world == 1 else [rank]
    css, wits = [], []
    for i in mine:
        cs = JP.PlonkCircuit.new_turbo_plonk(0)
        a = cs.create_variable(i + 2)
        for _ in range((1 << log_g) - 10):
            a = cs.add(a, cs.one())
        cs.finalize_for_arithmetization()
        css.append(cs)
        wits.append(H.fr_to_mont(cs.cp, cs.witness))
    n = css[0].eval_domain_size()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(0, n + 2, 0x6A660004)
    pks = [JP.PlonkKzgSnark.preprocess(srs, cs)[0] for cs in css]
    rng = H.test_rng()

    def run():
        if world == 1:
            return JP.PlonkKzgSnark.batch_prove(rng, css, pks, witness_monts=wits)
        return JP.PlonkKzgSnark.batch_prove_one_per_rank(rng, css[0], pks[0], witness_mont=wits[0])

    first = run().serialize_compressed()
    ts = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dt = max_over_ranks(torch, dist, dt)
        ts.append(dt * 1e3)
    ms = float(np.median(ts))
    sec = {"workload": "PlonkKzgSnark::batch_prove, %d TurboPlonk BN254 bench circuits of 2^%d gates -> one BatchProof, "
                       "StandardTranscript, blinding from test_rng" % (k_inst, log_g),
           "n_gpus": world, "instances": k_inst,
           "layout": "all instances on one GPU" if world == 1 else "one circuit per GPU; per round one all_gather of "
                     "commitments / evaluations, aggregated commitments are group sums of per-rank partial sums",
           "ms": round(ms, 3), "runs_ms": [round(t, 3) for t in ts], "ms_per_instance": round(ms / k_inst, 3),
           "proof_bytes": len(first), "first_proof_sha": hashlib.sha256(first).hexdigest()[:16],
           "parity": "BatchProof bytes equal oracle/plonk_ref.py batch_prove and pass its batch verifier at small sizes; "
                     "one-per-rank bytes equal the one-GPU batch_prove (tests/test_gpu_plonk.py, tests/test_gpu_dist_prove.py)"}
    srs.powers_of_g.free()
    del pks
    torch.cuda.empty_cache()
    return sec


BLS_G = np.array([0x5cb38790fd530c16, 0x7817fc679976fff5, 0x154f95c7143ba1c1, 0xf0ae6acdf3d0e747, 0xedce6ecc21dbf440,
                  0x120177419e0bfb75, 0xbaac93d50ce72271, 0x8c22631a7918fd8e, 0xdd595f13570725ce, 0x51ac582950405194,
                  0x0e1c8c3fad0059c0, 0x0bbc3efc5008a26a], dtype=np.uint64)  # BLS12-381 G1 generator, Montgomery


def batch_prove_section(L, torch, dist, world, rank, stream, log_g):
    """BASELINE config 5: PlonkKzgSnark::batch_prove of `world` independent UltraPlonk (Plookup) circuits of 2^log_g gates
    on BLS12-381, one circuit per GPU, the REAL call: gen_circuit_for_bench(UltraPlonk, range_bit_len = 8)
    (plonk/benches/bench.rs:29-46), StandardTranscript, blinding from test_rng, sorted-vector merge, 18 commitments per
    instance + 8 aggregated ones.  N = 1 proves one circuit (prove = the one-instance batch); N > 1 runs
    batch_prove_one_per_rank: per-instance commitments / evaluations gathered, aggregated commitments as group sums."""
    import hashlib

    from jellyfish_amd import hostglue as H
    from jellyfish_amd import plonk as JP

    t0 = time.perf_counter()
    cs = JP.PlonkCircuit.new_ultra_plonk(1, 8)
    a = cs.create_variable(rank + 2) if world > 1 else cs.zero()
    one = cs.one()
    for _ in range((1 << log_g) - 10):
        a = cs.add(a, one)
    cs.finalize_for_arithmetization()
    n = cs.eval_domain_size()
    wit = H.fr_to_mont(cs.cp, cs.witness)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    srs = JP.PlonkKzgSnark.universal_setup_for_testing(1, n + 2, 0x6A660005)
    pk, vk = JP.PlonkKzgSnark.preprocess(srs, cs)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    rng = H.test_rng()

    def run():
        if world == 1:
            return JP.PlonkKzgSnark.batch_prove(rng, [cs], [pk], witness_monts=[wit])
        return JP.PlonkKzgSnark.batch_prove_one_per_rank(rng, cs, pk, witness_mont=wit)

    first = run().serialize_compressed()
    ts = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dt = max_over_ranks(torch, dist, dt)
        ts.append(dt * 1e3)
    ms = float(np.median(ts))
    JP.PROFILE = {}
    run()
    rounds = dict(JP.PROFILE)
    JP.PROFILE = None
    # third timing (one GPU): round-1 / round-1.5 commitments from evaluations on the Lagrange-basis commit key
    ts_lag, lag_same, lag_setup_s, lag_rounds = None, None, None, None
    if world == 1:
        t0 = time.perf_counter()
        pk.keep_lagrange_commit_key()
        torch.cuda.synchronize()
        lag_setup_s = time.perf_counter() - t0
        rng = H.test_rng()
        lag_same = run().serialize_compressed() == first
        ts_lag = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts_lag.append((time.perf_counter() - t0) * 1e3)
        JP.PROFILE = {}
        run()
        lag_rounds = {k: v for k, v in JP.PROFILE.items() if k in ("r1_commit", "r1p_commit")}
        JP.PROFILE = None
        pk.drop_lagrange_commit_key()
    # second timing: the 24 proving-key polynomials' coset evaluations resident in HBM (same proof bytes)
    pk.keep_coset_evals()
    kept_bytes = pk.coset_evals_bytes()
    rng = H.test_rng()
    same = run().serialize_compressed() == first
    ts_res = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dt = max_over_ranks(torch, dist, dt)
        ts_res.append(dt * 1e3)
    ts_both = None
    if world == 1:      # both options together
        pk.keep_lagrange_commit_key()
        run()
        ts_both = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts_both.append((time.perf_counter() - t0) * 1e3)
        pk.drop_lagrange_commit_key()
    sec = {"workload": "PlonkKzgSnark::batch_prove, %d UltraPlonk (Plookup, range_bit_len 8) BLS12-381 bench circuits of "
                       "2^%d gates, one per GPU -> one BatchProof; StandardTranscript, blinding from test_rng; per circuit "
                       "11 iNTT(n) + 35 coset NTT(8n) + sorted-vector merge + 2 grand products + Plookup quotient + coset "
                       "iNTT(8n) + 18 commitments + 27 evaluations" % (world, log_g),
           "n_gpus": world, "instances": world, "domain_size": n,
           "ms": round(ms, 3), "runs_ms": [round(t, 3) for t in ts],
           "circuits_per_s": round(world / (ms * 1e-3), 3), "ns_per_gate": round(ms * 1e6 / (world << log_g), 2),
           "rounds_ms_rank0_with_syncs": rounds,
           "ms_pk_coset_evals_resident": round(float(np.median(ts_res)), 3),
           "runs_ms_pk_coset_evals_resident": [round(t, 3) for t in ts_res],
           "pk_coset_evals_resident": "second timing: ProvingKey.keep_coset_evals() holds the %d selector + %d sigma + 4 "
                                      "table polynomials' coset evaluations in HBM (%.2f GiB per rank), so a proof "
                                      "transforms only its own polynomials instead of the reference's 35 (prover.rs:534-567 "
                                      "recomputes them per proof); proof bytes equal to the first timing's: %s.  'ms' "
                                      "recomputes all 35 like the reference"
                                      % (pk.selectors.shape[0], pk.sigmas.shape[0], kept_bytes / 2**30, same),
           "ms_pk_coset_evals_resident_and_lagrange_commits": round(float(np.median(ts_both)), 3) if ts_both else None,
           "ms_lagrange_round1_commits": round(float(np.median(ts_lag)), 3) if ts_lag else None,
           "lagrange_round1_commits": "third timing (one GPU): ProvingKey.keep_lagrange_commit_key() -- the 6 wire and 2 sorted-"
                                      "lookup commitments are MSMs over evaluations + blinders on [L_i(beta)]g "
                                      "(jfhip_srs_lagrange, %s s at set-up) instead of over coefficients; commit rounds then "
                                      "%s ms; proof bytes equal to the first timing's: %s.  The gain is this circuit's "
                                      "(small witness values); 'ms' commits coefficient vectors like the reference"
                                      % ("%.2f" % lag_setup_s if lag_setup_s is not None else "-", lag_rounds, lag_same),
           "includes": "witness upload, all device work, host Merlin transcript + RNG, commitment / evaluation read-backs",
           "setup_not_timed": {"circuit_build_s": round(build_s, 2), "srs_and_preprocess_s": round(setup_s, 2)},
           "proof_bytes": len(first), "first_proof_sha": hashlib.sha256(first).hexdigest()[:16],
           "parity": "BatchProof bytes equal oracle/plonk_ref.py and verify at small sizes, the 2^22-gate proof verifies; "
                     "one-per-rank bytes equal the one-GPU batch_prove (tests/test_gpu_ultraplonk.py, "
                     "tests/test_gpu_dist_prove.py)"}
    srs.powers_of_g.free()
    del pk
    torch.cuda.empty_cache()
    return sec


def launch_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, RCCL) through
    torch.distributed.run and return its exit status.  Nothing here touches the GPU (no exec after GPU init)."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()          # no HIP work is started in this process: the ranks are child processes
    rehearse = os.environ.get("JF_BENCH_REHEARSE") == "1"
    if have < n_gpus and not rehearse:
        print("bench.py: --gpus %d asked for but only %d GPU(s) are visible; refusing to print a line for fewer GPUs "
              "(JF_BENCH_REHEARSE=1 rehearses the N > 1 code path with all ranks on cuda:0 over gloo)" % (n_gpus, have),
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="points per GPU = 2^log_n (metric: 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-prove", action="store_true")
    ap.add_argument("--ultra-log-n", type=int, default=22,
                    help="also run BASELINE config 5: batch_prove of UltraPlonk BLS12-381 circuits of 2^K gates, "
                         "one circuit per GPU (config 5 is K = 22; 0 skips it)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed 2^log_n MSM steps and the roofline of their dominant kernel: under "
                         "rocprofv3 --kernel-trace --stats every k_msm_accumulate launch is then one of the metric's size")
    args = ap.parse_args()
    if args.headline_only:
        args.no_ntt = args.no_prove = args.no_cpu_baseline = True
        args.ultra_log_n = 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ.get("WORLD_SIZE")),
              file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    from jellyfish_amd import dist as jdist
    from jellyfish_amd import lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # JF_BENCH_REHEARSE=1: all ranks share cuda:0 and rendezvous over gloo (a one-GPU box rehearsing the N > 1 code path;
    # RCCL refuses two ranks on one device).  Never set by the driver.
    rehearse = world > 1 and os.environ.get("JF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    L.load_library()
    L.init(local_rank if world > 1 else 0)

    n_total = 1 << args.log_n
    # The metric's MSM has 2^log_n points IN TOTAL.  N > 1: the point range is partitioned over the ranks (north_star: "MSM
    # point ranges are partitioned per GPU"), rank k holds the SRS shard [beta^i] g, i in shard_range(2^log_n, k, N), and
    # the matching scalars: strong scaling, and the same group element as at N = 1 (result_sha must not depend on N).
    lo, hi = jdist.shard_range(n_total, rank, world)
    n = hi - lo
    beta = uniform_fr(1, 0x6A660001, BN254_R)[0]

    def shifted_generator(e: int):
        """[beta^e] g: the exponent folded with the device field / group operations"""
        if e == 0:
            return BN254_G
        acc, base = None, beta.copy()
        while e:
            if e & 1:
                acc = base.copy() if acc is None else L.field_op(0, "mul", acc, base)
            base = L.field_op(0, "mul", base, base)
            e >>= 1
        return L.g1_mul(0, BN254_G, acc)

    t0 = time.time()
    srs = L.Srs.gen_for_testing(0, beta, shifted_generator(lo), n)
    srs_build_s = time.time() - t0
    # every rank draws the same 2^log_n scalars and keeps its range (N = 1: all of them)
    scalars_all = uniform_fr(n_total, 0xC0FFEE, BN254_R)
    scalars_host = np.ascontiguousarray(scalars_all[lo:hi])
    del scalars_all
    scalars = torch.from_numpy(scalars_host.view(np.int64)).cuda()
    stream = L.Stream(wrap_raw=torch.cuda.current_stream().cuda_stream)
    engine = jdist.HipMsmEngine(L, srs, stream)

    def timed_steps(step, warmup, steps):
        res = None
        for _ in range(warmup):
            res = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dt = max_over_ranks(torch, dist, dt)
        return dt, res

    elapsed, res = timed_steps(lambda: jdist.sharded_msm(engine, scalars), args.warmup, args.steps)
    result_bytes = bytes(res.cpu().numpy().tobytes())

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed / 1e6
        out = {
            "metric": "MSM throughput, BN254 G1, 2^%d points (UnivariateKzgPCS::commit arithmetic)%s"
                      % (args.log_n, "" if world == 1 else ", point range partitioned over %d GPUs" % world),
            "value": round(value, 3), "unit": "Mpts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,   # the total is 2^log_n points at every N
            "dtype": "u32 limbs (256-bit Montgomery, INT32 VALU)", "data": "synthetic",
            "config": {"workload": "msm_bn254_g1_2^%d" % args.log_n, "curve": "bn254",
                       "points_per_gpu": n, "total_points": n_total, "window_bits": srs.window_bits,
                       "srs": "[beta^i]g generated on device (%.2f s incl. window-multiple table)" % srs_build_s,
                       "scalars": "uniform in [0,r), Montgomery form, resident in HBM",
                       "sharding": "point range [k 2^%d / N, (k + 1) 2^%d / N) on rank k + all_gather of one XYZZ element "
                                   "per GPU; same group element as at N = 1" % (args.log_n, args.log_n)
                                   if world > 1 else "single GPU"},
            "result_sha": __import__("hashlib").sha256(result_bytes).hexdigest()[:16],
        }

    if world > 1 and not args.headline_only:
        # the round-1/2 figure, kept as an extra key: 2^log_n points PER rank (weak scaling, N * 2^log_n points in total)
        srs_w = L.Srs.gen_for_testing(0, beta, shifted_generator(rank * n_total), n_total)
        sc_w = torch.from_numpy(uniform_fr(n_total, 0xC0FFEE + rank, BN254_R).view(np.int64)).cuda()
        eng_w = jdist.HipMsmEngine(L, srs_w, stream)
        k_w = max(2, min(args.steps, 5))
        el_w, _ = timed_steps(lambda: jdist.sharded_msm(eng_w, sc_w), 1, k_w)
        if rank == 0:
            out["weak_scaling"] = {"points_per_gpu": n_total, "total_points": world * n_total, "steps": k_w,
                                   "ms_per_step": round(el_w / k_w * 1e3, 4),
                                   "value": round(world * n_total * k_w / el_w / 1e6, 3), "unit": "Mpts/s",
                                   "window_bits": srs_w.window_bits}
        srs_w.free()
        del sc_w, eng_w
        torch.cuda.empty_cache()

    # ---- roofline of the dominant kernel + secondary measurements: rank 0, N = 1 only ----------
    if rank == 0 and world == 1:
        L.set_profiling(True)
        reps = max(3, min(args.steps, 10))
        acc_ms, phases_sum = 0.0, {}
        for _ in range(reps):
            L.msm_device(srs, scalars.data_ptr(), n, engine._out.data_ptr(), stream=stream)
            ph = L.last_phase_ms(stream, 0)
            for k, v in ph.items():
                phases_sum[k] = phases_sum.get(k, 0.0) + v
        L.set_profiling(False)
        phases = {k: round(v / reps, 4) for k, v in phases_sum.items()}
        acc_ms = phases.get("bucket_accumulate", float("nan"))
        c = srs.window_bits
        windows = (255 + c - 1) // c
        macs_per_point = windows * 10 * 136  # W windows x (8M+2S mixed add) x (2*8^2+8) MACs per Montgomery product
        macs = float(macs_per_point) * n
        mad_peak = L.probe(0)
        copy_bw = L.probe(1)
        achieved = macs / (acc_ms * 1e-3)
        alg_bytes = 96.0 * n  # 64 B affine point + 32 B scalar per point
        traffic = None
        traffic_stale = None
        src_sha = kernel_source_sha()
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("workload") in (out["config"]["workload"], out["config"]["workload"] + "_per_gpu"):
                    traffic = pmc.get("k_msm_accumulate_hbm_bytes")
                    if pmc.get("kernel_source_sha") != src_sha:
                        traffic_stale = "counters taken on kernel sources %s (head %s), this run is %s" % (
                            pmc.get("kernel_source_sha"), pmc.get("git_head"), src_sha)
            except (OSError, ValueError) as e:
                traffic_stale = "profiles/pmc_latest.json unreadable: %r" % (e,)
        out["roofline"] = {
            "kernel": "k_msm_accumulate", "bound": "valu_int32",
            "achieved": round(achieved / 1e12, 4), "peak": round(mad_peak / 1e12, 4), "unit": "TMAC/s (v_mad_u64_u32)",
            "frac": round(achieved / mad_peak, 4), "traffic": traffic,
            "traffic_source": "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes), bytes per "
                              "launch; a committed measurement, not taken in this run" + (
                                  "" if not traffic_stale else "; STALE: " + traffic_stale),
            "algorithmic": "%d windows x 10 field mul x 136 MAC = %d MAC/point x 2^%d points" % (windows, macs_per_point, args.log_n),
            "avg_launch_ms": acc_ms,
            "peak_source": "v_mad_u64_u32 issue-rate micro-benchmark on this device (jfhip_probe 0); not in MI355X_MICROARCH.md",
            "hbm_view": {"bound": "hbm", "achieved": round(alg_bytes / (acc_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(alg_bytes / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "measured_copy_GBs": round(copy_bw / 1e9, 1)},
        }
        # VALU issue evidence from the rocprofv3 --pmc passes of the same workload (scripts/pmc_valu.sh, committed summary):
        # wave-instructions issued per shader cycle over the chip (1024 SIMDs; a wave64 v_mad_u64_u32 occupies its SIMD for
        # 4 cycles, 32-bit adds / shifts / moves for 2)
        pv_all = None
        try:
            pv_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu_latest.json")))
        except (OSError, ValueError) as e:
            out["roofline"]["valu_pmc"] = {"error": "profiles/pmc_valu_latest.json unreadable: %r" % (e,)}
        if pv_all is not None:
            pv = pv_all["kernels"]
            pv_current = pv_all.get("kernel_source_sha") == src_sha
            kk = [k for k in pv if "k_msm_accumulate" in k and "Bn254" in k]
            if kk:
                d = pv[kk[0]]
                cyc = d["GRBM_GUI_ACTIVE"] / 8.0
                out["roofline"]["valu_pmc"] = {
                    "kernel": kk[0], "source": "profiles/pmc_valu_latest.json (rocprofv3 --pmc, 2^24 points): a committed "
                                               "measurement, not taken in this run",
                    "taken_on": {"git_head": pv_all.get("git_head"), "kernel_source_sha": pv_all.get("kernel_source_sha")},
                    "stale": (not pv_current),
                    "executed_macs_per_launch": (d.get("SQ_INSTS_VALU_INT64", 0.0) * 64.0),
                    "SQ_INSTS_VALU_per_launch": d["SQ_INSTS_VALU"], "shader_cycles_per_launch": round(cyc),
                    "valu_wave_insts_per_cycle_chip": round(d["SQ_INSTS_VALU"] / cyc, 1),
                    "valu_insts_per_mixed_add": round(d["SQ_INSTS_VALU"] * 64.0 / (windows * float(1 << 24)), 1),
                    "int64_class_insts_per_launch": d.get("SQ_INSTS_VALU_INT64"),
                    "mad_issue_cycles_frac": round(d.get("SQ_INSTS_VALU_INT64", 0.0) * 4.0 / (cyc * 1024.0), 3),
                    "wait_inst_any_frac_of_wave_cycles": d.get("frac_of_wave_cycles:SQ_WAIT_INST_ANY"),
                    "simd_busy_frac_range": [
                        round(((d.get("SQ_INSTS_VALU_INT64", 0.0) + d.get("SQ_INSTS_VALU_INT32", 0.0)) * 4.0 + (
                            d["SQ_INSTS_VALU"] - d.get("SQ_INSTS_VALU_INT64", 0.0) - d.get("SQ_INSTS_VALU_INT32", 0.0)) * 2.0) / (
                                cyc * 1024.0), 3),
                        round(d["SQ_INSTS_VALU"] * 4.0 / (cyc * 1024.0), 3)],
                    "simd_busy_note": "share of all SIMD cycles (1024 SIMDs x shader cycles) that the issued VALU instructions "
                                      "occupy: a wave64 v_mad_u64_u32 / 64-bit shift / 32-bit multiply / carry add holds its "
                                      "SIMD for 4+ cycles, v_add_u32 / v_and / v_mov / 32-bit shifts for 2 "
                                      "(scripts/ubench/valu_rate.hip, profiles/r03_valu_rate.txt); lower end: only the "
                                      "INT64 + INT32 counter classes at 4, upper end: every instruction at 4.  "
                                      "executed_macs counts the 64-bit-class instructions the kernel issues (29-bit limbs: 171 "
                                      "v_mad_u64_u32 + 18 64-bit shifts per product) against the 136-per-product convention "
                                      "of `algorithmic`",
                }
        # second wall of this kernel: every mixed addition gathers one random 64-byte table point.  jfhip_probe(4, GiB)
        # measures the device's random 64-byte gather rate over a table of the same size, nothing else running.
        table_gib = max(1, int(round(windows * n * 64 / 2**30)))
        try:
            gather_peak = L.probe(4, table_gib)
            gathers = float(windows) * n
            out["roofline"]["gather_view"] = {
                "bound": "random 64-byte gathers", "table_GiB": table_gib,
                "achieved": round(gathers / (acc_ms * 1e-3) / 1e9, 3), "peak": round(gather_peak / 1e9, 3),
                "unit": "G gathers/s", "frac": round(gathers / (acc_ms * 1e-3) / gather_peak, 4),
                "peak_source": "gather-only micro-benchmark on this device over a %d GiB table (5.5e10/s at 1 GiB, "
                               "1.35e10/s at 12 GiB measured: per-XCD translation reach, not HBM bandwidth)" % table_gib,
                "note": "a rate that coincides with the kernel's, not a bound on it: with every gather redirected into one "
                        "1 GiB window the same launch takes 14.63 instead of 14.80-14.85 ms "
                        "(profiles/r03_slabtest_accumulate.log, DESIGN.md 3.2); the kernel is VALU-issue-bound"}
        except Exception as e:  # reporting only
            out["roofline"]["gather_view"] = {"error": repr(e)}
        out["phases_ms"] = phases

        if not args.no_ntt:
            log_m, batch = 22, 8
            m = 1 << log_m
            d = torch.from_numpy(uniform_fr(batch * m, 5, BLS_R).view(np.int64)).cuda()
            h = L.field_op(2, "to_mont", np.array([7, 0, 0, 0], dtype=np.uint64))
            L.set_profiling(True)
            res_ntt = {}
            for inverse in (False, True):
                L.ntt_device(1, d.data_ptr(), log_m, batch, inverse, h, stream=stream)
                stream.sync()
                ts = []
                for _ in range(5):
                    L.ntt_device(1, d.data_ptr(), log_m, batch, inverse, h, stream=stream)
                    ts.append(sum(L.last_phase_ms(stream, 1).values()))
                ms = float(np.median(ts))
                gbs = batch * m * 64 / (ms * 1e-3) / 1e9
                res_ntt["inverse" if inverse else "forward"] = {
                    "ms": round(ms, 4), "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                                     "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}}
            L.set_profiling(False)
            try:   # HBM bytes per transform from the rocprofv3 --pmc passes (profiles/pmc_latest.json), if collected
                # the transforms read with 16-byte-per-lane streaming loads, for which gfx950's FETCH_SIZE counts half the
                # bytes (MI355X_MICROARCH.md, HBM section): the corrected figure is 2 x FETCH_SIZE + WRITE_SIZE
                ntt_traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get(
                    "ntt_8x2^22_bls_hbm_bytes_per_transform_upper")
            except Exception:
                ntt_traffic = None
            # VALU view: the transform is bound by INT32 multiply-adds long before HBM.  Algorithmic products per element
            # = (log2 n) / 2 butterfly twiddles + 1 coset / n^-1 scaling, 136 MAC each (the convention of the MSM roofline).
            ntt_prods = log_m / 2.0 + 1.0
            ntt_insts = None
            if pv_all is not None:
                try:
                    per_pass = [v["SQ_INSTS_VALU"] * v.get("launches", 1) for k, v in pv_all["kernels"].items()
                                if "k_ntt_pass_r8" in k and "Bls381Fr" in k]
                    # scripts/pmc_run.py runs 4 transforms of 8 x 2^22 elements
                    ntt_insts = {"SQ_INSTS_VALU_per_element_and_transform": round(sum(per_pass) * 64.0 / (4.0 * batch * m), 1),
                                 "source": "profiles/pmc_valu_latest.json (committed measurement)",
                                 "stale": pv_all.get("kernel_source_sha") != src_sha}
                except (KeyError, TypeError):
                    ntt_insts = None
            for k in res_ntt:
                t_s = res_ntt[k]["ms"] * 1e-3
                a = ntt_prods * 136.0 * batch * m / t_s
                res_ntt[k]["roofline"]["valu_view"] = {
                    "bound": "valu_int32", "achieved": round(a / 1e12, 4), "peak": round(mad_peak / 1e12, 4),
                    "unit": "TMAC/s (v_mad_u64_u32)", "frac": round(a / mad_peak, 4),
                    "algorithmic": "%.0f products per element (log2(n)/2 twiddles + 1 scaling) x 136 MAC" % ntt_prods,
                    "valu_pmc": ntt_insts}
            for k in res_ntt:
                res_ntt[k]["roofline"]["traffic"] = ntt_traffic
                res_ntt[k]["roofline"]["traffic_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                                                          "scripts/pmc_collect.sh), per transform = 3 digit passes: "
                                                          "3.3 x the algorithmic 2.147e9 B")
            out["ntt"] = {"workload": "8 x 2^22 BLS12-381 Fr coset NTT (h = 7), device resident",
                          "algorithmic_bytes": "64 B per element per transform", **res_ntt}

        # BASELINE config 2: one commit of degree 2^20 (2^20 + 1 points, pcs/benches/kzg_gpu.rs:30-34), SRS resident;
        # device-resident scalars (kernels only) and host scalars (scalar upload + kernels + one point back, the
        # timed region of kzg_gpu.rs:58-83)
        try:
            if args.headline_only:
                raise RuntimeError("skipped (--headline-only)")
            n2 = (1 << 20) + 1
            srs2 = L.Srs.gen_for_testing(L.BN254, beta, BN254_G, n2)
            # its own 2^20 + 1 scalars when the headline ran on fewer (--log-n below 21): the device call reads n2 of them
            sc2_host = (np.ascontiguousarray(scalars_host[:n2]) if scalars_host.shape[0] >= n2
                        else uniform_fr(n2, 0xC0FFEE + 2, BN254_R))
            assert sc2_host.shape == (n2, 4)
            sc2 = torch.from_numpy(sc2_host.view(np.int64)).cuda()
            o2 = torch.zeros(128, dtype=torch.uint8, device="cuda")
            t_dev, t_host = [], []
            for i in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                L.msm_device(srs2, sc2.data_ptr(), n2, o2.data_ptr(), stream=stream)
                stream.sync()
                t_dev.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter()
                L.msm(srs2, sc2_host, stream=stream)
                t_host.append((time.perf_counter() - t0) * 1e3)
            md, mh = float(np.median(t_dev[2:])), float(np.median(t_host[2:]))
            out["msm_2^20"] = {"workload": "single G1 MSM, 2^20 + 1 points, BN254 (BASELINE config 2), window_bits %d" % srs2.window_bits,
                               "ms_device_resident": round(md, 3), "Mpts_s_device_resident": round(n2 / md / 1e3, 1),
                               "ms_host_scalars": round(mh, 3), "Mpts_s_host_scalars": round(n2 / mh / 1e3, 1),
                               "host_scalars": "32 MiB pageable host scalars uploaded per call + one affine point read back"}
            srs2.free()
            del sc2
        except Exception as e:  # secondary measurement
            out["msm_2^20"] = {"error": repr(e)}

        if not args.no_prove:
            out["plonk_prove"] = real_prove_section(L, torch, dist, 1)
            out["plonk_prove_hot_path"] = prove_section(L, torch, dist, 1, stream, beta)

        if not args.no_cpu_baseline:
            try:
                from oracle import coracle as C

                log_s = min(args.log_n, 24)
                ns = 1 << log_s
                assert scalars_host.shape[0] >= ns
                bases = srs.read(0, ns)
                threads = C.nthreads()
                t0 = time.perf_counter()
                cpu_res = C.msm(0, bases, scalars_host[:ns], threads)
                dt = time.perf_counter() - t0
                gpu_res, _ = L.msm(srs, scalars_host[:ns])
                out["cpu_baseline"] = {
                    "value": round(ns / dt / 1e6, 4), "unit": "Mpts/s", "cores": threads, "kind": "port",
                    "sample": "one MSM over the first 2^%d points/scalars of the same SRS and scalar set (%.1f s)" % (log_s, dt),
                    "matches_gpu": bool((cpu_res == gpu_res).all()),
                    "note": "CPU restatement (oracle/jf_oracle.c, OpenMP Pippenger), not arkworks: no Rust toolchain on the box"}
                if not args.no_prove:
                    # the prove path's two kernels on the same host cores: one 2^20-point MSM and one 2^23 coset NTT
                    # (BN254), extrapolated to config 4's 13 commitments + 26 size-8n + 7 size-n transforms
                    n20 = min(1 << 20, ns)
                    t0 = time.perf_counter()
                    C.msm(0, bases[:n20], scalars_host[:n20], threads)
                    t_msm = time.perf_counter() - t0
                    poly = np.zeros((1 << 23, 4), dtype=np.uint64)
                    poly[:n20] = scalars_host[:n20]
                    gen5 = L.field_op(0, "to_mont", np.array([5, 0, 0, 0], dtype=np.uint64))
                    t0 = time.perf_counter()
                    C.ntt(0, poly, 23, False, gen5, threads)
                    t_ntt = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    C.ntt(0, scalars_host[:n20].copy(), 20, True, None, threads)
                    t_intt = time.perf_counter() - t0
                    out["cpu_baseline"]["prove_path"] = {
                        "msm_2^20_ms": round(t_msm * 1e3, 1), "coset_ntt_2^23_ms": round(t_ntt * 1e3, 1),
                        "intt_2^20_ms": round(t_intt * 1e3, 1),
                        "estimate_ms": round((13 * t_msm + 26 * t_ntt + 7 * t_intt) * 1e3, 1),
                        "what": "13 x MSM(2^20) + 26 x NTT(2^23) + 7 x iNTT(2^20) of TurboPlonk BN254 2^20 gates on the "
                                "same %d host threads, each kernel timed once with the C port and multiplied out; the "
                                "quotient loop, grand product and polynomial glue are not in it" % threads}
            except Exception as e:  # the baseline is reporting only; never fail the bench on it
                out["cpu_baseline"] = {"value": None, "unit": "Mpts/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    # The secondary sections below run collectives at N > 1: if they stall (a rank failing inside a collective leaves
    # the others waiting), a watchdog still prints the headline line measured above, marks what is missing and ends
    # the process, so the contract's ONE JSON line is never lost to a secondary measurement.
    import copy
    import threading

    finished = threading.Event()
    snapshot = {"out": copy.deepcopy(out), "section": "start"}

    def watchdog():
        if finished.wait(float(os.environ.get("JF_BENCH_SECONDARY_TIMEOUT_S", "600"))):
            return
        # a stalled collective is a failure: say where, keep what was measured, exit non-zero
        msg = "bench.py: rank %d stalled in section '%s' (a rank failed or hung inside a collective)" % (rank, snapshot["section"])
        print(msg, file=sys.stderr, flush=True)
        if rank == 0 and snapshot["out"] is not None:
            line = snapshot["out"]
            line["secondary_sections"] = "STALLED in '%s'; sections not listed here did not finish" % snapshot["section"]
            print(json.dumps(line), flush=True)
        os._exit(3)

    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()

    def keep(key, sec):
        if rank == 0:
            out[key] = sec
            snapshot["out"] = copy.deepcopy(out)

    def enter(section):
        snapshot["section"] = section

    if world > 1 and not args.no_prove:
        # BASELINE config 4 at N GPUs: every commitment's point range is partitioned over the ranks
        enter("plonk_prove")
        try:
            sec = real_prove_section(L, torch, dist, world)
        except Exception as e:  # never lose the headline line to the secondary measurement
            sec = {"error": repr(e)}
        keep("plonk_prove", sec)
        enter("plonk_prove_hot_path")
        try:
            sec = prove_section(L, torch, dist, world, stream, beta)
        except Exception as e:
            sec = {"error": repr(e)}
        keep("plonk_prove_hot_path", sec)

    if not args.no_prove:
        enter("batch_prove")
        try:
            sec = real_batch_prove_section(L, torch, dist, world, rank)
        except Exception as e:
            sec = {"error": repr(e)}
        keep("batch_prove", sec)

    if args.ultra_log_n:
        enter("batch_prove_ultra")
        try:
            sec = batch_prove_section(L, torch, dist, world, rank, stream, args.ultra_log_n)
        except Exception as e:
            sec = {"error": repr(e)}
        keep("batch_prove_ultra", sec)

    finished.set()
    if rank == 0:
        print(json.dumps(out), flush=True)
    srs.free()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # a launcher (torch.distributed.run) reports only the exit status of a failed rank: say why, on stderr
        import traceback

        print("bench.py: rank %s failed:\n%s" % (os.environ.get("RANK", "0"), traceback.format_exc()), file=sys.stderr,
              flush=True)
        raise
