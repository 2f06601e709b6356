"""TurboPlonk `PlonkKzgSnark::{preprocess, prove}` with every polynomial operation on the device.

Host-side mirror of the reference interface (same names, argument meaning and error behaviour):
  * `PlonkCircuit`                       relation/src/constraint_system.rs:347-470, 558-790, 831-884, 943-981, 1116-1187
  * `PlonkKzgSnark.universal_setup_for_testing / preprocess / prove`
                                          plonk/src/proof_system/snark.rs:447-488, 492-580, 587-613 (-> 163-430)
  * prover rounds                         plonk/src/proof_system/prover.rs:77-91, 128-142, 183-229, 299-413, 457-488,
                                          491-661, 894-1015
The host keeps what the reference keeps on one thread: the Merlin transcript (`StandardTranscript`), the RNG draws for
the blinding factors (same order), a handful of scalar formulas.  Everything that touches a polynomial runs in libjfhip on
the GPU and stays resident in HBM between rounds: witness gather, iNTTs, blinding, the permutation grand product, 25
coset NTTs, the quotient evaluation, the coset iNTT, quotient splitting, 13 commitments (Pippenger MSM), evaluations,
linearisation and the two opening quotients.  Per proof 13 group elements and 10 field elements cross PCIe, plus the
witness upload.  There is no CPU path: without libjfhip / a GPU every call raises.

Checked against the big-integer restatement of prover and verifier (byte-identical proofs from the same RNG stream;
proofs verify) in tests/test_gpu_plonk.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

from . import hostglue as H
from . import lib as L
from .transcript import MerlinTranscript

GATE_WIDTH = 4
N_TURBO_PLONK_SELECTORS = 13


class PlonkError(RuntimeError):
    pass


PROFILE = None   # set to a dict to collect a per-round wall-clock breakdown of prove() (adds device syncs)


def _sel(q_lc=(0, 0, 0, 0), q_mul=(0, 0), q_hash=(0, 0, 0, 0), q_o=0, q_c=0, q_ecc=0):
    return tuple(q_lc) + tuple(q_mul) + tuple(q_hash) + (q_o, q_c, q_ecc)


class PlonkCircuit:
    """TurboPlonk constraint system (host side).  Variables 0 / 1 hold the constants 0 / 1."""

    def __init__(self, curve: int):
        self.cp = H.CURVES[curve]
        self.r = self.cp.r
        self.num_wire_types = GATE_WIDTH + 1
        self.witness: list[int] = [0, 1]
        self._gates: list[tuple] = []
        self._wires: list[list[int]] = [[] for _ in range(GATE_WIDTH + 1)]
        self.pub_input_gate_ids: list[int] = []
        self._n = 0
        self.enforce_constant(0, 0)
        self.enforce_constant(1, 1)

    @classmethod
    def new_turbo_plonk(cls, curve: int) -> "PlonkCircuit":
        return cls(curve)

    # ---- builder ---------------------------------------------------------------------------------
    def _check_var(self, *vs):
        for v in vs:
            if not 0 <= v < len(self.witness):
                raise PlonkError("VarIndexOutOfBound(%d, %d)" % (v, len(self.witness)))

    def _insert_gate(self, wire_vars, sel):
        if self._n:
            raise PlonkError("ModifyFinalizedCircuit")
        for w, v in zip(self._wires, wire_vars):
            w.append(v)
        self._gates.append(sel)

    def num_gates(self) -> int:
        return len(self._gates)

    def num_vars(self) -> int:
        return len(self.witness)

    def num_inputs(self) -> int:
        return len(self.pub_input_gate_ids)

    def zero(self) -> int:
        return 0

    def one(self) -> int:
        return 1

    def create_variable(self, val: int) -> int:
        if self._n:
            raise PlonkError("ModifyFinalizedCircuit")
        self.witness.append(val % self.r)
        return len(self.witness) - 1

    def create_constant_variable(self, val: int) -> int:
        v = self.create_variable(val)
        self.enforce_constant(v, val)
        return v

    def create_public_variable(self, val: int) -> int:
        v = self.create_variable(val)
        self.set_variable_public(v)
        return v

    def set_variable_public(self, var: int):
        self._check_var(var)
        self.pub_input_gate_ids.append(self.num_gates())
        self._insert_gate((0, 0, 0, 0, var), _sel(q_o=1))

    def enforce_constant(self, var: int, constant: int):
        self._check_var(var)
        self._insert_gate((0, 0, 0, 0, var), _sel(q_c=constant % self.r, q_o=1))

    def add_gate(self, a: int, b: int, c: int):
        self._check_var(a, b, c)
        self._insert_gate((a, b, 0, 0, c), _sel(q_lc=(1, 1, 0, 0), q_o=1))

    def add(self, a: int, b: int) -> int:
        self._check_var(a, b)
        c = self.create_variable(self.witness[a] + self.witness[b])
        self.add_gate(a, b, c)
        return c

    def sub_gate(self, a: int, b: int, c: int):
        self._check_var(a, b, c)
        self._insert_gate((a, b, 0, 0, c), _sel(q_lc=(1, self.r - 1, 0, 0), q_o=1))

    def sub(self, a: int, b: int) -> int:
        self._check_var(a, b)
        c = self.create_variable(self.witness[a] - self.witness[b])
        self.sub_gate(a, b, c)
        return c

    def mul_gate(self, a: int, b: int, c: int):
        self._check_var(a, b, c)
        self._insert_gate((a, b, 0, 0, c), _sel(q_mul=(1, 0), q_o=1))

    def mul(self, a: int, b: int) -> int:
        self._check_var(a, b)
        c = self.create_variable(self.witness[a] * self.witness[b])
        self.mul_gate(a, b, c)
        return c

    def enforce_bool(self, a: int):
        self._check_var(a)
        self._insert_gate((a, a, 0, 0, a), _sel(q_mul=(1, 0), q_o=1))

    def enforce_equal(self, a: int, b: int):
        self._check_var(a, b)
        self._insert_gate((a, b, 0, 0, 0), _sel(q_lc=(1, self.r - 1, 0, 0), q_o=1))

    def power5_gate(self, a: int) -> int:
        """c = a^5 through the q_hash selector (FifthRootGate shape)"""
        self._check_var(a)
        c = self.create_variable(pow(self.witness[a], 5, self.r))
        self._insert_gate((a, 0, 0, 0, c), _sel(q_hash=(1, 0, 0, 0), q_o=1))
        return c

    def pad_gates(self, n: int):
        for _ in range(n):
            self._insert_gate((0, 0, 0, 0, 0), _sel(q_lc=(1, self.r - 1, 0, 0), q_o=1))

    def public_input(self) -> list[int]:
        out = self._wires[GATE_WIDTH] if not self._n else self.wire_variables[GATE_WIDTH]
        return [self.witness[int(out[g])] for g in self.pub_input_gate_ids]

    # ---- finalisation ----------------------------------------------------------------------------
    def is_finalized(self) -> bool:
        return self._n != 0

    def eval_domain_size(self) -> int:
        if not self._n:
            raise PlonkError("UnfinalizedCircuit")
        return self._n

    def srs_size(self) -> int:
        return self.eval_domain_size() + 2

    def finalize_for_arithmetization(self):
        """pad to a power of two, io gates to the front, wire permutation (constraint_system.rs:1167-1187);
        the field-valued tables (k * g^j, sigma values, selector polynomials) are built on the device by preprocess"""
        if self._n:
            return
        n = 1
        while n < self.num_gates():
            n <<= 1
        if n.bit_length() - 1 > self.cp.two_adicity:
            raise PlonkError("DomainCreationError")
        W = self.num_wire_types
        pad = n - self.num_gates()
        wv = np.zeros((W, n), dtype=np.int64)
        for i in range(W):
            wv[i, :n - pad] = self._wires[i]
        gates = self._gates + [_sel()] * pad
        uniq: dict[tuple, int] = {}
        gate_kind = np.fromiter((uniq.setdefault(g, len(uniq)) for g in gates), dtype=np.int64, count=n)
        for gate_id in range(len(self.pub_input_gate_ids)):
            io = self.pub_input_gate_ids[gate_id]
            if io > gate_id:
                gate_kind[[gate_id, io]] = gate_kind[[io, gate_id]]
                wv[:, [gate_id, io]] = wv[:, [io, gate_id]]
                self.pub_input_gate_ids[gate_id] = gate_id
        # wire permutation: the wires of one variable, in (wire type, gate) order, form a cycle
        flat = wv.reshape(-1)
        order = np.argsort(flat, kind="stable")
        sv = flat[order]
        differs = sv[1:] != sv[:-1]
        is_start = np.concatenate(([True], differs))
        is_last = np.concatenate((differs, [True]))
        pos = np.arange(flat.size)
        group_start = np.maximum.accumulate(np.where(is_start, pos, 0))
        nxt = np.roll(order, -1)
        nxt[is_last] = order[group_start[is_last]]
        perm = np.empty_like(order)
        perm[order] = nxt
        self.wire_variables = wv
        self.wire_permutation = perm            # flat index wire_id * n + gate_id of the next wire in the cycle
        self.gate_kind = gate_kind
        self.gate_kinds = list(uniq.keys())
        self.k = H.compute_coset_representatives(self.cp, W, n)
        self._wires = self._gates = None
        self._n = n
        self.log_n = n.bit_length() - 1

    def check_circuit_satisfiability(self, pub_input) -> None:
        """gate equation on every row (constraint_system.rs:585-603, 894-938); host integers, for tests/small circuits"""
        if len(pub_input) != self.num_inputs():
            raise PlonkError("PubInputLenMismatch(%d, %d)" % (len(pub_input), self.num_inputs()))
        r = self.r
        pis = dict(zip(self.pub_input_gate_ids, pub_input))
        for gid in range(self._n):
            s = self.gate_kinds[int(self.gate_kind[gid])]
            w = [self.witness[int(self.wire_variables[i][gid])] for i in range(5)]
            lhs = (pis.get(gid, 0) + sum(s[i] * w[i] for i in range(4)) + s[4] * w[0] * w[1] + s[5] * w[2] * w[3]
                   + sum(s[6 + i] * pow(w[i], 5, r) for i in range(4)) + s[11] + s[12] * w[0] * w[1] * w[2] * w[3] * w[4])
            if (lhs - s[10] * w[4]) % r:
                raise PlonkError("GateCheckFailure(%d)" % gid)


# ---------------------------------------------------------------------------------------------------


@dataclass
class UniversalSrs:
    curve: int
    powers_of_g: "L.Srs"          # device resident, with its window-multiple table
    g: tuple
    h: tuple
    beta_h: tuple

    def max_degree(self) -> int:
        return len(self.powers_of_g) - 1


@dataclass
class VerifyingKey:
    domain_size: int
    num_inputs: int
    sigma_comms: list
    selector_comms: list
    k: list
    g: tuple
    h: tuple
    beta_h: tuple
    is_merged: bool = False

    def as_dict(self) -> dict:
        return dict(domain_size=self.domain_size, num_inputs=self.num_inputs, sigma_comms=self.sigma_comms,
                    selector_comms=self.selector_comms, k=self.k, g=self.g, h=self.h, beta_h=self.beta_h)


@dataclass
class ProvingKey:
    curve: int
    sigmas: torch.Tensor          # (W, n*4) coefficients
    selectors: torch.Tensor       # (13, n*4) coefficients
    commit_key: "L.Srs"
    vk: VerifyingKey
    # evaluation-form tables the grand product reads (the reference recomputes them from the circuit per proof)
    id_perm_vals: torch.Tensor = field(repr=False, default=None)
    sigma_vals: torch.Tensor = field(repr=False, default=None)
    wire_variables: torch.Tensor = field(repr=False, default=None)

    @property
    def wire_variables32(self) -> torch.Tensor:
        """the wire -> variable table as the uint32 indices jfhip_witness_gather_device takes (built once)"""
        if getattr(self, "_wv32", None) is None:
            self._wv32 = self.wire_variables.to(torch.int32).contiguous()
        return self._wv32
    # optional: the 13 selector + W sigma polynomials evaluated on the quotient coset (m = 8n points each), kept in HBM.
    # The reference recomputes these 18 coset FFTs in every proof (prover.rs:534-544); they depend on the proving key
    # only, and 288 GB of HBM holds them (18 x 256 MiB at n = 2^20).  Same values, so the same proof bytes.
    coset_evals: torch.Tensor = field(repr=False, default=None)
    coset_slice: tuple = (0, 1)   # (rank, world) the resident evaluations were computed for
    # selector polynomials that are identically zero for this circuit (a gate kind it never uses): their coset
    # evaluations are zero, so prove() fills those rows instead of transforming them (same values, same proof bytes)
    selector_is_zero: tuple = ()

    def domain_size(self) -> int:
        return self.vk.domain_size

    def quot_domain_size(self) -> int:
        n, W = self.vk.domain_size, self.sigmas.shape[0]
        m = 1
        while m < n * domain_size_ratio(n, W):
            m <<= 1
        return m

    def keep_coset_evals(self, stream=None, rank: int = 0, world: int = 1) -> "ProvingKey":
        """evaluate selectors and sigmas on the quotient coset once and keep them resident (see coset_evals).
        world > 1: only rank's slice of the coset (prove(sharded=True) with coset sharding, see quotient_slices)"""
        cp = H.CURVES[self.curve]
        n, m = self.vk.domain_size, self.quot_domain_size()
        if not coset_sharding_applies(n, m, world):
            rank, world = 0, 1
        ml = m // world
        nsel, W = self.selectors.shape[0], self.sigmas.shape[0]
        ev = torch.empty(nsel + W, ml * 4, dtype=torch.int64, device="cuda")
        ev[:nsel, :n * 4].copy_(self.selectors)
        ev[nsel:, :n * 4].copy_(self.sigmas)
        off = H.fr_to_mont(cp, [slice_offset(cp, m, rank)])[0]
        L.ntt_device(self.curve, ev.data_ptr(), ml.bit_length() - 1, nsel + W, inverse=False, coset_mont=off, in_len=n,
                     stream=_stream(stream))
        self.coset_evals = ev
        self.coset_slice = (rank, world)
        return self


@dataclass
class Proof:
    curve: int
    wires_poly_comms: list
    prod_perm_poly_comm: tuple
    split_quot_poly_comms: list
    opening_proof: tuple
    shifted_opening_proof: tuple
    wires_evals: list
    wire_sigma_evals: list
    perm_next_eval: int

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k in ("wires_poly_comms", "prod_perm_poly_comm", "split_quot_poly_comms",
                                              "opening_proof", "shifted_opening_proof", "wires_evals",
                                              "wire_sigma_evals", "perm_next_eval")}

    def serialize_compressed(self) -> bytes:
        """ark-serialize layout of `Proof` (plonk/src/proof_system/structs.rs:54-76); plookup_proof = None"""
        cp = H.CURVES[self.curve]
        out = bytearray()
        out += len(self.wires_poly_comms).to_bytes(8, "little")
        for c in self.wires_poly_comms:
            out += H.ser_g1(cp, c)
        out += H.ser_g1(cp, self.prod_perm_poly_comm)
        out += len(self.split_quot_poly_comms).to_bytes(8, "little")
        for c in self.split_quot_poly_comms:
            out += H.ser_g1(cp, c)
        out += H.ser_g1(cp, self.opening_proof) + H.ser_g1(cp, self.shifted_opening_proof)
        for v in (self.wires_evals, self.wire_sigma_evals):
            out += len(v).to_bytes(8, "little")
            for x in v:
                out += H.ser_fr(cp, x)
        out += H.ser_fr(cp, self.perm_next_eval)
        out.append(0)
        return bytes(out)


@dataclass
class ProofEvaluations:
    """plonk/src/proof_system/structs.rs:336-350"""
    wires_evals: list
    wire_sigma_evals: list
    perm_next_eval: int

    def as_dict(self) -> dict:
        return dict(wires_evals=self.wires_evals, wire_sigma_evals=self.wire_sigma_evals,
                    perm_next_eval=self.perm_next_eval)


@dataclass
class BatchProof:
    """plonk/src/proof_system/structs.rs:227-249 (TurboPlonk instances: plookup_proofs_vec = [None; k])"""
    curve: int
    wires_poly_comms_vec: list
    prod_perm_poly_comms_vec: list
    poly_evals_vec: list
    split_quot_poly_comms: list
    opening_proof: tuple
    shifted_opening_proof: tuple

    def __len__(self) -> int:
        return len(self.prod_perm_poly_comms_vec)

    def as_dict(self) -> dict:
        return dict(wires_poly_comms_vec=self.wires_poly_comms_vec, prod_perm_poly_comms_vec=self.prod_perm_poly_comms_vec,
                    poly_evals_vec=[e.as_dict() for e in self.poly_evals_vec],
                    split_quot_poly_comms=self.split_quot_poly_comms, opening_proof=self.opening_proof,
                    shifted_opening_proof=self.shifted_opening_proof)

    def to_proof(self) -> "Proof":
        """the single-instance `Proof` (inverse of `impl From<Proof> for BatchProof`, structs.rs:275-287)"""
        if len(self) != 1:
            raise PlonkError("a BatchProof over %d instances is not a Proof" % len(self))
        e = self.poly_evals_vec[0]
        return Proof(self.curve, self.wires_poly_comms_vec[0], self.prod_perm_poly_comms_vec[0],
                     self.split_quot_poly_comms, self.opening_proof, self.shifted_opening_proof, e.wires_evals,
                     e.wire_sigma_evals, e.perm_next_eval)

    def serialize_compressed(self) -> bytes:
        """ark-serialize layout of `BatchProof`"""
        cp = H.CURVES[self.curve]
        out = bytearray()

        def vec_g1(v):
            out.extend(len(v).to_bytes(8, "little"))
            for P in v:
                out.extend(H.ser_g1(cp, P))

        def vec_fr(v):
            out.extend(len(v).to_bytes(8, "little"))
            for x in v:
                out.extend(H.ser_fr(cp, x))

        k = len(self)
        out.extend(k.to_bytes(8, "little"))
        for comms in self.wires_poly_comms_vec:
            vec_g1(comms)
        vec_g1(self.prod_perm_poly_comms_vec)
        out.extend(k.to_bytes(8, "little"))
        for e in self.poly_evals_vec:
            vec_fr(e.wires_evals)
            vec_fr(e.wire_sigma_evals)
            out.extend(H.ser_fr(cp, e.perm_next_eval))
        out.extend(k.to_bytes(8, "little"))
        out.extend(bytes(k))
        vec_g1(self.split_quot_poly_comms)
        out.extend(H.ser_g1(cp, self.opening_proof))
        out.extend(H.ser_g1(cp, self.shifted_opening_proof))
        return bytes(out)


class StandardTranscript:
    """plonk/src/transcript/{mod.rs:46-186, standard.rs:14-39}"""

    def __init__(self, cp: H.CurveParams, label: bytes = b"PlonkProof"):
        self.cp = cp
        self.t = MerlinTranscript(label)

    def append_message(self, label: bytes, msg: bytes):
        self.t.append_message(label, msg)

    def append_vk_and_pub_input(self, vk: VerifyingKey, pub_input):
        cp = self.cp
        self.append_message(b"field size in bits", cp.r.bit_length().to_bytes(4, "little"))
        self.append_message(b"domain size", vk.domain_size.to_bytes(8, "little"))
        self.append_message(b"input size", vk.num_inputs.to_bytes(8, "little"))
        self.append_message(b"SRS G2 element", H.ser_g2(cp, vk.beta_h))
        self.append_field_elems(b"wire subsets separators", vk.k)
        self.append_commitments(b"selector commitments", vk.selector_comms)
        self.append_commitments(b"sigma commitments", vk.sigma_comms)
        self.append_field_elems(b"public input", pub_input)

    def append_commitments(self, label: bytes, comms):
        for c in comms:
            self.append_message(label, H.ser_g1(self.cp, c))

    def append_field_elems(self, label: bytes, xs):
        for x in xs:
            self.append_message(label, H.ser_fr(self.cp, x))

    def get_challenge(self, label: bytes) -> int:
        return int.from_bytes(self.t.challenge_bytes(label, 64), "little") % self.cp.r


# ---------------------------------------------------------------------------------------------------


def _dev_fr(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1)).cuda()


_STREAMS: dict = {}


def _stream(stream):
    """the caller's jfhip stream, else one wrapper (and one MSM/NTT workspace) per torch stream, kept for reuse"""
    if stream is not None:
        return stream
    raw = torch.cuda.current_stream().cuda_stream
    key = (torch.cuda.current_device(), raw)
    if key not in _STREAMS:
        _STREAMS[key] = L.Stream(wrap_raw=raw)
    return _STREAMS[key]


def _read_fr(cp, t: torch.Tensor) -> list[int]:
    return H.fr_from_mont(cp, t.cpu().numpy().view(np.uint64))


class _Committer:
    """batch_commit of device polynomials -> canonical affine points (None = identity).
    sharded: torch.distributed is initialised and the SRS is replicated on every rank; every commitment's coefficient
    range is partitioned over the ranks and ONE all_gather per commit round carries the partial sums
    (jellyfish_amd/dist.py sharded_commit_batch; BASELINE config 4 "MSM point ranges sharded across 8 MI355X")."""

    def __init__(self, srs: "L.Srs", cp, stream, sharded: bool = False, group=None):
        self.srs, self.cp, self.stream, self.sharded, self.group = srs, cp, stream, sharded, group
        self.point_bytes = 16 * cp.fq_limbs64
        if sharded:
            from . import dist as jdist

            self._jdist = jdist
            self._engine = jdist.HipMsmEngine(L, srs, stream)

    def __call__(self, polys, lens) -> list:
        cp, srs = self.cp, self.srs
        for ln in lens:
            if ln - 1 > len(srs) - 1:
                raise PlonkError("poly degree %d is larger than allowed %d" % (ln - 1, len(srs) - 1))
        if self.sharded:
            outs = self._jdist.sharded_commit_batch(self._engine, [p[:4 * ln] for p, ln in zip(polys, lens)],
                                                    group=self.group)
            rec = self.point_bytes + 4
            raw = torch.cat([o[:rec] for o in outs]).cpu().numpy().tobytes()
        else:
            rec = (self.point_bytes + 4 + 15) // 16 * 16
            out = torch.zeros(len(polys) * rec, dtype=torch.uint8, device="cuda")
            if len(polys) == 1:
                L.msm_device(srs, polys[0].data_ptr(), lens[0], out.data_ptr(), stream=self.stream)
            else:
                L.msm_batch_device(srs, [p.data_ptr() for p in polys], lens, out.data_ptr(), stream=self.stream)
            raw = out.cpu().numpy().tobytes()
        return [H.g1_from_record(cp, raw[i * rec:(i + 1) * rec]) for i in range(len(polys))]


def _commit_many(srs, cp, polys, lens, stream) -> list:
    return _Committer(srs, cp, stream)(polys, lens)


def domain_size_ratio(n: int, num_wire_types: int) -> int:
    """plonk/src/constants.rs:18-20"""
    return (num_wire_types * (n + 1) + 2) // n + 1


def coset_sharding_applies(n: int, m: int, world: int) -> bool:
    """the quotient coset splits over `world` ranks when world divides m / n (8 for Turbo/UltraPlonk)"""
    return world > 1 and (m // n) % world == 0 and n > 1


def slice_offset(cp, m: int, rank: int) -> int:
    """the points i = rank (mod G) of the quotient coset {h w_m^i} are the coset of size m / G with
    offset h w_m^rank (h = Fr::GENERATOR, prover.rs:524-527)"""
    w_m = pow(cp.two_adic_root, 1 << (cp.two_adicity - (m.bit_length() - 1)), cp.r)
    return cp.fr_generator * pow(w_m, rank, cp.r) % cp.r


SKIP_ZERO_POLYS = True   # False: transform every polynomial like the reference does (bench comparison; same results)


def _coset_ntt_rows(c: int, cs: torch.Tensor, zero_rows, log_len: int, off, in_len: int, st):
    """forward coset NTT of the rows of `cs` (one polynomial per row, first in_len coefficients set), except the rows in
    zero_rows, which stand for the zero polynomial: those are left alone (the masked quotient call never reads them).
    The reference transforms all of them (prover.rs:534-549); an identically-zero selector (a gate kind the circuit
    never uses) or an empty public input has zero evaluations anyway."""
    rows = cs.shape[0]
    zero_rows = set(zero_rows)
    j = 0
    while j < rows:
        if j in zero_rows:
            j += 1
            continue
        k = j
        while k < rows and k not in zero_rows:
            k += 1
        L.ntt_device(c, cs[j].data_ptr(), log_len, k - j, inverse=False, coset_mont=off, in_len=in_len, stream=st)
        j = k


def unshard_coeffs(cp, m: int, G: int) -> list:
    """gathered[g] = t(X) mod (X^ml - c_g), c_g = h^ml w_G^g (rank g's coset iNTT of its slice of the quotient
    evaluations), ml = m / G.  With t = sum_c X^(c ml) t_c:  gathered[g][j] = sum_c (h^ml)^c w_G^(g c) t_c[j], a size-G DFT
    per coefficient index; its inverse  t_c = sum_g coeffs[c][g] * gathered[g],  coeffs[c][g] = h^(-ml c) w_G^(-g c) / G."""
    r = cp.r
    ml = m // G
    w_m = pow(cp.two_adic_root, 1 << (cp.two_adicity - (m.bit_length() - 1)), r)
    w_G_inv = pow(pow(w_m, ml, r), -1, r)
    h_ml_inv = pow(pow(cp.fr_generator, ml, r), -1, r)
    g_inv = pow(G, -1, r)
    return [[pow(h_ml_inv, c, r) * g_inv % r * pow(w_G_inv, g * c, r) % r for g in range(G)] for c in range(G)]


def unshard_quotient(cp, curve: int, gathered: torch.Tensor, m: int, out: torch.Tensor, stream):
    """quotient coefficients from the G gathered slice polynomials (see unshard_coeffs): G lincombs of G terms"""
    G = gathered.shape[0]
    ml = m // G
    ptrs = [gathered[g].data_ptr() for g in range(G)]
    for c, coeffs in enumerate(unshard_coeffs(cp, m, G)):
        L.poly_lincomb_device(curve, ptrs, [ml] * G, H.fr_to_mont(cp, coeffs), out.data_ptr() + c * ml * 32, ml,
                              stream=stream)


class PlonkKzgSnark:
    @staticmethod
    def universal_setup_for_testing(curve: int, max_degree: int, beta: int, g=None, h=None) -> UniversalSrs:
        """[beta^i]g for i <= max_degree on the device, h and [beta]h (snark.rs:447-488).  beta, g, h are arguments
        (the reference draws them from its RNG)."""
        cp = H.CURVES[curve]
        g = g or cp.g1
        h = h or cp.g2
        L.load_library()
        srs = L.Srs.gen_for_testing(curve, H.fr_to_mont(cp, [beta])[0], H.g1_to_mont(cp, g), max_degree + 1)
        return UniversalSrs(curve, srs, g, h, H.g2_mul(cp, beta, h))

    @staticmethod
    def preprocess(srs: UniversalSrs, circuit: PlonkCircuit, stream=None):
        """selector / sigma polynomials and their commitments (snark.rs:492-580), all on the device"""
        cp = circuit.cp
        c = cp.curve_id
        n, log_n, W = circuit.eval_domain_size(), circuit.log_n, circuit.num_wire_types
        if srs.max_degree() < circuit.srs_size():
            raise PlonkError("IndexTooLarge")
        st = _stream(stream)
        # selector values -> coefficients (compute_selector_polynomials, constraint_system.rs:1359-1372)
        table = np.stack([H.fr_to_mont(cp, k) for k in circuit.gate_kinds])             # (kinds, 13, 4)
        sel_vals = np.ascontiguousarray(table[circuit.gate_kind].transpose(1, 0, 2))        # (13, n, 4)
        selectors = _dev_fr(sel_vals).view(N_TURBO_PLONK_SELECTORS, n * 4)
        L.ntt_device(c, selectors.data_ptr(), log_n, N_TURBO_PLONK_SELECTORS, inverse=True, stream=st)
        # extended identity permutation id[i*n + j] = k_i g^j (constraint_system.rs:1116-1132): the forward NTT of k_i X
        idp = torch.zeros(W, n * 4, dtype=torch.int64, device="cuda")
        idp[:, 4:8] = _dev_fr(H.fr_to_mont(cp, circuit.k)).view(W, 4)
        if n > 1:
            L.ntt_device(c, idp.data_ptr(), log_n, W, inverse=False, stream=st)
        # extended permutation (constraint_system.rs:1135-1161) and its polynomials (:1374-1392)
        perm = torch.from_numpy(circuit.wire_permutation).cuda()
        sigma_vals = idp.view(W * n, 4)[perm].contiguous().view(W, n * 4)
        sigmas = sigma_vals.clone()
        L.ntt_device(c, sigmas.data_ptr(), log_n, W, inverse=True, stream=st)
        comms = _commit_many(srs.powers_of_g, cp, [selectors[j] for j in range(N_TURBO_PLONK_SELECTORS)]
                             + [sigmas[j] for j in range(W)], [n] * (N_TURBO_PLONK_SELECTORS + W), st)
        vk = VerifyingKey(n, circuit.num_inputs(), comms[N_TURBO_PLONK_SELECTORS:], comms[:N_TURBO_PLONK_SELECTORS],
                          list(circuit.k), srs.g, srs.h, srs.beta_h)
        pk = ProvingKey(c, sigmas, selectors, srs.powers_of_g, vk, idp, sigma_vals,
                        torch.from_numpy(circuit.wire_variables.reshape(-1)).cuda())
        pk.selector_is_zero = tuple(bool(z) for z in (~torch.any(selectors != 0, dim=1)).cpu().tolist())
        return pk, vk

    @staticmethod
    def prove(rng, circuit: PlonkCircuit, prove_key: ProvingKey, extra_transcript_init_msg: bytes | None = None,
              stream=None, witness_mont: np.ndarray | None = None, sharded: bool = False, group=None) -> Proof:
        """One TurboPlonk proof (snark.rs:587-613 -> batch_prove_internal :163-430 with one instance).
        witness_mont: optional pre-converted witness table ((num_vars, 4) uint64 Montgomery), else converted here.
        sharded: every rank of `group` calls prove with the same inputs (same RNG state); the 13 commitments' point
        ranges are partitioned over the ranks (see _Committer); transforms and quotient run on every rank."""
        bp = PlonkKzgSnark._batch_prove_internal(rng, [circuit], [prove_key], extra_transcript_init_msg, stream,
                                                 None if witness_mont is None else [witness_mont], sharded, group)
        return bp.to_proof()

    @staticmethod
    def batch_prove(rng, circuits, prove_keys, extra_transcript_init_msg: bytes | None = None, stream=None,
                    witness_monts=None, sharded: bool = False, group=None) -> "BatchProof":
        """k TurboPlonk instances -> ONE aggregated BatchProof (snark.rs:61-74 -> :163-430): per round the instances
        run in order on the one RNG and the one transcript, the quotient is sum_i alpha^(3i) t_i (prover.rs:519-656)
        and the two opening proofs fold every instance's polynomials (prover.rs:356-413)."""
        return PlonkKzgSnark._batch_prove_internal(rng, list(circuits), list(prove_keys), extra_transcript_init_msg,
                                                   stream, witness_monts, sharded, group)

    @staticmethod
    def _batch_prove_internal(rng, circuits, prove_keys, extra_transcript_init_msg, stream, witness_monts, sharded,
                              group) -> "BatchProof":
        if not circuits:
            raise PlonkError("zero number of circuits/proving keys")
        if len(circuits) != len(prove_keys):
            raise PlonkError("the number of circuits %d != the number of proving keys %d"
                             % (len(circuits), len(prove_keys)))
        K = len(circuits)
        cp = circuits[0].cp
        c, r = cp.curve_id, cp.r
        n = circuits[0].eval_domain_size()
        W = circuits[0].num_wire_types
        for circuit, pk in zip(circuits, prove_keys):
            if circuit.eval_domain_size() != n:
                raise PlonkError("circuit domain size %d != expected domain size %d" % (circuit.eval_domain_size(), n))
            if pk.domain_size() != n:
                raise PlonkError("proving key domain size %d != expected domain size %d" % (pk.domain_size(), n))
            if circuit.num_inputs() != pk.vk.num_inputs:
                raise PlonkError("circuit.num_inputs %d != prove_key.num_inputs %d"
                                 % (circuit.num_inputs(), pk.vk.num_inputs))
            if circuit.num_wire_types != W or circuit.cp is not cp:
                raise PlonkError("inconsistent plonk circuit types")
        log_n = circuits[0].log_n
        m = 1
        while m < n * domain_size_ratio(n, W):
            m <<= 1
        log_m = m.bit_length() - 1
        n4 = n * 4
        ck = prove_keys[0].commit_key        # run_3rd_round / compute_opening_proofs use prove_keys[0].commit_key
        st = _stream(stream)
        commits = [_Committer(pk.commit_key, cp, st, sharded, group) for pk in prove_keys] if K > 1 else \
            [_Committer(ck, cp, st, sharded, group)]
        commit0 = commits[0]
        mont = lambda xs: H.fr_to_mont(cp, xs)  # noqa: E731
        gen = mont([cp.fr_generator])[0]
        group_gen = pow(cp.two_adic_root, 1 << (cp.two_adicity - log_n), r)

        import time as _time

        t_last = [_time.perf_counter()]

        def tick(name):
            if PROFILE is not None:
                torch.cuda.synchronize()
                now = _time.perf_counter()
                PROFILE[name] = round(PROFILE.get(name, 0.0) + (now - t_last[0]) * 1e3, 3)
                t_last[0] = now

        pub_inputs = [circuit.public_input() for circuit in circuits]

        def mask(poly: torch.Tensor, hiding_bound: int):
            """mask_polynomial (prover.rs:457-466): + (b_0 + .. + b_h X^h)(X^n - 1); poly has room for n + h + 1"""
            b = [H.fr_rand(cp, rng) for _ in range(hiding_bound + 1)]
            k = 4 * (hiding_bound + 1)
            low = _read_fr(cp, poly[:k])
            poly[:k] = _dev_fr(mont([(x - y) % r for x, y in zip(low, b)]))
            poly[n4:n4 + k] = _dev_fr(mont(b))

        def lincomb(terms, out_len: int) -> torch.Tensor:
            """sum_k s_k * poly_k for (scalar, device poly, length) terms; jfhip_poly_lincomb_device takes <= 32
            polynomials per call, longer sums chain through the running total"""
            out = torch.zeros(out_len * 4, dtype=torch.int64, device="cuda")
            acc = None
            for at in range(0, len(terms), 31):
                chunk = list(terms[at:at + 31])
                if acc is not None:
                    chunk.append((1, acc, out_len))
                    out = torch.zeros(out_len * 4, dtype=torch.int64, device="cuda")
                L.poly_lincomb_device(c, [p.data_ptr() for _, p, _ in chunk], [k for _, _, k in chunk],
                                      mont([s % r for s, _, _ in chunk]), out.data_ptr(), out_len, stream=st)
                acc = out
            return out

        # ---- Round 1 (prover.rs:77-91): wire polynomials, public-input polynomial; instances in order
        ln = n + 3
        l4 = ln * 4
        wire_vals_v, wire_polys_v, pub_poly_v, wires_poly_comms_vec, bufs = [], [], [], [], []
        for i, (circuit, pk) in enumerate(zip(circuits, prove_keys)):
            wm = witness_monts[i] if witness_monts is not None and witness_monts[i] is not None else mont(circuit.witness)
            wit = _dev_fr(wm).view(-1, 4)
            if wit.shape[0] != circuit.num_vars():
                raise PlonkError("witness table has %d rows, the circuit %d variables" % (wit.shape[0], circuit.num_vars()))
            # witness[wire_variable(i, j)] for every (wire type, gate) (constraint_system.rs:1422-1444): one gather kernel
            wire_vals = torch.empty(W, n4, dtype=torch.int64, device="cuda")
            L.witness_gather_device(wit.data_ptr(), wit.shape[0], pk.wire_variables32.data_ptr(), W * n,
                                    wire_vals.data_ptr(), stream=st)
            buf = torch.empty(W + 1, n4, dtype=torch.int64, device="cuda")
            buf[:W].copy_(wire_vals)
            buf[W].zero_()
            if pub_inputs[i]:
                buf[W, :4 * len(pub_inputs[i])] = _dev_fr(mont(pub_inputs[i]))   # io gates sit in rows 0 .. num_inputs-1
            # an empty public input is the zero polynomial: nothing to transform in row W
            L.ntt_device(c, buf.data_ptr(), log_n, W + (1 if pub_inputs[i] or not SKIP_ZERO_POLYS else 0), inverse=True,
                         stream=st)
            wire_vals_v.append(wire_vals)
            bufs.append(buf)
        # the transcript is initialised (snark.rs:231-237) while the device runs the witness gather and the iNTTs
        tr = StandardTranscript(cp)
        if extra_transcript_init_msg is not None:
            tr.append_message(b"extra info", extra_transcript_init_msg)
        for pk, pub_input in zip(prove_keys, pub_inputs):
            tr.append_vk_and_pub_input(pk.vk, pub_input)
        tick("r1_gather_intt_transcript_init")
        for i, (circuit, pk) in enumerate(zip(circuits, prove_keys)):
            buf = bufs[i]
            wire_polys = torch.zeros(W, l4, dtype=torch.int64, device="cuda")
            wire_polys[:, :n4].copy_(buf[:W])
            # mask_polynomial for the W wire polynomials with ONE read-back and two uploads (RNG draws in the
            # reference's order: b_0, b_1 of wire 0, then wire 1, ...)
            bs = [[H.fr_rand(cp, rng) for _ in range(2)] for _ in range(W)]
            low = _read_fr(cp, wire_polys[:, :8].contiguous().view(-1))
            wire_polys[:, :8] = _dev_fr(mont([(low[2 * j + t] - bs[j][t]) % r for j in range(W) for t in range(2)])).view(W, 8)
            wire_polys[:, n4:n4 + 8] = _dev_fr(mont([b for row in bs for b in row])).view(W, 8)
            tick("r1_mask")
            comms = commits[i]([wire_polys[j] for j in range(W)], [n + 2] * W)
            tr.append_commitments(b"witness_poly_comms", comms)
            tick("r1_commit")
            wire_polys_v.append(wire_polys)
            pub_poly_v.append(buf[W])
            wires_poly_comms_vec.append(comms)
        del bufs

        # ---- Round 2 (prover.rs:128-142; grand product constraint_system.rs:1394-1420)
        beta = tr.get_challenge(b"beta")
        gamma = tr.get_challenge(b"gamma")
        bg = mont([beta, gamma])
        perm_poly_v, prod_perm_poly_comms_vec = [], []
        for i, pk in enumerate(prove_keys):
            perm_poly = torch.zeros(l4, dtype=torch.int64, device="cuda")
            L.plonk_perm_product_device(c, log_n, W, wire_vals_v[i].data_ptr(), pk.id_perm_vals.data_ptr(),
                                        pk.sigma_vals.data_ptr(), bg[0], bg[1], perm_poly.data_ptr(), stream=st)
            L.ntt_device(c, perm_poly.data_ptr(), log_n, 1, inverse=True, stream=st)
            mask(perm_poly, 2)
            tick("r2_product_intt_mask")
            comm = commits[i]([perm_poly], [n + 3])[0]
            tr.append_commitments(b"perm_poly_comms", [comm])
            tick("r2_commit")
            perm_poly_v.append(perm_poly)
            prod_perm_poly_comms_vec.append(comm)
        del wire_vals_v

        # ---- Round 3 (prover.rs:183-198, 491-661, 894-943)
        alpha = tr.get_challenge(b"alpha")
        alpha_3 = pow(alpha, 3, r)
        alpha_bases = [pow(alpha_3, i, r) for i in range(K)]
        NSEL = N_TURBO_PLONK_SELECTORS
        # coset sharding: rank g of G evaluates the slice i = g (mod G) of the quotient coset, a coset of
        # size ml = m / G with offset h w_m^g; one all_gather of the slices' iNTTs rebuilds the quotient polynomial
        G, g = 1, 0
        if sharded:
            import torch.distributed as tdist

            G = tdist.get_world_size(group)
            g = tdist.get_rank(group)
            if not coset_sharding_applies(n, m, G):
                G, g = 1, 0
        ml = m // G
        log_ml = ml.bit_length() - 1
        off_int = slice_offset(cp, m, g)
        off = mont([off_int])[0]
        lc4 = min(l4, ml * 4)
        quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
        part = torch.zeros(ml * 4, dtype=torch.int64, device="cuda") if G > 1 else None
        for i, pk in enumerate(prove_keys):
            vk, wire_polys, perm_poly = pk.vk, wire_polys_v[i], perm_poly_v[i]
            if pk.coset_evals is not None:
                # selector / sigma coset evaluations are resident with the proving key: 7 transforms instead of 25
                fixed = pk.coset_evals
                if fixed.shape != (NSEL + W, ml * 4) or pk.coset_slice != (g, G):
                    raise PlonkError("proving key coset evaluations were kept for another slice of the quotient coset")
                npolys, o_w = W + 2, 0
                sel_ptr, sig_ptr = fixed[0].data_ptr(), fixed[NSEL].data_ptr()
            else:
                npolys, o_w = NSEL + 2 * W + 2, NSEL + W
            o_z, o_pi = o_w + W, o_w + W + 1
            cs = torch.empty(npolys, ml * 4, dtype=torch.int64, device="cuda")
            cs[:, n4:lc4].zero_()
            if pk.coset_evals is None:
                cs[:NSEL, :n4].copy_(pk.selectors)
                cs[NSEL:o_w, :n4].copy_(pk.sigmas)
                sel_ptr, sig_ptr = cs[0].data_ptr(), cs[NSEL].data_ptr()
            cs[o_w:o_z, :lc4].copy_(wire_polys[:, :lc4])
            cs[o_z, :lc4].copy_(perm_poly[:lc4])
            cs[o_pi, :n4].copy_(pub_poly_v[i])
            if lc4 < l4:
                # ml = n < n + 3 coefficients: reduce mod X^ml - off^ml (the slice's vanishing polynomial)
                fold = mont([1, pow(off_int, ml, r)])
                k = (l4 - lc4) // 4
                tmp = torch.empty(W + 1, k * 4, dtype=torch.int64, device="cuda")
                for j, poly in enumerate([wire_polys[j] for j in range(W)] + [perm_poly]):
                    L.poly_lincomb_device(c, [poly.data_ptr(), poly.data_ptr() + lc4 * 8], [k, k], fold,
                                          tmp[j].data_ptr(), k, stream=st)
                cs[o_w:o_z + 1, :k * 4].copy_(tmp)
            # identically-zero polynomials: no transform, and the quotient kernel is told not to read their rows
            zmask = sum(1 << j for j, z in enumerate(pk.selector_is_zero) if z) if SKIP_ZERO_POLYS else 0
            pi_zero = SKIP_ZERO_POLYS and not pub_inputs[i]
            zero_rows = [o_pi] if pi_zero else []
            if pk.coset_evals is None:
                zero_rows += [j for j in range(NSEL) if (zmask >> j) & 1]
            _coset_ntt_rows(c, cs, zero_rows, log_ml, off, lc4 // 4, st)
            tick("r3_coset_ntts")
            chal = mont([alpha, beta, gamma, alpha_bases[i]])
            # G == 1: the slice is the whole quotient coset (offset Fr::GENERATOR, log_ml = log_m)
            L.plonk_quotient_masked_device(c, W, log_n, log_ml, off, sel_ptr, sig_ptr, cs[o_w].data_ptr(),
                                           cs[o_z].data_ptr(), 0 if pi_zero else cs[o_pi].data_ptr(), 0, mont(vk.k), chal,
                                           zmask, (quot if G == 1 else part).data_ptr(), stream=st)
            del cs                             # stream-ordered: the next allocation is used on the same stream
            tick("r3_quotient_intt")
        if G == 1:
            L.ntt_device(c, quot.data_ptr(), log_m, 1, inverse=True, coset_mont=gen, stream=st)
        else:
            L.ntt_device(c, part.data_ptr(), log_ml, 1, inverse=True, coset_mont=off, stream=st)
            st.sync()
            if tdist.get_backend(group) == "gloo":
                host = torch.empty(G * ml * 4, dtype=torch.int64)
                tdist.all_gather_into_tensor(host, part.cpu(), group=group)
                gathered = host.cuda().view(G, ml * 4)
            else:
                gathered = torch.empty(G * ml * 4, dtype=torch.int64, device="cuda")
                tdist.all_gather_into_tensor(gathered, part, group=group)
                torch.cuda.current_stream().synchronize()
                gathered = gathered.view(G, ml * 4)
            unshard_quotient(cp, c, gathered, m, quot, st)
        tick("r3_quotient_intt")
        expected_degree = W * (n + 1) + 2
        # the split randomisers (prover.rs:926-940) are drawn while the device still works on the quotient: the next RNG
        # draws in the reference's order whether or not the degree check below passes
        randomizers = [H.fr_rand(cp, rng) for _ in range(W - 1)]
        randomizers_dev = _dev_fr(mont(randomizers)).view(W - 1, 4)
        split = torch.zeros(W, l4, dtype=torch.int64, device="cuda")
        step = (n + 2) * 4
        for i in range(W - 1):
            split[i, :step].copy_(quot[i * step:(i + 1) * step])
        last_len = expected_degree + 1 - (W - 1) * (n + 2)
        split[W - 1, :last_len * 4].copy_(quot[(W - 1) * step:(expected_degree + 1) * 4])
        split[:W - 1, step:step + 4] = randomizers_dev
        # ONE read-back: [tail non-zero flag, top coefficient, first coefficient of every piece]
        flag = torch.any(quot[(expected_degree + 1) * 4:] != 0).to(torch.int64).view(1)
        back = torch.cat([flag, quot[expected_degree * 4:(expected_degree + 1) * 4],
                          split[:, :4].contiguous().view(-1)]).cpu().numpy()
        del quot
        if back[0] != 0 or not back[1:5].any():
            raise PlonkError("WrongQuotientPolyDegree(expected %d)" % expected_degree)
        firsts = H.fr_from_mont(cp, back[5:].view(np.uint64))
        last_randomizer = 0
        for i in range(W - 1):
            firsts[i] = (firsts[i] - last_randomizer) % r
            last_randomizer = randomizers[i]
        firsts[W - 1] = (firsts[W - 1] - last_randomizer) % r
        split[:, :4] = _dev_fr(mont(firsts)).view(W, 4)
        split_lens = [n + 3] * (W - 1) + [last_len]
        tick("r3_split_mask")
        split_quot_poly_comms = commit0([split[j] for j in range(W)], split_lens)
        tr.append_commitments(b"quot_poly_comms", split_quot_poly_comms)
        tick("r3_commit")

        # ---- Round 4 (prover.rs:205-229): evaluations, instances in order
        zeta = tr.get_challenge(b"zeta")
        zeta_omega = zeta * group_gen % r
        zm, zom = mont([zeta])[0], mont([zeta_omega])[0]
        evals_dev = torch.zeros(K * (2 * W) * 4, dtype=torch.int64, device="cuda")
        at_zeta_v = []
        for i, pk in enumerate(prove_keys):
            at_zeta = [(wire_polys_v[i][j], n + 2) for j in range(W)] + [(pk.sigmas[j], n) for j in range(W - 1)]
            base = evals_dev.data_ptr() + i * 2 * W * 32
            L.poly_eval_batch_device(c, [p.data_ptr() for p, _ in at_zeta], [k for _, k in at_zeta], zm, base, stream=st)
            L.poly_eval_batch_device(c, [perm_poly_v[i].data_ptr()], [n + 3], zom, base + (2 * W - 1) * 32, stream=st)
            at_zeta_v.append(at_zeta)
        ev_all = _read_fr(cp, evals_dev)
        poly_evals_vec = []
        for i in range(K):
            ev = ev_all[i * 2 * W:(i + 1) * 2 * W]
            pe = ProofEvaluations(ev[:W], ev[W:2 * W - 1], ev[2 * W - 1])
            tr.append_field_elems(b"wire_evals", pe.wires_evals)
            tr.append_field_elems(b"wire_sigma_evals", pe.wire_sigma_evals)
            tr.append_field_elems(b"perm_next_eval", [pe.perm_next_eval])
            poly_evals_vec.append(pe)
        tick("r4_evals")

        # linearisation polynomial (snark.rs:364-389; prover.rs:299-352, 946-1015) as device linear combinations
        vanish_eval = (pow(zeta, n, r) - 1) % r
        zeta_to_n_plus_2 = (vanish_eval + 1) * zeta % r * zeta % r
        lagrange_1_eval = vanish_eval * pow(n * (zeta - 1) % r, -1, r) % r
        terms = []
        coeff = 1
        for i in range(W):
            if i:
                coeff = coeff * zeta_to_n_plus_2 % r
            terms.append((-vanish_eval * coeff % r, split[i], split_lens[i]))
        for i, pk in enumerate(prove_keys):
            ab, vk, pe = alpha_bases[i], pk.vk, poly_evals_vec[i]
            w = pe.wires_evals
            q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r),
                         pow(w[2], 5, r), pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
            terms += [(s % r * ab % r, pk.selectors[j], n) for j, s in enumerate(q_scalars)]
            cz = alpha
            for j in range(W):
                cz = cz * (w[j] + beta * vk.k[j] % r * zeta + gamma) % r
            cz = (cz + alpha * alpha % r * lagrange_1_eval) % r
            terms.append((cz * ab % r, perm_poly_v[i], n + 3))
            csig = alpha * beta % r * pe.perm_next_eval % r
            for j in range(W - 1):
                csig = csig * (w[j] + beta * pe.wire_sigma_evals[j] + gamma) % r
            terms.append((-csig * ab % r, pk.sigmas[W - 1], n))
        lin_poly = lincomb(terms, n + 3)
        tick("r4_lin_poly")

        # ---- Round 5 (prover.rs:356-413, 470-488): batched opening quotients over every instance
        v = tr.get_challenge(b"v")
        polys = [(lin_poly, n + 3)]
        for at_zeta in at_zeta_v:
            polys += at_zeta
        open_terms, cur = [], 1
        for p, k in polys:
            open_terms.append((cur, p, k))
            cur = cur * v % r
        open_poly = lincomb(open_terms, n + 3)
        if K == 1:
            shifted_poly = perm_poly_v[0]
        else:
            sh_terms, cur = [], 1
            for z in perm_poly_v:
                sh_terms.append((cur, z, n + 3))
                cur = cur * v % r
            shifted_poly = lincomb(sh_terms, n + 3)
        wpoly = torch.zeros(2, l4, dtype=torch.int64, device="cuda")
        L.poly_divide_linear_device(c, open_poly.data_ptr(), n + 3, zm, wpoly[0].data_ptr(), stream=st)
        L.poly_divide_linear_device(c, shifted_poly.data_ptr(), n + 3, zom, wpoly[1].data_ptr(), stream=st)
        tick("r5_open_polys")
        opening_proof, shifted_opening_proof = commit0([wpoly[0][4:], wpoly[1][4:]], [n + 2, n + 2])
        tick("r5_commit")

        return BatchProof(c, wires_poly_comms_vec, prod_perm_poly_comms_vec, poly_evals_vec, split_quot_poly_comms,
                          opening_proof, shifted_opening_proof)

    @staticmethod
    def batch_prove_one_per_rank(rng, circuit: PlonkCircuit, prove_key: ProvingKey,
                                 extra_transcript_init_msg: bytes | None = None, stream=None,
                                 witness_mont: np.ndarray | None = None, group=None) -> "BatchProof":
        """batch_prove with ONE circuit per GPU (BASELINE config 5's layout): rank i of `group` holds
        instance i (its circuit and proving key) and every rank returns the SAME BatchProof, byte-identical to
        batch_prove(rng, [circuit_0 .. circuit_{G-1}], ...) on one GPU from the same RNG state.
        * every rank runs the same transcript and the same RNG (it draws every instance's blinding in the reference's
          order, snark.rs:245-318, prover.rs:926-940, and uses its own);
        * per-instance commitments and evaluations are exchanged with one small all_gather per round;
        * everything that mixes instances -- the quotient sum_i alpha^(3i) t_i, its split, the linearisation and the
          two opening polynomials (prover.rs:644-655, 356-413) -- is linear in the instances: each rank keeps its own
          contribution, commits it without normalising (jfhip_msm_partial_device) and the aggregated commitment is
          the group sum of the gathered partial sums.  No polynomial crosses xGMI."""
        import torch.distributed as tdist

        from . import dist as jdist

        G = tdist.get_world_size(group)
        me = tdist.get_rank(group)
        cp = circuit.cp
        c, r = cp.curve_id, cp.r
        pk, vk = prove_key, prove_key.vk
        n, log_n, W = circuit.eval_domain_size(), circuit.log_n, circuit.num_wire_types
        if pk.domain_size() != n:
            raise PlonkError("proving key domain size %d != expected domain size %d" % (pk.domain_size(), n))
        if circuit.num_inputs() != vk.num_inputs:
            raise PlonkError("circuit.num_inputs %d != prove_key.num_inputs %d" % (circuit.num_inputs(), vk.num_inputs))
        m = 1
        while m < n * domain_size_ratio(n, W):
            m <<= 1
        log_m = m.bit_length() - 1
        n4, ln = n * 4, n + 3
        l4 = ln * 4
        st = _stream(stream)
        commit = _Committer(pk.commit_key, cp, st)
        engine = jdist.HipMsmEngine(L, pk.commit_key, st)
        mont = lambda xs: H.fr_to_mont(cp, xs)  # noqa: E731
        gen = mont([cp.fr_generator])[0]
        group_gen = pow(cp.two_adic_root, 1 << (cp.two_adicity - log_n), r)
        rec = 16 * cp.fq_limbs64 + 4

        def gather_points(points) -> list:
            """[rank][j] of every rank's affine points (packed records through one all_gather)"""
            raw = bytearray()
            for P in points:
                raw += H.g1_to_record(cp, P)
            mine = torch.frombuffer(raw, dtype=torch.uint8).cuda()
            out = jdist._all_gather_bytes(mine, G, group).cpu().numpy().tobytes()
            k = len(points)
            return [[H.g1_from_record(cp, out[(q * k + j) * rec:(q * k + j + 1) * rec]) for j in range(k)]
                    for q in range(G)]

        def gather_ints(xs) -> list:
            raw = b"".join(int(x).to_bytes(32, "little") for x in xs)
            mine = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
            out = jdist._all_gather_bytes(mine, G, group).cpu().numpy().tobytes()
            k = len(xs)
            return [[int.from_bytes(out[(q * k + j) * 32:(q * k + j + 1) * 32], "little") for j in range(k)]
                    for q in range(G)]

        def aggregated_commit(polys, lens) -> list:
            """sum over ranks of commit(rank's polynomial): un-normalised partial sums, one all_gather, local fold"""
            for k in lens:
                if k - 1 > len(pk.commit_key) - 1:
                    raise PlonkError("poly degree %d is larger than allowed %d" % (k - 1, len(pk.commit_key) - 1))
            parts = torch.cat([engine.partial(p[:4 * k]) for p, k in zip(polys, lens)])
            st.sync()
            allp = jdist._all_gather_bytes(parts, G, group).view(G, len(polys), engine.partial_bytes)
            per = allp.transpose(0, 1).contiguous()
            out = []
            for j in range(len(polys)):
                raw = engine.fold(per[j].reshape(-1), G).cpu().numpy().tobytes()
                out.append(H.g1_from_record(cp, raw))
            return out

        def lincomb(terms, out_len: int) -> torch.Tensor:
            out = torch.zeros(out_len * 4, dtype=torch.int64, device="cuda")
            L.poly_lincomb_device(c, [p.data_ptr() for _, p, _ in terms], [k for _, _, k in terms],
                                  mont([s % r for s, _, _ in terms]), out.data_ptr(), out_len, stream=st)
            return out

        # every rank needs every instance's verifying key and public input for the shared transcript
        pub_input = circuit.public_input()
        everyone = [None] * G
        tdist.all_gather_object(everyone, (vk, pub_input), group=group)
        for other_vk, _ in everyone:
            if other_vk.domain_size != n:
                raise PlonkError("circuit domain size %d != expected domain size %d" % (other_vk.domain_size, n))
        tr = StandardTranscript(cp)
        if extra_transcript_init_msg is not None:
            tr.append_message(b"extra info", extra_transcript_init_msg)
        for other_vk, other_pub in everyone:
            tr.append_vk_and_pub_input(other_vk, other_pub)

        # ---- Round 1: the RNG draws of all instances in order, this rank keeps its own
        bs = [[[H.fr_rand(cp, rng) for _ in range(2)] for _ in range(W)] for _ in range(G)][me]
        wit = _dev_fr(witness_mont if witness_mont is not None else mont(circuit.witness)).view(-1, 4)
        if wit.shape[0] != circuit.num_vars():
            raise PlonkError("witness table has %d rows, the circuit %d variables" % (wit.shape[0], circuit.num_vars()))
        wire_vals = torch.empty(W, n4, dtype=torch.int64, device="cuda")
        L.witness_gather_device(wit.data_ptr(), wit.shape[0], pk.wire_variables32.data_ptr(), W * n, wire_vals.data_ptr(),
                                stream=st)
        buf = torch.empty(W + 1, n4, dtype=torch.int64, device="cuda")
        buf[:W].copy_(wire_vals)
        buf[W].zero_()
        if pub_input:
            buf[W, :4 * len(pub_input)] = _dev_fr(mont(pub_input))
        L.ntt_device(c, buf.data_ptr(), log_n, W + 1, inverse=True, stream=st)
        wire_polys = torch.zeros(W, l4, dtype=torch.int64, device="cuda")
        wire_polys[:, :n4].copy_(buf[:W])
        pub_poly = buf[W]
        low = _read_fr(cp, wire_polys[:, :8].contiguous().view(-1))
        wire_polys[:, :8] = _dev_fr(mont([(low[2 * j + t] - bs[j][t]) % r for j in range(W) for t in range(2)])).view(W, 8)
        wire_polys[:, n4:n4 + 8] = _dev_fr(mont([b for row in bs for b in row])).view(W, 8)
        wires_poly_comms_vec = gather_points(commit([wire_polys[j] for j in range(W)], [n + 2] * W))
        for comms in wires_poly_comms_vec:
            tr.append_commitments(b"witness_poly_comms", comms)

        # ---- Round 2
        beta = tr.get_challenge(b"beta")
        gamma = tr.get_challenge(b"gamma")
        bg = mont([beta, gamma])
        b = [[H.fr_rand(cp, rng) for _ in range(3)] for _ in range(G)][me]
        perm_poly = torch.zeros(l4, dtype=torch.int64, device="cuda")
        L.plonk_perm_product_device(c, log_n, W, wire_vals.data_ptr(), pk.id_perm_vals.data_ptr(),
                                    pk.sigma_vals.data_ptr(), bg[0], bg[1], perm_poly.data_ptr(), stream=st)
        L.ntt_device(c, perm_poly.data_ptr(), log_n, 1, inverse=True, stream=st)
        low = _read_fr(cp, perm_poly[:12])
        perm_poly[:12] = _dev_fr(mont([(x - y) % r for x, y in zip(low, b)]))
        perm_poly[n4:n4 + 12] = _dev_fr(mont(b))
        prod_perm_poly_comms_vec = [row[0] for row in gather_points(commit([perm_poly], [n + 3]))]
        tr.append_commitments(b"perm_poly_comms", prod_perm_poly_comms_vec)

        # ---- Round 3: this rank's share alpha^(3 me) t_me of the quotient
        alpha = tr.get_challenge(b"alpha")
        alpha_base = pow(alpha, 3 * me, r)
        NSEL = N_TURBO_PLONK_SELECTORS
        if pk.coset_evals is not None:
            fixed = pk.coset_evals
            if fixed.shape != (NSEL + W, m * 4) or pk.coset_slice != (0, 1):
                raise PlonkError("proving key coset evaluations were kept for another slice of the quotient coset")
            npolys, o_w = W + 2, 0
            sel_ptr, sig_ptr = fixed[0].data_ptr(), fixed[NSEL].data_ptr()
        else:
            npolys, o_w = NSEL + 2 * W + 2, NSEL + W
        o_z, o_pi = o_w + W, o_w + W + 1
        cs = torch.empty(npolys, m * 4, dtype=torch.int64, device="cuda")
        cs[:, n4:l4].zero_()
        if pk.coset_evals is None:
            cs[:NSEL, :n4].copy_(pk.selectors)
            cs[NSEL:o_w, :n4].copy_(pk.sigmas)
            sel_ptr, sig_ptr = cs[0].data_ptr(), cs[NSEL].data_ptr()
        cs[o_w:o_z, :l4].copy_(wire_polys)
        cs[o_z, :l4].copy_(perm_poly)
        cs[o_pi, :n4].copy_(pub_poly)
        zmask = sum(1 << j for j, z in enumerate(pk.selector_is_zero) if z) if SKIP_ZERO_POLYS else 0
        pi_zero = SKIP_ZERO_POLYS and not pub_input
        zero_rows = [o_pi] if pi_zero else []
        if pk.coset_evals is None:
            zero_rows += [j for j in range(NSEL) if (zmask >> j) & 1]
        _coset_ntt_rows(c, cs, zero_rows, log_m, gen, ln, st)
        quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
        L.plonk_quotient_masked_device(c, W, log_n, log_m, gen, sel_ptr, sig_ptr, cs[o_w].data_ptr(), cs[o_z].data_ptr(),
                                       0 if pi_zero else cs[o_pi].data_ptr(), 0, mont(vk.k),
                                       mont([alpha, beta, gamma, alpha_base]), zmask, quot.data_ptr(), stream=st)
        del cs
        L.ntt_device(c, quot.data_ptr(), log_m, 1, inverse=True, coset_mont=gen, stream=st)
        expected_degree = W * (n + 1) + 2
        tail_nonzero = int(bool(torch.any(quot[(expected_degree + 1) * 4:] != 0).item()))
        top = _read_fr(cp, quot[expected_degree * 4:(expected_degree + 1) * 4])[0]
        checks = gather_ints([tail_nonzero, top])
        if any(t for t, _ in checks) or sum(x for _, x in checks) % r == 0:
            raise PlonkError("WrongQuotientPolyDegree(expected %d)" % expected_degree)
        split = torch.zeros(W, l4, dtype=torch.int64, device="cuda")
        step = (n + 2) * 4
        for i in range(W - 1):
            split[i, :step].copy_(quot[i * step:(i + 1) * step])
        last_len = expected_degree + 1 - (W - 1) * (n + 2)
        split[W - 1, :last_len * 4].copy_(quot[(W - 1) * step:(expected_degree + 1) * 4])
        del quot
        randomizers = [H.fr_rand(cp, rng) for _ in range(W - 1)]     # drawn on every rank, applied to instance 0's share
        if me == 0:
            firsts = _read_fr(cp, split[:, :4].contiguous())
            last_randomizer = 0
            for i in range(W - 1):
                firsts[i] = (firsts[i] - last_randomizer) % r
                split[i, step:step + 4] = _dev_fr(mont([randomizers[i]]))
                last_randomizer = randomizers[i]
            firsts[W - 1] = (firsts[W - 1] - last_randomizer) % r
            split[:, :4] = _dev_fr(mont(firsts)).view(W, 4)
        split_lens = [n + 3] * (W - 1) + [last_len]
        split_quot_poly_comms = aggregated_commit([split[j] for j in range(W)], split_lens)
        tr.append_commitments(b"quot_poly_comms", split_quot_poly_comms)

        # ---- Round 4
        zeta = tr.get_challenge(b"zeta")
        zeta_omega = zeta * group_gen % r
        zm, zom = mont([zeta])[0], mont([zeta_omega])[0]
        evals_dev = torch.zeros((2 * W) * 4, dtype=torch.int64, device="cuda")
        at_zeta = [(wire_polys[j], n + 2) for j in range(W)] + [(pk.sigmas[j], n) for j in range(W - 1)]
        L.poly_eval_batch_device(c, [p.data_ptr() for p, _ in at_zeta], [k for _, k in at_zeta], zm,
                                 evals_dev.data_ptr(), stream=st)
        L.poly_eval_batch_device(c, [perm_poly.data_ptr()], [n + 3], zom, evals_dev.data_ptr() + (2 * W - 1) * 32,
                                 stream=st)
        poly_evals_vec = []
        for ev in gather_ints(_read_fr(cp, evals_dev)):
            pe = ProofEvaluations(ev[:W], ev[W:2 * W - 1], ev[2 * W - 1])
            tr.append_field_elems(b"wire_evals", pe.wires_evals)
            tr.append_field_elems(b"wire_sigma_evals", pe.wire_sigma_evals)
            tr.append_field_elems(b"perm_next_eval", [pe.perm_next_eval])
            poly_evals_vec.append(pe)
        # this rank's share of the linearisation polynomial: quotient part over its split share + its instance's part
        vanish_eval = (pow(zeta, n, r) - 1) % r
        zeta_to_n_plus_2 = (vanish_eval + 1) * zeta % r * zeta % r
        lagrange_1_eval = vanish_eval * pow(n * (zeta - 1) % r, -1, r) % r
        terms = []
        coeff = 1
        for i in range(W):
            if i:
                coeff = coeff * zeta_to_n_plus_2 % r
            terms.append((-vanish_eval * coeff % r, split[i], split_lens[i]))
        pe = poly_evals_vec[me]
        w = pe.wires_evals
        q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r), pow(w[2], 5, r),
                     pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
        terms += [(s % r * alpha_base % r, pk.selectors[j], n) for j, s in enumerate(q_scalars)]
        cz = alpha
        for j in range(W):
            cz = cz * (w[j] + beta * vk.k[j] % r * zeta + gamma) % r
        cz = (cz + alpha * alpha % r * lagrange_1_eval) % r
        terms.append((cz * alpha_base % r, perm_poly, n + 3))
        csig = alpha * beta % r * pe.perm_next_eval % r
        for j in range(W - 1):
            csig = csig * (w[j] + beta * pe.wire_sigma_evals[j] + gamma) % r
        terms.append((-csig * alpha_base % r, pk.sigmas[W - 1], n))
        lin_share = lincomb(terms, n + 3)

        # ---- Round 5: this rank's share of the two batched opening polynomials; division by (X - z) is linear too
        v = tr.get_challenge(b"v")
        cur = pow(v, 1 + me * (2 * W - 1), r)
        open_terms = [(1, lin_share, n + 3)]
        for p, k in at_zeta:
            open_terms.append((cur, p, k))
            cur = cur * v % r
        open_share = lincomb(open_terms, n + 3)
        shifted_share = lincomb([(pow(v, me, r), perm_poly, n + 3)], n + 3)
        wpoly = torch.zeros(2, l4, dtype=torch.int64, device="cuda")
        L.poly_divide_linear_device(c, open_share.data_ptr(), n + 3, zm, wpoly[0].data_ptr(), stream=st)
        L.poly_divide_linear_device(c, shifted_share.data_ptr(), n + 3, zom, wpoly[1].data_ptr(), stream=st)
        opening_proof, shifted_opening_proof = aggregated_commit([wpoly[0][4:], wpoly[1][4:]], [n + 2, n + 2])
        return BatchProof(c, wires_poly_comms_vec, prod_perm_poly_comms_vec, poly_evals_vec, split_quot_poly_comms,
                          opening_proof, shifted_opening_proof)
