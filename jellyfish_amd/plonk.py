"""TurboPlonk / UltraPlonk (Plookup) `PlonkKzgSnark::{preprocess, prove, batch_prove}` with every polynomial operation on
the device.

Host-side mirror of the reference interface (same names, argument meaning and error behaviour):
  * `PlonkCircuit`                       relation/src/constraint_system.rs:347-470, 558-790, 831-884, 943-981, 1116-1187
  * `PlonkKzgSnark.universal_setup_for_testing / preprocess / prove`
                                          plonk/src/proof_system/snark.rs:447-488, 492-580, 587-613 (-> 163-430)
  * prover rounds                         plonk/src/proof_system/prover.rs:77-91, 128-142, 183-229, 299-413, 457-488,
                                          491-661, 894-1015
  * UltraPlonk / Plookup                  circuit: relation/src/constraint_system.rs:304-312, 445-481, 604-648, 845-864,
                                          1171-1179, 1460-1688, gadgets/ultraplonk/lookup_table.rs:21-71;
                                          rounds 1.5 / 2.5 / 4.5: snark.rs:254-318, 346-362, prover.rs:99-124, 147-178,
                                          233-296, 421-453, 766-882, 1017-1087; PlookupProof structs.rs:208-221, 460-536
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

import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import hostglue as H
from . import lib as L
from .transcript import MerlinTranscript

GATE_WIDTH = 4
N_TURBO_PLONK_SELECTORS = 13
# UltraPlonk wire roles (relation/src/constraint_system.rs:304-312)
RANGE_WIRE_ID = 5
LOOKUP_KEY_WIRE_ID, LOOKUP_VAL_1_WIRE_ID, LOOKUP_VAL_2_WIRE_ID = 0, 1, 2
TABLE_VAL_1_WIRE_ID, TABLE_VAL_2_WIRE_ID = 3, 4
# an UltraPlonk gate kind = 13 selectors + (q_lookup, q_dom_sep, table_key, table_dom_sep)   (gates/mod.rs:55-69)
_NO_LOOKUP = (0, 0, 0, 0)
_Q_LOOKUP, _Q_DOM_SEP, _TABLE_KEY, _TABLE_DOM_SEP = 13, 14, 15, 16


class PlonkError(RuntimeError):
    pass


PROFILE = None   # set to a dict to collect a per-round wall-clock breakdown of prove() (adds device syncs)
# JFHIP_MARKERS=1: a roctx mark at the end of every step of prove() (with a device sync, so that the kernels of a step lie
# between its mark and the previous one): `rocprofv3 --kernel-trace --marker-trace` then segments a proof by round
MARKERS = os.environ.get("JFHIP_MARKERS", "0") not in ("", "0")


def _sel(q_lc=(0, 0, 0, 0), q_mul=(0, 0), q_hash=(0, 0, 0, 0), q_o=0, q_c=0, q_ecc=0):
    return tuple(q_lc) + tuple(q_mul) + tuple(q_hash) + (q_o, q_c, q_ecc)


class PlonkCircuit:
    """TurboPlonk / UltraPlonk constraint system (host side).  Variables 0 / 1 hold the constants 0 / 1."""

    def __init__(self, curve: int, range_bit_len: int | None = None):
        self.cp = H.CURVES[curve]
        self.r = self.cp.r
        self.ultra = range_bit_len is not None
        self._range_bit_len = range_bit_len
        self.num_wire_types = GATE_WIDTH + 1 + (1 if self.ultra else 0)
        self.witness: list[int] = [0, 1]
        self._gates: list[tuple] = []
        self._wires: list[list[int]] = [[] for _ in range(self.num_wire_types)]
        self.pub_input_gate_ids: list[int] = []
        self._table_gate_ids: list[tuple] = []
        self._num_table_elems = 0
        self._n = 0
        self.enforce_constant(0, 0)
        self.enforce_constant(1, 1)

    @classmethod
    def new_turbo_plonk(cls, curve: int) -> "PlonkCircuit":
        return cls(curve)

    @classmethod
    def new_ultra_plonk(cls, curve: int, range_bit_len: int) -> "PlonkCircuit":
        """constraint_system.rs:445-448"""
        return cls(curve, range_bit_len)

    def support_lookup(self) -> bool:
        return self.ultra

    def range_bit_len(self) -> int:
        if not self.ultra:
            raise PlonkError("call range_bit_len() with non-ultraplonk circuit")
        return self._range_bit_len

    def range_size(self) -> int:
        return 1 << self.range_bit_len()

    # ---- builder ---------------------------------------------------------------------------------
    def _check_var(self, *vs):
        for v in vs:
            if not 0 <= v < len(self.witness):
                raise PlonkError("VarIndexOutOfBound(%d, %d)" % (v, len(self.witness)))

    def _insert_gate(self, wire_vars, sel, lk=_NO_LOOKUP):
        if self._n:
            raise PlonkError("ModifyFinalizedCircuit")
        for w, v in zip(self._wires, wire_vars):      # the GATE_WIDTH + 1 gate wires; the range wire has its own list
            w.append(v)
        self._gates.append(sel + lk if self.ultra else sel)

    def num_gates(self) -> int:
        return len(self._gates) if not self._n else int(self.gate_kind.size)

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

    def add_range_check_variable(self, var: int):
        """the variable joins the range wire and is looked up in [0, range_size) (constraint_system.rs:475-481)"""
        if not self.ultra:
            raise PlonkError("WrongPlonkType")
        if self._n:
            raise PlonkError("ModifyFinalizedCircuit")
        self._check_var(var)
        self._wires[RANGE_WIRE_ID].append(var)

    def create_table_and_lookup_variables(self, lookup_vars, table_vars):
        """a key-value table (keys 0 .. len-1, values = pairs of variables) and (key, val0, val1) variable triples to look
        up in it: one LookupGate per row (relation/src/gadgets/ultraplonk/lookup_table.rs:21-71)"""
        if not self.ultra:
            raise PlonkError("WrongPlonkType")
        for t in list(lookup_vars) + list(table_vars):
            self._check_var(*t)
        cnt = max(len(lookup_vars), len(table_vars))
        self._table_gate_ids.append((self.num_gates(), cnt))
        table_ctr = len(self._table_gate_ids)
        for i in range(cnt):
            q_dom_sep, (key, val0, val1) = (table_ctr, lookup_vars[i]) if i < len(lookup_vars) else (0, (0, 0, 0))
            table_dom_sep, table_key, (tv0, tv1) = (table_ctr, i, table_vars[i]) if i < len(table_vars) else (0, 0, (0, 0))
            self._insert_gate((key, val0, val1, tv0, tv1), _sel(), lk=(1, q_dom_sep, table_key, table_dom_sep))
        self._num_table_elems += cnt

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
        """pad to a power of two, io gates to the front, table lookup gates to the rear, wire permutation
        (constraint_system.rs:1167-1187); the field-valued tables (k * g^j, sigma values, selector polynomials) are built
        on the device by preprocess"""
        if self._n:
            return
        slots = self.num_gates()
        if self.ultra:   # range gates and lookup gates need separate slots (:1171-1179)
            slots = max(slots, max(self.range_size(), len(self._wires[RANGE_WIRE_ID])) + self._num_table_elems + 1)
        n = 1
        while n < slots:
            n <<= 1
        if n.bit_length() - 1 > self.cp.two_adicity:
            raise PlonkError("DomainCreationError")
        W = self.num_wire_types
        wv = np.zeros((W, n), dtype=np.int64)
        for i in range(W):
            wv[i, :len(self._wires[i])] = self._wires[i]
        gates = self._gates + [_sel() + (_NO_LOOKUP if self.ultra else ())] * (n - len(self._gates))
        uniq: dict[tuple, int] = {}
        gate_kind = np.fromiter((uniq.setdefault(g, len(uniq)) for g in gates), dtype=np.int64, count=n)
        GW = GATE_WIDTH + 1   # rearrange_gates swaps the gate wires only; the range wire stays (:836-840, 855-859)
        for gate_id in range(len(self.pub_input_gate_ids)):
            io = self.pub_input_gate_ids[gate_id]
            if io > gate_id:
                gate_kind[[gate_id, io]] = gate_kind[[io, gate_id]]
                wv[:GW, [gate_id, io]] = wv[:GW, [io, gate_id]]
                self.pub_input_gate_ids[gate_id] = gate_id
        if self.ultra:
            # table lookup gates to the rear, never the very last slot, relative order kept (:845-864)
            cur = n - 2
            for table_gate_id, table_size in reversed(self._table_gate_ids):
                for gate_id in range(table_gate_id + table_size - 1, table_gate_id - 1, -1):
                    if gate_id < cur:
                        gate_kind[[gate_id, cur]] = gate_kind[[cur, gate_id]]
                        wv[:GW, [gate_id, cur]] = wv[:GW, [cur, gate_id]]
                        cur -= 1
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
        """gate equation on every row, range gates and key-value lookups (constraint_system.rs:585-650, 894-938); host
        integers, for tests / small circuits"""
        if len(pub_input) != self.num_inputs():
            raise PlonkError("PubInputLenMismatch(%d, %d)" % (len(pub_input), self.num_inputs()))
        r = self.r
        pis = dict(zip(self.pub_input_gate_ids, pub_input))
        wit = lambda i, g: self.witness[int(self.wire_variables[i][g])]   # noqa: E731
        for gid in range(self._n):
            s = self.gate_kinds[int(self.gate_kind[gid])]
            w = [wit(i, gid) for i in range(5)]
            lhs = (pis.get(gid, 0) + sum(s[i] * w[i] for i in range(4)) + s[4] * w[0] * w[1] + s[5] * w[2] * w[3]
                   + sum(s[6 + i] * pow(w[i], 5, r) for i in range(4)) + s[11] + s[12] * w[0] * w[1] * w[2] * w[3] * w[4])
            if (lhs - s[10] * w[4]) % r:
                raise PlonkError("GateCheckFailure(%d)" % gid)
        if not self.ultra:
            return
        for gid in range(self._n):
            if wit(RANGE_WIRE_ID, gid) >= self.range_size():
                raise PlonkError("GateCheckFailure(%d): Range gate failed" % gid)
        table = {(0, 0, 0, 0)}
        kinds = [self.gate_kinds[int(k)] for k in self.gate_kind]
        for gid, s in enumerate(kinds):
            if s[_Q_LOOKUP]:
                table.add((s[_TABLE_DOM_SEP], s[_TABLE_KEY], wit(TABLE_VAL_1_WIRE_ID, gid), wit(TABLE_VAL_2_WIRE_ID, gid)))
        for gid, s in enumerate(kinds):
            if s[_Q_LOOKUP] and (s[_Q_DOM_SEP], wit(LOOKUP_KEY_WIRE_ID, gid), wit(LOOKUP_VAL_1_WIRE_ID, gid),
                                 wit(LOOKUP_VAL_2_WIRE_ID, gid)) not in table:
                raise PlonkError("GateCheckFailure(%d): Lookup gate failed" % gid)


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
    # PlookupVerifyingKey (structs.rs:736-753): range_table_comm, key_table_comm, table_dom_sep_comm, q_dom_sep_comm
    plookup_vk: dict | None = None

    def as_dict(self) -> dict:
        d = dict(domain_size=self.domain_size, num_inputs=self.num_inputs, sigma_comms=self.sigma_comms,
                 selector_comms=self.selector_comms, k=self.k, g=self.g, h=self.h, beta_h=self.beta_h)
        if self.plookup_vk is not None:
            d["plookup_vk"] = dict(self.plookup_vk)
        return d


@dataclass
class ProvingKey:
    curve: int
    sigmas: torch.Tensor          # (W, n*4) coefficients
    selectors: torch.Tensor       # (13 or 14, n*4) coefficients; UltraPlonk: q_lookup last
    commit_key: "L.Srs"
    vk: VerifyingKey
    # evaluation-form tables the grand product reads (the reference recomputes them from the circuit per proof)
    id_perm_vals: torch.Tensor = field(repr=False, default=None)
    sigma_vals: torch.Tensor = field(repr=False, default=None)
    wire_variables: torch.Tensor = field(repr=False, default=None)
    # PlookupProvingKey (structs.rs:558-573), UltraPlonk only: coefficients of range_table, key_table, table_dom_sep,
    # q_dom_sep (4, n*4), and the same four vectors + q_lookup in evaluation form (5, n*4) for the merged-vector kernel
    plookup_polys: torch.Tensor = field(repr=False, default=None)
    plookup_table_vals: torch.Tensor = field(repr=False, default=None)

    @property
    def supports_lookup(self) -> bool:
        return self.plookup_polys is not None

    @property
    def wire_variables32(self) -> torch.Tensor:
        """the wire -> variable table as the uint32 indices jfhip_witness_gather_device takes (built once)"""
        if getattr(self, "_wv32", None) is None:
            self._wv32 = self.wire_variables.to(torch.int32).contiguous()
        return self._wv32
    # optional: the selector + sigma (+ 4 Plookup table) polynomials evaluated on the quotient coset (m = 8n points each),
    # kept in HBM.  The reference recomputes these 18 (24) coset FFTs in every proof (prover.rs:534-567); they depend on the
    # proving key only, and 288 GB of HBM holds them (18 x 256 MiB at n = 2^20).  Same values, so the same proof bytes.
    coset_evals: torch.Tensor = field(repr=False, default=None)
    coset_slice: tuple = (0, 1)   # (rank, world) the resident evaluations were computed for
    # the same, one tensor per slice (g, G) of quotient_slices(): what the one-GPU prover evaluates the quotient on
    coset_evals_slices: dict = field(repr=False, default=None)
    # selector polynomials that are identically zero for this circuit (a gate kind it never uses): their coset
    # evaluations are zero, so prove() fills those rows instead of transforming them (same values, same proof bytes)
    selector_is_zero: tuple = ()
    # optional: the Lagrange-basis form of the commit key for this domain ([L_i(beta)]g, then 3 vanishing-polynomial points;
    # jfhip_srs_lagrange).  With it the round-1 commitments (wire polynomials, Plookup sorted polynomials) are MSMs over the
    # witness VALUES and the blinders instead of over the coefficients the iFFT makes of them: the same group elements --
    # the same proof bytes -- but witness values are often a few bits wide (counters, booleans, range-checked limbs), the
    # bucket sort drops their zero digits, and the MSM costs a fraction of a dense one.  The reference commits the
    # coefficient vectors (prover.rs:77-91, 99-124); a dense witness gains nothing and loses nothing.
    lagrange_key: "L.Srs" = field(repr=False, default=None)

    def domain_size(self) -> int:
        return self.vk.domain_size

    def keep_lagrange_commit_key(self, window_bits: int = 0) -> "ProvingKey":
        """derive the Lagrange-basis commit key once (one group DFT of size n and a second window table; see lagrange_key).
        window_bits: the MSM window of that table (0: the library's choice for dense scalars of this length; a witness of
        small values folds fewer buckets with a narrower window)"""
        if self.lagrange_key is None:
            n = self.vk.domain_size
            wb = window_bits or int(os.environ.get("JF_LAGRANGE_WINDOW_BITS", "0"))
            self.lagrange_key = self.commit_key.lagrange(n.bit_length() - 1, 3, wb)
        return self

    def drop_lagrange_commit_key(self) -> "ProvingKey":
        if self.lagrange_key is not None:
            self.lagrange_key.free()
            self.lagrange_key = None
        return self

    def coset_evals_bytes(self) -> int:
        """HBM held by keep_coset_evals()"""
        ts = list(self.coset_evals_slices.values()) if self.coset_evals_slices else []
        if self.coset_evals is not None:
            ts.append(self.coset_evals)
        return sum(t.numel() * 8 for t in ts)

    def quot_domain_size(self) -> int:
        n, W = self.vk.domain_size, self.sigmas.shape[0]
        m = 1
        while m < n * domain_size_ratio(n, W):
            m <<= 1
        return m

    def keep_coset_evals(self, stream=None, rank: int = 0, world: int = 1) -> "ProvingKey":
        """evaluate selectors, sigmas (and the Plookup tables) on the quotient coset once and keep them resident (see
        coset_evals).  world > 1: only rank's slice of the coset (prove(sharded=True) with coset sharding); world == 1:
        the slices of quotient_slices(), which is what the one-GPU prover evaluates the quotient on"""
        cp = H.CURVES[self.curve]
        n, m = self.vk.domain_size, self.quot_domain_size()
        if not coset_sharding_applies(n, m, world):
            rank, world = 0, 1
        nsel, W = self.selectors.shape[0], self.sigmas.shape[0]
        ntab = 4 if self.supports_lookup else 0
        st = _stream(stream)

        def evals_on(g, G):
            ml = m // G
            with _torch_on(st):
                ev = torch.zeros(nsel + W + ntab, ml * 4, dtype=torch.int64, device="cuda")
                rows = [self.selectors, self.sigmas] + ([self.plookup_polys] if ntab else [])
                at = 0
                for polys in rows:
                    k = polys.shape[0]
                    if ml >= n:
                        ev[at:at + k, :n * 4].copy_(polys)
                    else:                 # never: selectors have n coefficients and the smallest slice n points
                        raise PlonkError("slice smaller than the domain")
                    at += k
                off = H.fr_to_mont(cp, [slice_offset(cp, m, g)])[0]
                L.ntt_device(self.curve, ev.data_ptr(), ml.bit_length() - 1, nsel + W + ntab, inverse=False,
                             coset_mont=off, in_len=n, stream=st)
            return ev

        slices = quotient_slices(n, m, W * (n + 1) + 3) if (world == 1 and QUOTIENT_SLICES) else [(rank, world)]
        if len(slices) > 1:
            self.coset_evals, self.coset_slice = None, (0, 1)
            self.coset_evals_slices = {sl: evals_on(*sl) for sl in slices}
        else:
            self.coset_evals_slices = None
            self.coset_evals = evals_on(rank, world)
            self.coset_slice = (rank, world)
        return self


@dataclass
class PlookupEvaluations:
    """plonk/src/proof_system/structs.rs:460-536"""
    range_table_eval: int
    key_table_eval: int
    table_dom_sep_eval: int
    q_dom_sep_eval: int
    h_1_eval: int
    q_lookup_eval: int
    prod_next_eval: int
    range_table_next_eval: int
    key_table_next_eval: int
    table_dom_sep_next_eval: int
    h_1_next_eval: int
    h_2_next_eval: int
    q_lookup_next_eval: int
    w_3_next_eval: int
    w_4_next_eval: int

    FIELDS = ("range_table_eval", "key_table_eval", "table_dom_sep_eval", "q_dom_sep_eval", "h_1_eval", "q_lookup_eval",
              "prod_next_eval", "range_table_next_eval", "key_table_next_eval", "table_dom_sep_next_eval", "h_1_next_eval",
              "h_2_next_eval", "q_lookup_next_eval", "w_3_next_eval", "w_4_next_eval")   # declaration = serialisation order

    def evals_vec(self) -> list:
        """the order at zeta in the transcript and the opening (structs.rs:511-520)"""
        return [self.range_table_eval, self.key_table_eval, self.h_1_eval, self.q_lookup_eval, self.table_dom_sep_eval,
                self.q_dom_sep_eval]

    def next_evals_vec(self) -> list:
        """the order at zeta * g (structs.rs:523-535)"""
        return [self.prod_next_eval, self.range_table_next_eval, self.key_table_next_eval, self.h_1_next_eval,
                self.h_2_next_eval, self.q_lookup_next_eval, self.w_3_next_eval, self.w_4_next_eval,
                self.table_dom_sep_next_eval]

    @classmethod
    def from_vecs(cls, at_zeta, at_next) -> "PlookupEvaluations":
        rt, kt, h1, ql, tds, qds = at_zeta
        pn, rtn, ktn, h1n, h2n, qln, w3n, w4n, tdsn = at_next
        return cls(rt, kt, tds, qds, h1, ql, pn, rtn, ktn, tdsn, h1n, h2n, qln, w3n, w4n)

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k in self.FIELDS}


@dataclass
class PlookupProof:
    """plonk/src/proof_system/structs.rs:208-221"""
    h_poly_comms: list
    prod_lookup_poly_comm: tuple
    poly_evals: PlookupEvaluations

    def as_dict(self) -> dict:
        return dict(h_poly_comms=self.h_poly_comms, prod_lookup_poly_comm=self.prod_lookup_poly_comm,
                    poly_evals=self.poly_evals.as_dict())

    def serialize_into(self, cp, out: bytearray):
        out += len(self.h_poly_comms).to_bytes(8, "little")
        for c in self.h_poly_comms:
            out += H.ser_g1(cp, c)
        out += H.ser_g1(cp, self.prod_lookup_poly_comm)
        for k in PlookupEvaluations.FIELDS:
            out += H.ser_fr(cp, getattr(self.poly_evals, k))


def _ser_option_plookup(cp, out: bytearray, pp):
    """ark-serialize Option<PlookupProof>: a flag byte, then the value"""
    if pp is None:
        out.append(0)
    else:
        out.append(1)
        pp.serialize_into(cp, out)


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
    plookup_proof: PlookupProof | None = None

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k in ("wires_poly_comms", "prod_perm_poly_comm", "split_quot_poly_comms",
                                           "opening_proof", "shifted_opening_proof", "wires_evals",
                                           "wire_sigma_evals", "perm_next_eval")}
        if self.plookup_proof is not None:
            d["plookup_proof"] = self.plookup_proof.as_dict()
        return d

    def serialize_compressed(self) -> bytes:
        """ark-serialize layout of `Proof` (plonk/src/proof_system/structs.rs:54-76)"""
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
        _ser_option_plookup(cp, out, self.plookup_proof)
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
    """plonk/src/proof_system/structs.rs:227-249"""
    curve: int
    wires_poly_comms_vec: list
    prod_perm_poly_comms_vec: list
    poly_evals_vec: list
    split_quot_poly_comms: list
    opening_proof: tuple
    shifted_opening_proof: tuple
    plookup_proofs_vec: list | None = None      # None = every instance is TurboPlonk ([None; k])

    def __len__(self) -> int:
        return len(self.prod_perm_poly_comms_vec)

    def as_dict(self) -> dict:
        d = dict(wires_poly_comms_vec=self.wires_poly_comms_vec, prod_perm_poly_comms_vec=self.prod_perm_poly_comms_vec,
                 poly_evals_vec=[e.as_dict() for e in self.poly_evals_vec],
                 split_quot_poly_comms=self.split_quot_poly_comms, opening_proof=self.opening_proof,
                 shifted_opening_proof=self.shifted_opening_proof)
        if self.plookup_proofs_vec is not None and any(p is not None for p in self.plookup_proofs_vec):
            d["plookup_proofs_vec"] = [None if p is None else p.as_dict() for p in self.plookup_proofs_vec]
        return d

    def to_proof(self) -> "Proof":
        """the single-instance `Proof` (inverse of `impl From<Proof> for BatchProof`, structs.rs:275-287)"""
        if len(self) != 1:
            raise PlonkError("a BatchProof over %d instances is not a Proof" % len(self))
        e = self.poly_evals_vec[0]
        return Proof(self.curve, self.wires_poly_comms_vec[0], self.prod_perm_poly_comms_vec[0],
                     self.split_quot_poly_comms, self.opening_proof, self.shifted_opening_proof, e.wires_evals,
                     e.wire_sigma_evals, e.perm_next_eval,
                     self.plookup_proofs_vec[0] if self.plookup_proofs_vec is not None else None)

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
        for pp in self.plookup_proofs_vec if self.plookup_proofs_vec is not None else [None] * k:
            _ser_option_plookup(cp, out, pp)
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


class _torch_on:
    """run torch ops on the stream the library calls go to.  With the default argument (stream=None) that is torch's
    current stream already; a caller-supplied jfhip stream (e.g. L.Stream(), a non-blocking HIP stream) is entered as a
    torch ExternalStream, ordered after the work already queued on torch's current stream on entry and before what
    follows on exit -- so torch copies / allocations and libjfhip kernels never run unordered."""

    def __init__(self, st):
        raw = st.raw
        self.ext = None
        if raw != torch.cuda.current_stream().cuda_stream:
            self.ext = torch.cuda.ExternalStream(raw)

    def __enter__(self):
        if self.ext is not None:
            self.outer = torch.cuda.current_stream()
            self.ext.wait_stream(self.outer)
            self.ctx = torch.cuda.stream(self.ext)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ext is not None:
            self.ctx.__exit__(*exc)
            self.outer.wait_stream(self.ext)
        return False


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


QUOTIENT_SLICES = True   # False: evaluate the quotient on all m = 8 n points like the reference (bench comparison; same results)


def quotient_slices(n: int, m: int, ncoeffs: int) -> list:
    """disjoint slices (g, G) -- the points i = g (mod G) of the quotient coset, m / G of them -- with at least ncoeffs
    points in total: i even (4 n points), i = 1 (mod 4) (2 n), i = 3 (mod 8) (n) as far as needed; [(0, 1)] when that does
    not apply (ratio other than 8, tiny domains)"""
    if m != 8 * n or n < 16:
        return [(0, 1)]
    out, have = [], 0
    for g, G in ((0, 2), (1, 4), (3, 8)):
        out.append((g, G))
        have += m // G
        if have >= ncoeffs:
            return out
    return [(0, 1)]


def unslice_matrix(cp, m: int, slices) -> tuple:
    """t(X) = sum_k X^(k s) t_k in chunks of s = the smallest slice size.  Slice j (q_j s points, offset o_j) holds
    t mod (X^(q_j s) - c_j), c_j = o_j^(q_j s): its chunk e is sum over k = e (mod q_j) of c_j^((k - e) / q_j) t_k.  Returns
    (s, inverse of that square matrix over Fr): t_k = sum_(j, e) inv[k][(j, e)] * (chunk e of slice j)."""
    r = cp.r
    sizes = [m // G for _, G in slices]
    s = min(sizes)
    qs = [sz // s for sz in sizes]
    K = sum(qs)
    rows = []
    for (g, G), q in zip(slices, qs):
        cj = pow(slice_offset(cp, m, g), q * s, r)
        for e in range(q):
            rows.append([pow(cj, (k - e) // q, r) if k % q == e else 0 for k in range(K)])
    # Gauss-Jordan over Fr (K <= 7)
    a = [row[:] + [1 if i == j else 0 for j in range(K)] for i, row in enumerate(rows)]
    for col in range(K):
        piv = next(i for i in range(col, K) if a[i][col])
        a[col], a[piv] = a[piv], a[col]
        inv = pow(a[col][col], -1, r)
        a[col] = [x * inv % r for x in a[col]]
        for i in range(K):
            if i != col and a[i][col]:
                f = a[i][col]
                a[i] = [(x - f * y) % r for x, y in zip(a[i], a[col])]
    return s, [row[K:] for row in a]


def unslice_quotient(cp, curve: int, parts, slices, m: int, stream) -> torch.Tensor:
    """quotient coefficients (m entries, zero beyond the slices' total) from the slice polynomials: K lincombs of <= K terms"""
    s, inv = unslice_matrix(cp, m, slices)
    ptrs = []
    for part in parts:
        ptrs += [part.data_ptr() + e * s * 32 for e in range(part.shape[0] // (4 * s))]
    K = len(ptrs)
    out = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
    for k in range(K):
        terms = [(ptrs[j], inv[k][j]) for j in range(K) if inv[k][j]]
        L.poly_lincomb_device(curve, [t[0] for t in terms], [s] * len(terms), H.fr_to_mont(cp, [t[1] for t in terms]),
                              out.data_ptr() + k * s * 32, s, stream=stream)
    return out


def _sorted_lookup_vector(table: torch.Tensor, lookups: torch.Tensor, stream=None) -> torch.Tensor:
    """compute_lookup_sorted_vec_polynomials (relation/src/constraint_system.rs:1597-1620) before its two iffts, on
    device rows of 4 x int64 (one field element each; equal elements have equal Montgomery limbs): every table element,
    in table order, followed by as many copies as it is looked up -- counted at its FIRST occurrence in the table, as
    the reference's HashMap entry is removed once used.  table: (n, 4), lookups: (n - 1, 4) -> (2n - 1, 4).
    jfhip_plonk_lookup_sorted_device: a hash table of first positions, wave-aggregated counts, a scan, an expansion."""
    n, k = table.shape[0], lookups.shape[0]
    table, lookups = table.contiguous(), lookups.contiguous()
    out = torch.empty(n + k, 4, dtype=torch.int64, device=table.device)
    try:
        L.plonk_lookup_sorted_device(table.data_ptr(), n, lookups.data_ptr(), k, out.data_ptr(), stream=stream)
    except L.JfHipError as e:
        if "sorted vector has wrong length" in str(e):
            raise PlonkError("The sorted vector has wrong length, some lookup variables might be outside the table") from e
        raise
    return out


class _Placement:
    """which rank proves which instance of a batch.  owners = None: every instance is on this GPU.  Otherwise
    owners[i] is the rank (of `group`) that holds instance i's circuit and proving key; per-instance commitments and
    evaluations travel as small python objects through all_gather_object, aggregated commitments as group sums of
    un-normalised per-rank partial sums (one all_gather of 128 / 192 bytes per commitment)."""

    def __init__(self, owners, group):
        self.owners = owners
        self.group = group
        self.dist = owners is not None
        if self.dist:
            import torch.distributed as tdist

            self.tdist = tdist
            self.world = tdist.get_world_size(group)
            self.me = tdist.get_rank(group)
        else:
            self.world, self.me = 1, 0

    def local(self, i: int) -> bool:
        return not self.dist or self.owners[i] == self.me

    def gather(self, mine: dict, count: int) -> list:
        """{instance: object} of this rank's instances -> the list over all instances"""
        if not self.dist:
            return [mine.get(i) for i in range(count)]
        parts = [None] * self.world
        self.tdist.all_gather_object(parts, mine, group=self.group)
        merged = {}
        for d in parts:
            merged.update(d)
        return [merged.get(i) for i in range(count)]


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
        """selector / sigma (/ Plookup table) polynomials and their commitments (snark.rs:492-580), all on the device"""
        cp = circuit.cp
        c = cp.curve_id
        n, log_n, W = circuit.eval_domain_size(), circuit.log_n, circuit.num_wire_types
        if srs.max_degree() < circuit.srs_size():
            raise PlonkError("IndexTooLarge")
        ultra = circuit.support_lookup()
        NSEL = N_TURBO_PLONK_SELECTORS + (1 if ultra else 0)
        st = _stream(stream)
        with _torch_on(st):
            # selector values -> coefficients (compute_selector_polynomials, constraint_system.rs:1359-1372)
            table = np.stack([H.fr_to_mont(cp, k) for k in circuit.gate_kinds])             # (kinds, 13 or 17, 4)
            by_gate = table[circuit.gate_kind]                                              # (n, 13 or 17, 4)
            sel_vals = np.ascontiguousarray(by_gate[:, :NSEL].transpose(1, 0, 2))           # (NSEL, n, 4)
            selectors = _dev_fr(sel_vals).view(NSEL, n * 4)
            L.ntt_device(c, selectors.data_ptr(), log_n, NSEL, inverse=True, stream=st)
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
            polys = [selectors[j] for j in range(NSEL)] + [sigmas[j] for j in range(W)]
            plookup_polys = table_vals = None
            if ultra:
                # compute_{range_table, key_table, table_dom_sep, q_dom_sep}_polynomial (constraint_system.rs:1460-1493)
                if n < circuit.range_size():
                    raise PlonkError("Domain size %d < range size %d" % (n, circuit.range_size()))
                rng_tab = np.zeros((n, 4), dtype=np.uint64)
                rng_tab[:circuit.range_size()] = H.fr_to_mont(cp, range(circuit.range_size()))
                rows = [rng_tab, by_gate[:, _TABLE_KEY], by_gate[:, _TABLE_DOM_SEP], by_gate[:, _Q_DOM_SEP],
                        by_gate[:, _Q_LOOKUP]]
                table_vals = _dev_fr(np.ascontiguousarray(np.stack(rows))).view(5, n * 4)
                plookup_polys = table_vals[:4].clone()
                L.ntt_device(c, plookup_polys.data_ptr(), log_n, 4, inverse=True, stream=st)
                polys += [plookup_polys[j] for j in range(4)]
            comms = _commit_many(srs.powers_of_g, cp, polys, [n] * len(polys), st)
            vk = VerifyingKey(n, circuit.num_inputs(), comms[NSEL:NSEL + W], comms[:NSEL], list(circuit.k), srs.g, srs.h,
                              srs.beta_h)
            if ultra:
                pc = comms[NSEL + W:]
                vk.plookup_vk = dict(range_table_comm=pc[0], key_table_comm=pc[1], table_dom_sep_comm=pc[2],
                                     q_dom_sep_comm=pc[3])
            pk = ProvingKey(c, sigmas, selectors, srs.powers_of_g, vk, idp, sigma_vals,
                            torch.from_numpy(circuit.wire_variables.reshape(-1)).cuda(), plookup_polys, table_vals)
            pk.selector_is_zero = tuple(bool(z) for z in (~torch.any(selectors != 0, dim=1)).cpu().tolist())
        return pk, vk

    @staticmethod
    def prove(rng, circuit: PlonkCircuit, prove_key: ProvingKey, extra_transcript_init_msg: bytes | None = None,
              stream=None, witness_mont: np.ndarray | None = None, sharded: bool = False, group=None) -> Proof:
        """One TurboPlonk / UltraPlonk proof (snark.rs:587-613 -> batch_prove_internal :163-430 with one instance).
        witness_mont: optional pre-converted witness table ((num_vars, 4) uint64 Montgomery), else converted here.
        sharded: every rank of `group` calls prove with the same inputs (same RNG state); the commitments' point
        ranges are partitioned over the ranks (see _Committer) and the quotient coset is split into slices."""
        bp = PlonkKzgSnark._batch_prove_internal(rng, [circuit], [prove_key], extra_transcript_init_msg, stream,
                                                 None if witness_mont is None else [witness_mont], sharded, group)
        return bp.to_proof()

    @staticmethod
    def batch_prove(rng, circuits, prove_keys, extra_transcript_init_msg: bytes | None = None, stream=None,
                    witness_monts=None, sharded: bool = False, group=None) -> "BatchProof":
        """k instances -> ONE aggregated BatchProof (snark.rs:61-74 -> :163-430): per round the instances run in order
        on the one RNG and the one transcript, the quotient is sum_i alpha_base_i t_i (prover.rs:519-656) and the two
        opening proofs fold every instance's polynomials (prover.rs:356-413)."""
        return PlonkKzgSnark._batch_prove_internal(rng, list(circuits), list(prove_keys), extra_transcript_init_msg,
                                                   stream, witness_monts, sharded, group)

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
        * everything that mixes instances -- the quotient sum_i alpha_base_i t_i, its split, the linearisation and the
          two opening polynomials (prover.rs:644-655, 356-413) -- is linear in the instances: each rank keeps its own
          contribution, commits it without normalising (jfhip_msm_partial_device) and the aggregated commitment is
          the group sum of the gathered partial sums.  No polynomial crosses xGMI."""
        import torch.distributed as tdist

        G = tdist.get_world_size(group)
        me = tdist.get_rank(group)
        circuits = [circuit if i == me else None for i in range(G)]
        pks = [prove_key if i == me else None for i in range(G)]
        wms = None if witness_mont is None else [witness_mont if i == me else None for i in range(G)]
        return PlonkKzgSnark._batch_prove_internal(rng, circuits, pks, extra_transcript_init_msg, stream, wms, False,
                                                   group, owners=list(range(G)))

    @staticmethod
    def batch_prove_placed(rng, circuits, prove_keys, owners, extra_transcript_init_msg: bytes | None = None, stream=None,
                           witness_monts=None, group=None) -> "BatchProof":
        """batch_prove with instance i proved on rank owners[i] (every rank owns at least one instance; entries of
        circuits / prove_keys for instances of other ranks are ignored and may be None)"""
        return PlonkKzgSnark._batch_prove_internal(rng, list(circuits), list(prove_keys), extra_transcript_init_msg,
                                                   stream, witness_monts, False, group, owners=list(owners))

    @staticmethod
    def _batch_prove_internal(rng, circuits, prove_keys, extra_transcript_init_msg, stream, witness_monts, sharded,
                              group, owners=None) -> "BatchProof":
        if not circuits:
            raise PlonkError("zero number of circuits/proving keys")
        if len(circuits) != len(prove_keys):
            raise PlonkError("the number of circuits %d != the number of proving keys %d"
                             % (len(circuits), len(prove_keys)))
        if owners is not None and (sharded or len(owners) != len(circuits)):
            raise PlonkError("instance placement needs one owner per instance and excludes point-range sharding")
        st = _stream(stream)
        with _torch_on(st):
            return PlonkKzgSnark._prove_rounds(rng, circuits, prove_keys, extra_transcript_init_msg, st, witness_monts,
                                               sharded, group, _Placement(owners, group))

    @staticmethod
    def _prove_rounds(rng, circuits, prove_keys, extra_transcript_init_msg, st, witness_monts, sharded, group,
                      place) -> "BatchProof":
        K = len(circuits)
        mine = [i for i in range(K) if place.local(i)]
        # ---- parameter checks (snark.rs:181-228) on this rank's instances, BEFORE any collective; with placement the
        # verdicts travel with the verifying keys so that every rank raises the same error
        err = None
        cp = n = W = None
        if not mine:
            err = "rank %d owns no instance of the batch" % place.me
        for i in mine:
            circuit, pk = circuits[i], prove_keys[i]
            if circuit is None or pk is None:
                err = err or "instance %d is placed on this rank but has no circuit / proving key" % i
                continue
            if cp is None:
                cp, n, W = circuit.cp, circuit.eval_domain_size(), circuit.num_wire_types
            if circuit.eval_domain_size() != n:
                err = err or "circuit domain size %d != expected domain size %d" % (circuit.eval_domain_size(), n)
            if pk.domain_size() != n:
                err = err or "proving key domain size %d != expected domain size %d" % (pk.domain_size(), n)
            if circuit.num_inputs() != pk.vk.num_inputs:
                err = err or "circuit.num_inputs %d != prove_key.num_inputs %d" % (circuit.num_inputs(), pk.vk.num_inputs)
            if circuit.support_lookup() != pk.supports_lookup:
                err = err or "Mismatched Plonk types between the proving key and the circuit"
            if circuit.num_wire_types != W or circuit.cp is not cp:
                err = err or "inconsistent plonk circuit types"
            nv = witness_monts[i].shape[0] if witness_monts is not None and witness_monts[i] is not None \
                else len(circuit.witness)
            if nv != circuit.num_vars():
                err = err or "witness table has %d rows, the circuit %d variables" % (nv, circuit.num_vars())
        meta_mine = {}
        if err is None:
            meta_mine = {i: (prove_keys[i].vk, circuits[i].public_input(), circuits[i].support_lookup(), n, W,
                             cp.curve_id) for i in mine}
        if place.dist:
            verdicts = [None] * place.world
            place.tdist.all_gather_object(verdicts, err, group=place.group)
            err = next((e for e in verdicts if e is not None), None)
        if err is not None:
            raise PlonkError(err)
        meta = place.gather(meta_mine, K)
        for (_, _, _, n_i, W_i, c_i) in meta:
            if n_i != n:
                raise PlonkError("circuit domain size %d != expected domain size %d" % (n_i, n))
            if W_i != W or c_i != cp.curve_id:
                raise PlonkError("inconsistent plonk circuit types")
        vks = [x[0] for x in meta]
        pub_inputs = [x[1] for x in meta]
        lookup = [x[2] for x in meta]

        c, r = cp.curve_id, cp.r
        log_n = n.bit_length() - 1
        m = 1
        while m < n * domain_size_ratio(n, W):
            m <<= 1
        log_m = m.bit_length() - 1
        n4 = n * 4
        first = mine[0]
        ck = prove_keys[first].commit_key    # run_3rd_round / compute_opening_proofs use prove_keys[0].commit_key
        commits = {i: _Committer(prove_keys[i].commit_key, cp, st, sharded, group) for i in mine}
        commit0 = commits[first]
        # round-1 commitments from the witness values when the proving key holds the Lagrange-basis key; sharded: the
        # ranks partition the evaluation ranges exactly as they partition coefficient ranges on the monomial key
        lag_commits = {i: (_Committer(prove_keys[i].lagrange_key, cp, st, sharded, group)
                           if prove_keys[i].lagrange_key is not None else None) for i in mine}
        if place.dist:
            from . import dist as jdist

            engine = jdist.HipMsmEngine(L, ck, st)
        mont = lambda xs: H.fr_to_mont(cp, xs)  # noqa: E731
        gen = mont([cp.fr_generator])[0]
        group_gen = pow(cp.two_adic_root, 1 << (cp.two_adicity - log_n), r)
        group_gen_inv = pow(group_gen, -1, r)

        import time as _time

        t_last = [_time.perf_counter()]

        def tick(name):
            if MARKERS:
                torch.cuda.synchronize()
                with L.marker("prove step end: " + name):
                    pass
            if PROFILE is not None:
                torch.cuda.synchronize()
                now = _time.perf_counter()
                PROFILE[name] = round(PROFILE.get(name, 0.0) + (now - t_last[0]) * 1e3, 3)
                t_last[0] = now

        if MARKERS:
            torch.cuda.synchronize()
            with L.marker("prove step end: start"):
                pass

        def draw(count: int) -> list:
            return [H.fr_rand(cp, rng) for _ in range(count)]

        def apply_mask(poly: torch.Tensor, b: list):
            """mask_polynomial (prover.rs:457-466): + (b_0 + .. + b_h X^h)(X^n - 1); poly has room for n + h + 1"""
            k = 4 * len(b)
            bdev = _dev_fr(mont(b))
            # on the device (no read-back of the low coefficients: the host would wait for the iNTT and the device for the host)
            poly[:k] = lincomb([(1, poly[:k], len(b)), (r - 1, bdev, len(b))], len(b))
            poly[n4:n4 + k] = bdev

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

        def commit_sum(polys, lens) -> list:
            """commitments to the batch-wide polynomials (split quotient, opening polynomials).  One GPU: plain
            commits.  With placement every rank holds its own share of each polynomial: un-normalised partial sums,
            one all_gather, local fold (the polynomial is the sum of the shares and commit is linear)."""
            if not place.dist:
                return commit0(polys, lens)
            for k in lens:
                if k - 1 > len(ck) - 1:
                    raise PlonkError("poly degree %d is larger than allowed %d" % (k - 1, len(ck) - 1))
            parts = torch.cat([engine.partial(p[:4 * k]) for p, k in zip(polys, lens)])
            st.sync()
            allp = jdist._all_gather_bytes(parts, place.world, place.group).view(place.world, len(polys),
                                                                                 engine.partial_bytes)
            per = allp.transpose(0, 1).contiguous()
            out = []
            for j in range(len(polys)):
                raw = engine.fold(per[j].reshape(-1), place.world).cpu().numpy().tobytes()
                out.append(H.g1_from_record(cp, raw))
            return out

        # ---- Round 1 (prover.rs:77-91): wire polynomials, public-input polynomial; instances in order
        ln = n + 3
        l4 = ln * 4
        wire_vals_v, wire_polys_v, pub_poly_v, bufs = {}, {}, {}, {}
        for i in mine:
            circuit, pk = circuits[i], prove_keys[i]
            wm = witness_monts[i] if witness_monts is not None and witness_monts[i] is not None else mont(circuit.witness)
            wit = _dev_fr(wm).view(-1, 4)
            # witness[wire_variable(i, j)] for every (wire type, gate) (constraint_system.rs:1422-1444): one gather kernel;
            # an index beyond the witness table (a proving key of another circuit) is an error, not a silent row 0
            wire_vals = torch.empty(W, n4, dtype=torch.int64, device="cuda")
            L.witness_gather_device(wit.data_ptr(), wit.shape[0], pk.wire_variables32.data_ptr(), W * n,
                                    wire_vals.data_ptr(), check=True, stream=st)
            buf = torch.empty(W + 1, n4, dtype=torch.int64, device="cuda")
            buf[:W].copy_(wire_vals)
            buf[W].zero_()
            if pub_inputs[i]:
                buf[W, :4 * len(pub_inputs[i])] = _dev_fr(mont(pub_inputs[i]))   # io gates sit in rows 0 .. num_inputs-1
            # an empty public input is the zero polynomial: nothing to transform in row W
            L.ntt_device(c, buf.data_ptr(), log_n, W + (1 if pub_inputs[i] or not SKIP_ZERO_POLYS else 0), inverse=True,
                         stream=st)
            wire_vals_v[i] = wire_vals
            bufs[i] = buf
        # the transcript is initialised (snark.rs:231-237) while the device runs the witness gather and the iNTTs
        tr = StandardTranscript(cp)
        if extra_transcript_init_msg is not None:
            tr.append_message(b"extra info", extra_transcript_init_msg)
        for vk, pub_input in zip(vks, pub_inputs):
            tr.append_vk_and_pub_input(vk, pub_input)
        tick("r1_gather_intt_transcript_init")
        comms_mine = {}
        for i in range(K):
            # RNG draws in the reference's order on every rank: b_0, b_1 of wire 0, then wire 1, ...
            bs = [draw(2) for _ in range(W)]
            if not place.local(i):
                continue
            buf = bufs[i]
            wire_polys = torch.zeros(W, l4, dtype=torch.int64, device="cuda")
            wire_polys[:, :n4].copy_(buf[:W])
            # mask_polynomial for the W wire polynomials, on the device
            for j in range(W):
                apply_mask(wire_polys[j], bs[j])
            tick("r1_mask")
            if lag_commits[i] is not None:
                # the same commitments from the witness values: MSM over (w(1), w(omega), .., b_0, b_1) on the Lagrange key
                lw = torch.empty(W, n4 + 8, dtype=torch.int64, device="cuda")
                lw[:, :n4].copy_(wire_vals_v[i])
                lw[:, n4:] = wire_polys[:, n4:n4 + 8]
                comms_mine[i] = lag_commits[i]([lw[j] for j in range(W)], [n + 2] * W)
            else:
                comms_mine[i] = commits[i]([wire_polys[j] for j in range(W)], [n + 2] * W)
            tick("r1_commit")
            wire_polys_v[i] = wire_polys
            pub_poly_v[i] = buf[W]
        del bufs
        wires_poly_comms_vec = place.gather(comms_mine, K)
        for comms in wires_poly_comms_vec:
            tr.append_commitments(b"witness_poly_comms", comms)

        # ---- Round 1.5 (Plookup; snark.rs:254-284, prover.rs:99-124, constraint_system.rs:1495-1514, 1577-1624)
        tau = tr.get_challenge(b"tau") if any(lookup) else None
        h_polys_v, sorted_v, table_v, lkw_v, h_comms_mine = {}, {}, {}, {}, {}
        for i in range(K):
            if not lookup[i]:
                continue
            b1, b2 = draw(3), draw(3)
            if not place.local(i):
                continue
            pk = prove_keys[i]
            merged = torch.empty(2, n4, dtype=torch.int64, device="cuda")   # merged table, merged lookup witness
            L.plonk_lookup_merge_device(c, log_n, wire_vals_v[i].data_ptr(), pk.plookup_table_vals.data_ptr(),
                                        mont([tau])[0], merged[0].data_ptr(), merged[1].data_ptr(), stream=st)
            # only the first n - 1 rows look up; the sorted concatenation has 2n - 1 entries
            srt = _sorted_lookup_vector(merged[0].view(n, 4), merged[1].view(n, 4)[:n - 1], st).reshape(-1)
            h_polys = torch.zeros(2, l4, dtype=torch.int64, device="cuda")
            h_polys[0, :n4].copy_(srt[:n4])
            h_polys[1, :n4].copy_(srt[n4 - 4:])
            hb = torch.empty(2, n4, dtype=torch.int64, device="cuda")
            hb.copy_(h_polys[:, :n4])
            L.ntt_device(c, hb.data_ptr(), log_n, 2, inverse=True, stream=st)
            h_polys[:, :n4].copy_(hb)
            apply_mask(h_polys[0], b1)
            apply_mask(h_polys[1], b2)
            tick("r1p_sorted_intt_mask")
            if lag_commits[i] is not None:
                lh = torch.empty(2, n4 + 12, dtype=torch.int64, device="cuda")
                lh[0, :n4].copy_(srt[:n4])
                lh[1, :n4].copy_(srt[n4 - 4:])
                lh[:, n4:] = h_polys[:, n4:n4 + 12]
                h_comms_mine[i] = lag_commits[i]([lh[0], lh[1]], [n + 3, n + 3])
            else:
                h_comms_mine[i] = commits[i]([h_polys[0], h_polys[1]], [n + 3, n + 3])
            tick("r1p_commit")
            h_polys_v[i], sorted_v[i], table_v[i], lkw_v[i] = h_polys, srt, merged[0], merged[1]
        h_comms_vec = place.gather(h_comms_mine, K) if any(lookup) else None
        for i in range(K):
            if lookup[i]:
                tr.append_commitments(b"h_poly_comms", h_comms_vec[i])

        # ---- Round 2 (prover.rs:128-142; grand product constraint_system.rs:1394-1420)
        beta = tr.get_challenge(b"beta")
        gamma = tr.get_challenge(b"gamma")
        bg = mont([beta, gamma])
        perm_poly_v, perm_comms_mine = {}, {}
        for i in range(K):
            b = draw(3)
            if not place.local(i):
                continue
            pk = prove_keys[i]
            perm_poly = torch.zeros(l4, dtype=torch.int64, device="cuda")
            L.plonk_perm_product_device(c, log_n, W, wire_vals_v[i].data_ptr(), pk.id_perm_vals.data_ptr(),
                                        pk.sigma_vals.data_ptr(), bg[0], bg[1], perm_poly.data_ptr(), stream=st)
            L.ntt_device(c, perm_poly.data_ptr(), log_n, 1, inverse=True, stream=st)
            apply_mask(perm_poly, b)
            tick("r2_product_intt_mask")
            perm_comms_mine[i] = commits[i]([perm_poly], [n + 3])[0]
            tick("r2_commit")
            perm_poly_v[i] = perm_poly
        del wire_vals_v
        prod_perm_poly_comms_vec = place.gather(perm_comms_mine, K)
        tr.append_commitments(b"perm_poly_comms", prod_perm_poly_comms_vec)

        # ---- Round 2.5 (Plookup; snark.rs:298-318, prover.rs:147-178, constraint_system.rs:1516-1575)
        pl_poly_v, pl_comms_mine = {}, {}
        for i in range(K):
            if not lookup[i]:
                continue
            b = draw(3)
            if not place.local(i):
                continue
            pl_poly = torch.zeros(l4, dtype=torch.int64, device="cuda")
            L.plonk_lookup_product_device(c, log_n, lkw_v[i].data_ptr(), table_v[i].data_ptr(), sorted_v[i].data_ptr(),
                                          bg[0], bg[1], pl_poly.data_ptr(), stream=st)
            L.ntt_device(c, pl_poly.data_ptr(), log_n, 1, inverse=True, stream=st)
            apply_mask(pl_poly, b)
            tick("r2p_product_intt_mask")
            pl_comms_mine[i] = commits[i]([pl_poly], [n + 3])[0]
            tick("r2p_commit")
            pl_poly_v[i] = pl_poly
        del sorted_v, table_v, lkw_v
        pl_comms_vec = place.gather(pl_comms_mine, K) if any(lookup) else None
        for i in range(K):
            if lookup[i]:
                tr.append_commitments(b"plookup_poly_comms", [pl_comms_vec[i]])

        # ---- Round 3 (prover.rs:183-198, 491-661, 894-943)
        alpha = tr.get_challenge(b"alpha")
        alpha_3, alpha_7 = pow(alpha, 3, r), pow(alpha, 7, r)
        alpha_bases, ab = [], 1
        for i in range(K):
            alpha_bases.append(ab)
            ab = ab * (alpha_7 if lookup[i] else alpha_3) % r      # prover.rs:651-655
        # coset sharding: rank g of G evaluates the slice i = g (mod G) of the quotient coset, a coset of
        # size ml = m / G with offset h w_m^g; one all_gather of the slices' iNTTs rebuilds the quotient polynomial
        G, g = 1, 0
        if sharded:
            import torch.distributed as tdist

            G = tdist.get_world_size(group)
            g = tdist.get_rank(group)
            if not coset_sharding_applies(n, m, G):
                G, g = 1, 0
        def slice_quotient(g, G):
            """sum_i alpha_base_i t_i evaluated on the slice i = g (mod G) of the quotient coset (a coset of ml = m / G
            points with offset h w_m^g), transformed back: the coefficients of t(X) mod (X^ml - (h w_m^g)^ml)"""
            ml = m // G
            log_ml = ml.bit_length() - 1
            off_int = slice_offset(cp, m, g)
            off = mont([off_int])[0]
            lc4 = min(l4, ml * 4)
            out = torch.zeros(ml * 4, dtype=torch.int64, device="cuda")
            for i in mine:
                pk, vk, wire_polys, perm_poly = prove_keys[i], vks[i], wire_polys_v[i], perm_poly_v[i]
                NSEL = pk.selectors.shape[0]
                nlk = 7 if lookup[i] else 0
                fixed = None
                if pk.coset_evals_slices is not None and (g, G) in pk.coset_evals_slices:
                    fixed = pk.coset_evals_slices[(g, G)]
                elif pk.coset_evals is not None:
                    fixed = pk.coset_evals
                    if pk.coset_slice != (g, G):
                        raise PlonkError("proving key coset evaluations were kept for another slice of the quotient coset")
                if fixed is not None:
                    # selector / sigma (/ table) coset evaluations are resident with the proving key
                    if fixed.shape != (NSEL + W + (4 if lookup[i] else 0), ml * 4):
                        raise PlonkError("proving key coset evaluations were kept for another slice of the quotient coset")
                    npolys, o_w = W + 2 + nlk, 0
                    sel_ptr, sig_ptr = fixed[0].data_ptr(), fixed[NSEL].data_ptr()
                else:
                    npolys, o_w = NSEL + 2 * W + 2 + nlk, NSEL + W
                o_z, o_pi, o_lk = o_w + W, o_w + W + 1, o_w + W + 2
                cs = torch.empty(npolys, ml * 4, dtype=torch.int64, device="cuda")
                cs[:, n4:lc4].zero_()
                if fixed is None:
                    cs[:NSEL, :n4].copy_(pk.selectors)
                    cs[NSEL:o_w, :n4].copy_(pk.sigmas)
                    sel_ptr, sig_ptr = cs[0].data_ptr(), cs[NSEL].data_ptr()
                cs[o_w:o_z, :lc4].copy_(wire_polys[:, :lc4])
                cs[o_z, :lc4].copy_(perm_poly[:lc4])
                cs[o_pi, :n4].copy_(pub_poly_v[i])
                online = [wire_polys[j] for j in range(W)] + [perm_poly]      # polynomials of n + 2 / n + 3 coefficients
                online_rows = list(range(o_w, o_z + 1))
                if lookup[i]:
                    # the quotient kernel's Plookup rows: range_table, key_table, table_dom_sep, q_dom_sep, h_1, h_2, prod_lookup
                    if fixed is None:
                        cs[o_lk:o_lk + 4, :n4].copy_(pk.plookup_polys)
                    cs[o_lk + 4:o_lk + 6, :lc4].copy_(h_polys_v[i][:, :lc4])
                    cs[o_lk + 6, :lc4].copy_(pl_poly_v[i][:lc4])
                    online += [h_polys_v[i][0], h_polys_v[i][1], pl_poly_v[i]]
                    online_rows += [o_lk + 4, o_lk + 5, o_lk + 6]
                if lc4 < l4:
                    # ml = n < n + 3 coefficients: reduce mod X^ml - off^ml (the slice's vanishing polynomial)
                    fold = mont([1, pow(off_int, ml, r)])
                    k = (l4 - lc4) // 4
                    tmp = torch.empty(len(online), k * 4, dtype=torch.int64, device="cuda")
                    for j, poly in enumerate(online):
                        L.poly_lincomb_device(c, [poly.data_ptr(), poly.data_ptr() + lc4 * 8], [k, k], fold,
                                              tmp[j].data_ptr(), k, stream=st)
                    for j, row in enumerate(online_rows):
                        cs[row, :k * 4].copy_(tmp[j])
                # identically-zero polynomials: no transform, and the quotient kernel is told not to read their rows
                zmask = sum(1 << j for j, z in enumerate(pk.selector_is_zero) if z) if SKIP_ZERO_POLYS else 0
                pi_zero = SKIP_ZERO_POLYS and not pub_inputs[i]
                zero_rows = [o_pi] if pi_zero else []
                if fixed is None:
                    zero_rows += [j for j in range(NSEL) if (zmask >> j) & 1]
                elif lookup[i]:
                    zero_rows += list(range(o_lk, o_lk + 4))          # resident: copied below, not transformed
                _coset_ntt_rows(c, cs, zero_rows, log_ml, off, lc4 // 4, st)
                if fixed is not None and lookup[i]:
                    cs[o_lk:o_lk + 4].copy_(fixed[NSEL + W:])
                tick("r3_coset_ntts")
                chal = mont([alpha, beta, gamma] + ([tau] if lookup[i] else []) + [alpha_bases[i]])
                # G == 1: the slice is the whole quotient coset (offset Fr::GENERATOR, log_ml = log_m)
                L.plonk_quotient_masked_device(c, W, log_n, log_ml, off, sel_ptr, sig_ptr, cs[o_w].data_ptr(),
                                               cs[o_z].data_ptr(), 0 if pi_zero else cs[o_pi].data_ptr(),
                                               cs[o_lk].data_ptr() if lookup[i] else 0, mont(vk.k), chal,
                                               zmask, out.data_ptr(), stream=st)
                del cs                             # stream-ordered: the next allocation is used on the same stream
                tick("r3_quotient_intt")
            L.ntt_device(c, out.data_ptr(), log_ml, 1, inverse=True, coset_mont=off, stream=st)
            return out

        ml = m // G
        expected_degree = W * (n + 1) + 2
        # One GPU: the quotient has expected_degree + 1 <= W (n + 2) coefficients, fewer than the m = 8 n points the
        # reference evaluates (prover.rs:524-527).  Disjoint slices of the same coset -- 4 n + 2 n points for TurboPlonk,
        # 4 n + 2 n + n for UltraPlonk -- determine it as well (unslice_quotient): 6 / 8 (7 / 8) of the coset transforms
        # and of the quotient kernel's points, same polynomial, same proof bytes.
        slices = [(g, G)]
        if G == 1 and QUOTIENT_SLICES and all(prove_keys[i].coset_evals is None for i in mine):
            slices = quotient_slices(n, m, expected_degree + 1)
        if len(slices) > 1:
            quot = unslice_quotient(cp, c, [slice_quotient(gg, GG) for gg, GG in slices], slices, m, st)
        elif G == 1:
            quot = slice_quotient(0, 1)
        else:
            part = slice_quotient(g, G)
            quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
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
        # the split randomisers (prover.rs:926-940) are drawn while the device still works on the quotient: the next RNG
        # draws in the reference's order whether or not the degree check below passes
        randomizers = draw(W - 1)
        owns_first = place.local(0)           # the blinding of the split goes on the share that holds instance 0
        split = torch.zeros(W, l4, dtype=torch.int64, device="cuda")
        step = (n + 2) * 4
        for i in range(W - 1):
            split[i, :step].copy_(quot[i * step:(i + 1) * step])
        last_len = expected_degree + 1 - (W - 1) * (n + 2)
        split[W - 1, :last_len * 4].copy_(quot[(W - 1) * step:(expected_degree + 1) * 4])
        if owns_first:
            split[:W - 1, step:step + 4] = _dev_fr(mont(randomizers)).view(W - 1, 4)
        # ONE read-back: [tail non-zero flag, top coefficient, first coefficient of every piece]
        flag = torch.any(quot[(expected_degree + 1) * 4:] != 0).to(torch.int64).view(1)
        back = torch.cat([flag, quot[expected_degree * 4:(expected_degree + 1) * 4],
                          split[:, :4].contiguous().view(-1)]).cpu().numpy()
        del quot
        # degree check on the whole quotient = the sum of the ranks' shares (prover.rs:900-903)
        if place.dist:
            per_rank = [None] * place.world
            place.tdist.all_gather_object(per_rank, (int(back[0]), H.fr_from_mont(cp, back[1:5].view(np.uint64))[0]),
                                          group=place.group)
            tail_bad, top = any(t for t, _ in per_rank), sum(x for _, x in per_rank) % r
        else:
            tail_bad, top = bool(back[0]), int(back[1:5].any())
        if tail_bad or not top:
            raise PlonkError("WrongQuotientPolyDegree(expected %d)" % expected_degree)
        if owns_first:
            firsts = H.fr_from_mont(cp, back[5:].view(np.uint64))
            last_randomizer = 0
            for i in range(W - 1):
                firsts[i] = (firsts[i] - last_randomizer) % r
                last_randomizer = randomizers[i]
            firsts[W - 1] = (firsts[W - 1] - last_randomizer) % r
            split[:, :4] = _dev_fr(mont(firsts)).view(W, 4)
        split_lens = [n + 3] * (W - 1) + [last_len]
        tick("r3_split_mask")
        split_quot_poly_comms = commit_sum([split[j] for j in range(W)], split_lens)
        tr.append_commitments(b"quot_poly_comms", split_quot_poly_comms)
        tick("r3_commit")

        # ---- Round 4 / 4.5 (prover.rs:205-296): evaluations at zeta and zeta * g, instances in order
        zeta = tr.get_challenge(b"zeta")
        zeta_omega = zeta * group_gen % r
        zm, zom = mont([zeta])[0], mont([zeta_omega])[0]
        at_zeta_v, at_next_v, evals_mine = {}, {}, {}
        pending = []
        for i in mine:
            pk = prove_keys[i]
            at_zeta = [(wire_polys_v[i][j], n + 2) for j in range(W)] + [(pk.sigmas[j], n) for j in range(W - 1)]
            at_next = [(perm_poly_v[i], n + 3)]
            if lookup[i]:
                rt, kt, tds, qds = (pk.plookup_polys[j] for j in range(4))
                ql = pk.selectors[pk.selectors.shape[0] - 1]
                h1, h2 = h_polys_v[i][0], h_polys_v[i][1]
                # plookup_open_polys_ref / plookup_shifted_open_polys_ref (prover.rs:421-453)
                at_zeta += [(rt, n), (kt, n), (h1, n + 3), (ql, n), (tds, n), (qds, n)]
                at_next += [(pl_poly_v[i], n + 3), (rt, n), (kt, n), (h1, n + 3), (h2, n + 3), (ql, n),
                            (wire_polys_v[i][3], n + 2), (wire_polys_v[i][4], n + 2), (tds, n)]
            ev = torch.zeros((len(at_zeta) + len(at_next)) * 4, dtype=torch.int64, device="cuda")
            L.poly_eval_batch_device(c, [p.data_ptr() for p, _ in at_zeta], [k for _, k in at_zeta], zm, ev.data_ptr(),
                                     stream=st)
            L.poly_eval_batch_device(c, [p.data_ptr() for p, _ in at_next], [k for _, k in at_next], zom,
                                     ev.data_ptr() + len(at_zeta) * 32, stream=st)
            at_zeta_v[i], at_next_v[i] = at_zeta, at_next
            pending.append((i, ev))
        if pending:
            flat = _read_fr(cp, torch.cat([ev for _, ev in pending]))
            at = 0
            for i, ev in pending:
                cnt = ev.numel() // 4
                evals_mine[i] = flat[at:at + cnt]
                at += cnt
        evals_all = place.gather(evals_mine, K)
        poly_evals_vec, plookup_evals_vec = [], []
        for i in range(K):
            ev = evals_all[i]
            nz = 2 * W - 1 + (6 if lookup[i] else 0)
            pe = ProofEvaluations(ev[:W], ev[W:2 * W - 1], ev[nz])
            poly_evals_vec.append(pe)
            plookup_evals_vec.append(PlookupEvaluations.from_vecs(ev[2 * W - 1:nz], ev[nz + 1:nz + 10]) if lookup[i]
                                     else None)
        for pe in poly_evals_vec:
            tr.append_field_elems(b"wire_evals", pe.wires_evals)
            tr.append_field_elems(b"wire_sigma_evals", pe.wire_sigma_evals)
            tr.append_field_elems(b"perm_next_eval", [pe.perm_next_eval])
        for ple in plookup_evals_vec:
            if ple is not None:      # transcript/mod.rs:155-162
                tr.append_field_elems(b"plookup_evals", ple.evals_vec())
                tr.append_field_elems(b"plookup_next_evals", ple.next_evals_vec())
        tick("r4_evals")

        # linearisation polynomial (snark.rs:364-389; prover.rs:299-352, 946-1087) as device linear combinations: the
        # quotient part over this rank's split share + the part of each of its instances
        vanish_eval = (pow(zeta, n, r) - 1) % r
        zeta_to_n_plus_2 = (vanish_eval + 1) * zeta % r * zeta % r
        lagrange_1_eval = vanish_eval * pow(n * (zeta - 1) % r, -1, r) % r
        lagrange_n_eval = vanish_eval * group_gen_inv % r * pow(n * (zeta - group_gen_inv) % r, -1, r) % r
        terms = []
        coeff = 1
        for i in range(W):
            if i:
                coeff = coeff * zeta_to_n_plus_2 % r
            terms.append((-vanish_eval * coeff % r, split[i], split_lens[i]))
        for i in mine:
            pk, vk, pe, ab = prove_keys[i], vks[i], poly_evals_vec[i], alpha_bases[i]
            w = pe.wires_evals
            q_scalars = [w[0], w[1], w[2], w[3], w[0] * w[1], w[2] * w[3], pow(w[0], 5, r), pow(w[1], 5, r),
                         pow(w[2], 5, r), pow(w[3], 5, r), -w[4], 1, w[0] * w[1] * w[2] * w[3] * w[4]]
            terms += [(s % r * ab % r, pk.selectors[j], n) for j, s in enumerate(q_scalars)]   # q_lookup is not in r_circ
            cz = alpha
            for j in range(W):
                cz = cz * (w[j] + beta * vk.k[j] % r * zeta + gamma) % r
            cz = (cz + alpha * alpha % r * lagrange_1_eval) % r
            terms.append((cz * ab % r, perm_poly_v[i], n + 3))
            csig = alpha * beta % r * pe.perm_next_eval % r
            for j in range(W - 1):
                csig = csig * (w[j] + beta * pe.wire_sigma_evals[j] + gamma) % r
            terms.append((-csig * ab % r, pk.sigmas[W - 1], n))
            if lookup[i]:
                # compute_lin_poly_plookup_contribution (prover.rs:1017-1087)
                le = plookup_evals_vec[i]
                a4, a5, a6 = pow(alpha, 4, r), pow(alpha, 5, r), pow(alpha, 6, r)

                def merged_table(rg, ky, ql, w3, w4, td):          # eval_merged_table, structs.rs:899-914
                    return (rg + ql * tau % r * (td + tau * (ky + tau * (w3 + tau * w4)))) % r

                mt = merged_table(le.range_table_eval, le.key_table_eval, le.q_lookup_eval, w[3], w[4],
                                  le.table_dom_sep_eval)
                mt_next = merged_table(le.range_table_next_eval, le.key_table_next_eval, le.q_lookup_next_eval,
                                       le.w_3_next_eval, le.w_4_next_eval, le.table_dom_sep_next_eval)
                # eval_merged_lookup_witness, structs.rs:916-931
                ml_ev = (w[5] + le.q_lookup_eval * tau % r * (le.q_dom_sep_eval + tau * (w[0] + tau * (w[1] + tau * w[2])))) % r
                b1 = (1 + beta) % r
                zmg = (zeta - group_gen_inv) % r
                c_p = (a4 * lagrange_1_eval + a5 * lagrange_n_eval
                       + a6 * zmg % r * b1 % r * (gamma + ml_ev) % r * (gamma * b1 + mt + beta * mt_next)) % r
                terms.append((c_p * ab % r, pl_poly_v[i], n + 3))
                c_h2 = -a6 * zmg % r * le.prod_next_eval % r * (gamma * b1 + le.h_1_eval + beta * le.h_1_next_eval) % r
                terms.append((c_h2 * ab % r, h_polys_v[i][1], n + 3))
        lin_poly = lincomb(terms, n + 3)
        tick("r4_lin_poly")

        # ---- Round 5 (prover.rs:356-413, 470-488): batched opening quotients over every instance; division by
        # (X - z) is linear, so with placement each rank divides and commits its own share
        v = tr.get_challenge(b"v")
        open_terms = [(1, lin_poly, n + 3)]
        sh_terms = []
        pos_z, pos_n = 1, 0                     # exponent of v of the next polynomial at zeta / at zeta * g
        for i in range(K):
            cnt_z = 2 * W - 1 + (6 if lookup[i] else 0)
            cnt_n = 1 + (9 if lookup[i] else 0)
            if place.local(i):
                cur = pow(v, pos_z, r)
                for p, k in at_zeta_v[i]:
                    open_terms.append((cur, p, k))
                    cur = cur * v % r
                cur = pow(v, pos_n, r)
                for p, k in at_next_v[i]:
                    sh_terms.append((cur, p, k))
                    cur = cur * v % r
            pos_z += cnt_z
            pos_n += cnt_n
        open_poly = lincomb(open_terms, n + 3)
        if not place.dist and K == 1 and not lookup[0]:
            shifted_poly = perm_poly_v[0]
        else:
            shifted_poly = lincomb(sh_terms, n + 3)
        wpoly = torch.zeros(2, l4, dtype=torch.int64, device="cuda")
        L.poly_divide_linear_device(c, open_poly.data_ptr(), n + 3, zm, wpoly[0].data_ptr(), stream=st)
        L.poly_divide_linear_device(c, shifted_poly.data_ptr(), n + 3, zom, wpoly[1].data_ptr(), stream=st)
        tick("r5_open_polys")
        opening_proof, shifted_opening_proof = commit_sum([wpoly[0][4:], wpoly[1][4:]], [n + 2, n + 2])
        tick("r5_commit")

        plookup_proofs_vec = None
        if any(lookup):
            plookup_proofs_vec = [PlookupProof(h_comms_vec[i], pl_comms_vec[i], plookup_evals_vec[i]) if lookup[i] else None
                                  for i in range(K)]
        return BatchProof(c, wires_poly_comms_vec, prod_perm_poly_comms_vec, poly_evals_vec, split_quot_poly_comms,
                          opening_proof, shifted_opening_proof, plookup_proofs_vec)
