"""Host-side mirror of the reference's plonky_block_proof_gen API over the C ABI (include/bpg.h).

Same names, argument meaning and error behaviour as the Rust crate:
  ProverStateBuilder / ProverState      plonky_block_proof_gen/src/prover_state.rs:17-100
  generate_txn_proof / agg / block      plonky_block_proof_gen/src/proof_gen.rs:39-110
  GeneratedTxnProof / AggProof / BlockProof / AggregatableProof
                                        plonky_block_proof_gen/src/proof_types.rs:12-87
  VerifierState                         plonky_block_proof_gen/src/verifier_state.rs:19-71
  ProofGenError                         plonky_block_proof_gen/src/proof_gen.rs:16-36
There is no CPU fallback: everything below calls libbpg.so.
"""
import ctypes as C
import struct
from dataclasses import dataclass

from ._lib import BpgError, check, lib, take_buffer

NUM_TABLES = 7
TABLES = ("arithmetic", "byte_packing", "cpu", "keccak", "keccak_sponge", "logic", "memory")
IR_WORDS, PV_WORDS = 25, 13

ProofGenError = BpgError  # Result<T, ProofGenError(String)> -> exception carrying the message


class BpConfig(C.Structure):
    """bp_config (include/bpg.h)."""
    _fields_ = ([("table_log_lo", C.c_uint32 * NUM_TABLES), ("table_log_hi", C.c_uint32 * NUM_TABLES)]
                + [(n, C.c_uint32) for n in ("stark_rate_bits", "stark_cap_height", "stark_num_queries",
                                             "stark_pow_bits", "arity_bits", "final_poly_bits", "rec_log_n",
                                             "rec_n_cols", "rec_n_const", "rec_rate_bits", "rec_num_queries",
                                             "rec_pow_bits", "shrink_depth", "rec_air_id")]
                + [("device", C.c_int32), ("n_workers", C.c_uint32), ("arena_bytes", C.c_uint64)])


def _bind():
    L = lib()
    if getattr(L, "_pg_bound", False):
        return L
    vp, u8p, szp = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
    L.bp_config_default.argtypes = [C.POINTER(BpConfig)]
    L.bp_config_default.restype = None
    L.bp_state_build.argtypes = [C.POINTER(BpConfig), C.POINTER(vp)]
    L.bp_state_free.argtypes = [vp]
    L.bp_state_free.restype = None
    L.bp_state_device_bytes.argtypes = [vp]
    L.bp_state_device_bytes.restype = C.c_uint64
    L.bp_state_warnings.argtypes = [vp]
    L.bp_state_warnings.restype = C.c_char_p
    L.bp_generate_txn_proof.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(u8p), szp]
    L.bp_generate_agg_proof.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_int,
                                        C.POINTER(u8p), szp]
    L.bp_generate_block_proof.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(u8p), szp,
                                          C.POINTER(C.c_uint64)]
    L.bp_verifier_state_from_prover.argtypes = [vp, C.POINTER(vp)]
    L.bp_verifier_state_build.argtypes = [C.POINTER(BpConfig), C.POINTER(vp)]
    L.bp_verifier_state_from_caps.argtypes = [C.POINTER(BpConfig), C.POINTER(C.c_uint64), C.POINTER(vp)]
    L.bp_verifier_state_free.argtypes = [vp]
    L.bp_verifier_state_free.restype = None
    L.bp_verify_block_proof.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.bp_verify_proof.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.bp_ir_encode.argtypes = [C.c_uint64] * 4 + [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint32),
                                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.bp_ir_encode_dummy.argtypes = [C.c_uint64] * 3 + [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint32),
                                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.bp_ir_set_keccak_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_ir_set_logic_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_ir_set_memory_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_ir_set_arithmetic_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_ir_set_arithmetic_mul_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_ir_set_byte_packing_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_ir_set_keccak_sponge_air.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bp_keccak256_sponge_rows.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t,
                                           C.POINTER(C.c_size_t)]
    L.bp_keccak256_permutation_inputs.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
    L.bp_generate_txn_proof_witness.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.bp_generate_txn_proof_keccak.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t,
                                               C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.bp_proof_public_values.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L._pg_bound = True
    return L


@dataclass(frozen=True)
class PublicValues:
    """Subset of plonky2_evm PublicValues carried by the synthetic workload."""
    txn_number_before: int
    txn_number_after: int
    gas_used_before: int
    gas_used_after: int
    state_root_before: tuple
    state_root_after: tuple
    block_number: int

    @classmethod
    def from_words(cls, w):
        return cls(w[0], w[1], w[2], w[3], tuple(w[4:8]), tuple(w[8:12]), w[12])


def state_root_after(root_before, seed, txn_number):
    """bp_state_root_after: the synthetic state transition of one (non-dummy) transaction."""
    L = _bind()
    L.bp_state_root_after.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 4)()
    check(L.bp_state_root_after((C.c_uint64 * 4)(*root_before), seed, txn_number, out))
    return tuple(out)


def public_values_of(proof_bytes):
    L = _bind()
    pv = (C.c_uint64 * PV_WORDS)()
    kind = C.c_int()
    check(L.bp_proof_public_values(proof_bytes, len(proof_bytes), pv, C.byref(kind)))
    return PublicValues.from_words(list(pv)), kind.value


@dataclass(frozen=True)
class TxnProofGenIR:
    """Synthetic stand-in for protocol_decoder::types::TxnProofGenIR (= GenerationInputs,
    protocol_decoder/src/types.rs:48): what one txn proof is generated from."""
    block_number: int
    txn_number_before: int
    gas_used_before: int
    gas_used_after: int
    state_root_before: tuple
    seed: int
    table_log_n: tuple
    table_width: tuple
    dummy: bool = False   # a padding entry (decoding.rs:484-520): proven, but txn number / gas / state root stay
    keccak_air: bool = False   # the Keccak table (index 3) is a real Keccak-f[1600] trace (AIR 1, 2431 columns)
    keccak_inputs: tuple = None   # ... attesting THESE permutations ([n][25] lanes) instead of seeded ones; not part of
                                  # the 25-word IR: handed to bp_generate_txn_proof_keccak beside it
    logic_air: bool = False    # the logic table (index 5) is proven with the logic AIR (AIR 2, 524 columns)
    memory_air: bool = False   # the memory table (index 6) with the memory AIR (AIR 3, 45 columns)
    arithmetic_air: bool = False   # the arithmetic table (index 0) with the arithmetic AIR (AIR 4, 309 columns)
    byte_packing_air: bool = False   # the byte-packing table (index 1) with the byte-packing AIR (AIR 5, 299 columns)
    keccak_sponge_air: bool = False  # the Keccak sponge table (index 4) with the Keccak sponge AIR (AIR 6, 2414 columns)
    arithmetic_mul_air: bool = False   # the arithmetic table (index 0) with the multiplication AIR instead (AIR 7, 1217 columns)
    witness: tuple = None   # ((table index, ((words of an item), ...)), ...): data for tables with an AIR instead of a
                            # seeded witness (bp_generate_txn_proof_witness); like keccak_inputs not part of the 25-word IR

    def to_bytes(self):
        L = _bind()
        out = (C.c_uint64 * IR_WORDS)()
        root = (C.c_uint64 * 4)(*self.state_root_before)
        logs, widths = (C.c_uint32 * NUM_TABLES)(*self.table_log_n), (C.c_uint32 * NUM_TABLES)(*self.table_width)
        if self.dummy:
            if self.gas_used_after != self.gas_used_before:
                raise ValueError("a dummy entry uses no gas (decoding.rs:503-506)")
            check(L.bp_ir_encode_dummy(self.block_number, self.txn_number_before, self.gas_used_before, root, self.seed,
                                       logs, widths, out))
        else:
            check(L.bp_ir_encode(self.block_number, self.txn_number_before, self.gas_used_before, self.gas_used_after,
                                 root, self.seed, logs, widths, out))
        if self.keccak_air:
            check(L.bp_ir_set_keccak_air(out, 1))
        if self.logic_air:
            check(L.bp_ir_set_logic_air(out, 1))
        if self.memory_air:
            check(L.bp_ir_set_memory_air(out, 1))
        if self.arithmetic_air:
            check(L.bp_ir_set_arithmetic_air(out, 1))
        if self.byte_packing_air:
            check(L.bp_ir_set_byte_packing_air(out, 1))
        if self.keccak_sponge_air:
            check(L.bp_ir_set_keccak_sponge_air(out, 1))
        if self.arithmetic_mul_air:
            check(L.bp_ir_set_arithmetic_mul_air(out, 1))
        return struct.pack("<%dQ" % IR_WORDS, *out)


@dataclass(frozen=True)
class GeneratedTxnProof:
    p_vals: PublicValues
    intern: bytes


@dataclass(frozen=True)
class GeneratedAggProof:
    p_vals: PublicValues
    intern: bytes


@dataclass(frozen=True)
class GeneratedBlockProof:
    b_height: int
    intern: bytes


class AggregatableProof:
    """Sum type Txn | Agg (proof_types.rs:46-87)."""

    def __init__(self, proof):
        if not isinstance(proof, (GeneratedTxnProof, GeneratedAggProof)):
            raise TypeError("AggregatableProof wraps a txn or an agg proof")
        self._p = proof

    def public_values(self):
        return self._p.p_vals

    def is_agg(self):
        return isinstance(self._p, GeneratedAggProof)

    def intern(self):
        return self._p.intern


def _as_aggregatable(p):
    return p if isinstance(p, AggregatableProof) else AggregatableProof(p)


class ProverState:
    """Pre-processed circuits resident in HBM (prover_state.rs:17-20).  Immutable; share freely
    between threads."""

    def __init__(self, handle, cfg):
        self._h, self.cfg = handle, cfg

    @property
    def device_bytes(self):
        return _bind().bp_state_device_bytes(self._h)

    @property
    def warnings(self):
        """What bp_state_build found about its environment (GPU_MAX_HW_QUEUES below n_workers, the host-wait mode
        the device was left in); "" when there is nothing to say."""
        return _bind().bp_state_warnings(self._h).decode("utf-8", "replace") if self._h else ""

    def close(self):
        if self._h:
            _bind().bp_state_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass


class ProverStateBuilder:
    """Builder with the reference's default ranges (constants.rs:6-18) and
    set_<table>_circuit_size(range) setters (prover_state.rs:55-75)."""

    def __init__(self):
        self.cfg = BpConfig()
        _bind().bp_config_default(C.byref(self.cfg))

    def _set(self, idx, size):
        self.cfg.table_log_lo[idx], self.cfg.table_log_hi[idx] = size.start, size.stop
        return self

    def set_arithmetic_circuit_size(self, size): return self._set(0, size)
    def set_byte_packing_circuit_size(self, size): return self._set(1, size)
    def set_cpu_circuit_size(self, size): return self._set(2, size)
    def set_keccak_circuit_size(self, size): return self._set(3, size)
    def set_keccak_sponge_circuit_size(self, size): return self._set(4, size)
    def set_logic_circuit_size(self, size): return self._set(5, size)
    def set_memory_circuit_size(self, size): return self._set(6, size)

    def set(self, **kw):
        """Non-reference knobs: device, n_workers, arena_bytes, recursion/STARK shape parameters."""
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            setattr(self.cfg, k, v)
        return self

    def build(self):
        h = C.c_void_p()
        check(_bind().bp_state_build(C.byref(self.cfg), C.byref(h)))
        return ProverState(h, self.cfg)

    def build_verifier(self):
        h = C.c_void_p()
        check(_bind().bp_verifier_state_build(C.byref(self.cfg), C.byref(h)))
        return VerifierState(h)


def _out():
    return C.POINTER(C.c_uint8)(), C.c_size_t()


def keccak256_permutation_inputs(data: bytes):
    """bp_keccak256_permutation_inputs: -> (digest, [n_perms][25] lanes): the states that go into the Keccak-f
    permutations of Keccak-256(data) -- what a Keccak table attesting this hash contains."""
    L = _bind()
    n = C.c_size_t()
    check(L.bp_keccak256_permutation_inputs(data, len(data), None, None, 0, C.byref(n)))
    states, digest = (C.c_uint64 * (25 * n.value))(), C.create_string_buffer(32)
    check(L.bp_keccak256_permutation_inputs(data, len(data), digest, states, n.value, C.byref(n)))
    return digest.raw, [list(states[25 * i:25 * i + 25]) for i in range(n.value)]


class TxnWitness(C.Structure):
    """bp_txn_witness (include/bpg.h)"""
    _fields_ = [(n, t) for name in ("keccak_inputs:n_perms:has_keccak", "logic_ops:n_logic_ops:has_logic",
                                    "memory_log:n_memory_ops:has_memory", "arithmetic_ops:n_arithmetic_ops:has_arithmetic",
                                    "byte_sequences:n_byte_sequences:has_byte_packing",
                                    "sponge_rows:n_sponge_rows:has_keccak_sponge")
                for n, t in zip(name.split(":"), (C.c_void_p, C.c_size_t, C.c_int))]


WITNESS_FIELDS = {3: ("keccak_inputs", "n_perms", "has_keccak", 25), 5: ("logic_ops", "n_logic_ops", "has_logic", 9),
                  6: ("memory_log", "n_memory_ops", "has_memory", 11), 0: ("arithmetic_ops", "n_arithmetic_ops", "has_arithmetic", 9),
                  1: ("byte_sequences", "n_byte_sequences", "has_byte_packing", 6),
                  4: ("sponge_rows", "n_sponge_rows", "has_keccak_sponge", 44)}


def keccak256_sponge_rows(data: bytes):
    """bp_keccak256_sponge_rows: -> (digest, [n_blocks][44] words): the rows a Keccak sponge table (AIR 6) absorbing
    `data` contains -- flags, message bytes in the block, the block as absorbed, the state before it."""
    L = _bind()
    n = C.c_size_t()
    check(L.bp_keccak256_sponge_rows(data, len(data), None, None, 0, C.byref(n)))
    rows, digest = (C.c_uint64 * (44 * n.value))(), C.create_string_buffer(32)
    check(L.bp_keccak256_sponge_rows(data, len(data), digest, rows, n.value, C.byref(n)))
    return digest.raw, [list(rows[44 * i:44 * i + 44]) for i in range(n.value)]


def generate_txn_proof(p_state, gen_inputs, abort_signal=None, keccak_inputs=None, witness=None):
    """proof_gen.rs:39-56.  abort_signal: optional shared flag (the reference's Option<Arc<AtomicBool>>): a
    ctypes.c_uint8 / c_bool (one byte, what AtomicBool is: bp_generate_txn_proof_u8) or a ctypes.c_int32.
    keccak_inputs: the permutation inputs ([n][25] lanes) of the transaction's Keccak table, for an IR with
    keccak_air=True (bp_generate_txn_proof_keccak); default: gen_inputs.keccak_inputs if it has any.
    witness: {table index: [[words of an item], ...]} for the tables proven with an AIR (0 arithmetic [9], 1 byte packing
    [6], 3 Keccak [25], 4 Keccak sponge [44], 5 logic [9], 6 memory [11]; bp_generate_txn_proof_witness); default:
    gen_inputs.witness."""
    L = _bind()
    ir = gen_inputs.to_bytes() if isinstance(gen_inputs, TxnProofGenIR) else bytes(gen_inputs)
    out, n = _out()
    flag = C.byref(abort_signal) if abort_signal is not None else None
    if keccak_inputs is None:
        keccak_inputs = getattr(gen_inputs, "keccak_inputs", None)
    if witness is None and getattr(gen_inputs, "witness", None) is not None:
        witness = dict(gen_inputs.witness)
    if witness is not None:
        if keccak_inputs is not None and 3 not in witness:
            witness = {**witness, 3: keccak_inputs}
        if abort_signal is not None and C.sizeof(abort_signal) != 1:
            raise ValueError("bp_generate_txn_proof_witness takes the one-byte abort flag")
        w, keep = TxnWitness(), []
        for t, items in witness.items():
            ptr_f, n_f, has_f, words = WITNESS_FIELDS[t]
            flat = [int(x) for it in items for x in it]
            if len(flat) % words:
                raise ValueError("witness items of table %d have %d words each" % (t, words))
            a = (C.c_uint64 * max(len(flat), 1))(*flat)
            keep.append(a)
            setattr(w, ptr_f, C.cast(a, C.c_void_p))
            setattr(w, n_f, len(flat) // words)
            setattr(w, has_f, 1)
        check(L.bp_generate_txn_proof_witness(p_state._h, ir, len(ir), C.byref(w), flag, C.byref(out), C.byref(n)))
    elif keccak_inputs is not None:
        flat = [int(x) for st in keccak_inputs for x in st]
        arr = (C.c_uint64 * max(len(flat), 1))(*flat)
        if abort_signal is not None and C.sizeof(abort_signal) != 1:
            raise ValueError("bp_generate_txn_proof_keccak takes the one-byte abort flag")
        check(L.bp_generate_txn_proof_keccak(p_state._h, ir, len(ir), arr, len(flat) // 25, flag, C.byref(out), C.byref(n)))
    elif abort_signal is not None and C.sizeof(abort_signal) == 1:
        L.bp_generate_txn_proof_u8.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p,
                                               C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        check(L.bp_generate_txn_proof_u8(p_state._h, ir, len(ir), flag, C.byref(out), C.byref(n)))
    else:
        check(L.bp_generate_txn_proof(p_state._h, ir, len(ir), flag, C.byref(out), C.byref(n)))
    intern = take_buffer(out, n)
    return GeneratedTxnProof(public_values_of(intern)[0], intern)


def _witness_struct(gen_inputs, keccak_inputs, witness):
    """(TxnWitness or None, the ctypes arrays it points into) from the forms generate_txn_proof accepts."""
    if keccak_inputs is None:
        keccak_inputs = getattr(gen_inputs, "keccak_inputs", None)
    if witness is None and getattr(gen_inputs, "witness", None) is not None:
        witness = dict(gen_inputs.witness)
    if witness is None and keccak_inputs is None:
        return None, []
    witness = dict(witness or {})
    if keccak_inputs is not None and 3 not in witness:
        witness[3] = keccak_inputs
    w, keep = TxnWitness(), []
    for t, items in witness.items():
        ptr_f, n_f, has_f, words = WITNESS_FIELDS[t]
        flat = [int(x) for it in items for x in it]
        if len(flat) % words:
            raise ValueError("witness items of table %d have %d words each" % (t, words))
        a = (C.c_uint64 * max(len(flat), 1))(*flat)
        keep.append(a)
        setattr(w, ptr_f, C.cast(a, C.c_void_p))
        setattr(w, n_f, len(flat) // words)
        setattr(w, has_f, 1)
    return w, keep


def generate_txn_table_proofs(p_state, gen_inputs, keccak_inputs=None, witness=None):
    """What upstream's `prove` yields before the recursion (its AllProof; reached from proof_gen.rs:44-52): the seven
    table proofs of the transaction on their one transcript, with the public values and the lookup challenges, as
    bytes (bp_generate_txn_table_proofs).  Arguments as for generate_txn_proof."""
    L = _bind()
    ir = gen_inputs.to_bytes() if isinstance(gen_inputs, TxnProofGenIR) else bytes(gen_inputs)
    out, n = _out()
    w, keep = _witness_struct(gen_inputs, keccak_inputs, witness)
    L.bp_generate_txn_table_proofs.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    check(L.bp_generate_txn_table_proofs(p_state._h, ir, len(ir), C.byref(w) if w is not None else None, None,
                                         C.byref(out), C.byref(n)))
    del keep
    return take_buffer(out, n)


def verify_txn_table_proofs(cfg, table_proofs, gen_inputs=None):
    """upstream's verify_proof(all_stark, all_proof, config) on the CPU: every table proof against the shared transcript
    and the cross-table lookups between the tables proven with their AIRs.  cfg: a BpConfig (ProverState.cfg).
    gen_inputs (a TxnProofGenIR or its bytes): the statement -- which AIR proves each table, the shapes, the public
    values -- is then the verifier's (bp_verify_txn_table_proofs_for); without it the blob's header is taken at its
    word and the caller must check it.  Raises ProofGenError when rejected."""
    L = _bind()
    b = bytes(table_proofs)
    if gen_inputs is None:
        L.bp_verify_txn_table_proofs.argtypes = [C.POINTER(BpConfig), C.c_char_p, C.c_size_t]
        check(L.bp_verify_txn_table_proofs(C.byref(cfg), b, len(b)))
        return
    ir = gen_inputs.to_bytes() if hasattr(gen_inputs, "to_bytes") else bytes(gen_inputs)
    L.bp_verify_txn_table_proofs_for.argtypes = [C.POINTER(BpConfig), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    check(L.bp_verify_txn_table_proofs_for(C.byref(cfg), ir, len(ir), b, len(b)))


def generate_agg_proof(p_state, lhs_child, rhs_child):
    """proof_gen.rs:61-79."""
    L = _bind()
    lhs, rhs = _as_aggregatable(lhs_child), _as_aggregatable(rhs_child)
    out, n = _out()
    check(L.bp_generate_agg_proof(p_state._h, lhs.intern(), len(lhs.intern()), int(lhs.is_agg()), rhs.intern(),
                                  len(rhs.intern()), int(rhs.is_agg()), C.byref(out), C.byref(n)))
    intern = take_buffer(out, n)
    return GeneratedAggProof(public_values_of(intern)[0], intern)


def generate_block_proof(p_state, prev_opt_parent_b_proof, curr_block_agg_proof):
    """proof_gen.rs:85-110."""
    L = _bind()
    parent = prev_opt_parent_b_proof.intern if prev_opt_parent_b_proof is not None else None
    out, n = _out()
    h = C.c_uint64()
    check(L.bp_generate_block_proof(p_state._h, parent, len(parent) if parent else 0, curr_block_agg_proof.intern,
                                    len(curr_block_agg_proof.intern), C.byref(out), C.byref(n), C.byref(h)))
    return GeneratedBlockProof(h.value, take_buffer(out, n))


class VerifierState:
    """verifier_state.rs:19-23; built from a ProverState (:46-52) or a builder (:34-42)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_prover_state(cls, p_state):
        h = C.c_void_p()
        check(_bind().bp_verifier_state_from_prover(p_state._h, C.byref(h)))
        return cls(h)

    @classmethod
    def from_caps(cls, cfg, caps_words):
        """Verifier-only deployment (no GPU): cfg is a BpConfig, caps_words the 3 circuit caps."""
        h = C.c_void_p()
        arr = (C.c_uint64 * len(caps_words))(*[int(x) for x in caps_words])
        check(_bind().bp_verifier_state_from_caps(C.byref(cfg), arr, C.byref(h)))
        return cls(h)

    def verify(self, block_proof):
        """verifier_state.rs:56-71.  Raises ProofGenError when the proof is rejected."""
        b = block_proof.intern if isinstance(block_proof, GeneratedBlockProof) else bytes(block_proof)
        check(_bind().bp_verify_block_proof(self._h, b, len(b)))

    def verify_any(self, proof):
        b = proof.intern if hasattr(proof, "intern") and not callable(proof.intern) else bytes(proof)
        check(_bind().bp_verify_proof(self._h, b, len(b)))

    def close(self):
        if self._h:
            _bind().bp_verifier_state_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass
