"""Host-side mirror of `BlockTrace::into_txn_proof_gen_ir`:
protocol_decoder/src/processed_block_trace.rs:38-50 and decoding.rs:81-177.  The work is done by the native
library (csrc/decoding.cpp, csrc/mpt.cpp) behind `bp_decode_block_trace`; this module serialises the payload into
the ABI's byte form and parses the resulting `GenerationInputs` back into objects named like the reference's.

    irs = into_txn_proof_gen_ir(block_trace, OtherBlockData(...))

`TxnProofGenIR = GenerationInputs` (protocol_decoder/src/types.rs:48); `BlockMetadata` / `BlockHashes` are
upstream plonky2_evm types and travel as opaque bytes.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import trace_protocol as tp
from ._lib import check, lib, take_buffer
from .partial_trie import PartialTrie, Reader


@dataclass
class BlockLevelData:
    """types.rs:62-67 (b_meta / b_hashes opaque)."""
    b_meta: bytes = b""
    b_hashes: bytes = b""
    withdrawals: List[Tuple[bytes, int]] = field(default_factory=list)   # (address, amount)


@dataclass
class OtherBlockData:
    """types.rs:51-55."""
    b_data: BlockLevelData = field(default_factory=BlockLevelData)
    checkpoint_state_trie_root: bytes = bytes(32)


@dataclass
class TrieInputs:
    state_trie: PartialTrie
    transactions_trie: PartialTrie
    receipts_trie: PartialTrie
    storage_tries: List[Tuple[bytes, PartialTrie]]


@dataclass
class TrieRoots:
    state_root: bytes
    transactions_root: bytes
    receipts_root: bytes


@dataclass
class GenerationInputs:
    """The fields the reference fills at decoding.rs:131-145."""
    txn_number_before: int
    gas_used_before: int
    gas_used_after: int
    signed_txn: Optional[bytes]
    withdrawals: List[Tuple[bytes, int]]
    tries: TrieInputs
    trie_roots_after: TrieRoots
    checkpoint_state_trie_root: bytes
    contract_code: Dict[bytes, bytes]
    block_metadata: bytes
    block_hashes: bytes


TxnProofGenIR = GenerationInputs


def _u32(v): return int(v).to_bytes(4, "little")
def _blob(b): return _u32(len(b)) + bytes(b)
def _u256(v): return int(v).to_bytes(32, "big")


def trace_to_binary(bt: "tp.BlockTrace", other: OtherBlockData, code_table: Optional[Dict[bytes, bytes]] = None) -> bytes:
    """The "BPGTRAC1" byte form of include/bpg.h.  `code_table` stands in for the reference's
    `ProcessingMeta::resolve_code_hash_fn` (code hash -> bytes for contracts that are read but not created
    in this block and not carried by the witness)."""
    if not isinstance(bt.trie_pre_images, tp.CombinedPreImages):
        raise NotImplementedError("separate trie pre-images are todo!() in the reference (processed_block_trace.rs:93-118)")
    o = [b"BPGTRAC1", _blob(bt.trie_pre_images.compact.bytes), _u32(len(bt.txn_info))]
    for t in bt.txn_info:
        o.append(_u32(len(t.traces)))
        for addr, tr in t.traces.items():
            flags = ((tr.balance is not None) | (tr.nonce is not None) << 1 | (tr.storage_read is not None) << 2
                     | (tr.storage_written is not None) << 3 | (bool(tr.self_destructed)) << 6)
            if tr.code_usage is not None:
                flags |= 16 if tr.code_usage.kind == "read" else 32
            o += [bytes(addr), bytes([flags])]
            if tr.balance is not None:
                o.append(_u256(tr.balance))
            if tr.nonce is not None:
                o.append(_u256(tr.nonce))
            if tr.storage_read is not None:
                o += [_u32(len(tr.storage_read)), *map(bytes, tr.storage_read)]
            if tr.storage_written is not None:
                o.append(_u32(len(tr.storage_written)))
                for k, v in tr.storage_written.items():
                    o += [bytes(k), _u256(v)]
            if tr.code_usage is not None:
                o.append(bytes(tr.code_usage.data) if tr.code_usage.kind == "read" else _blob(tr.code_usage.data))
        m = t.meta
        o += [_blob(m.byte_code), _blob(m.new_txn_trie_node_byte), _blob(m.new_receipt_trie_node_byte),
              int(m.gas_used).to_bytes(8, "little")]
    o += [bytes(other.checkpoint_state_trie_root), _blob(other.b_data.b_meta), _blob(other.b_data.b_hashes),
          _u32(len(other.b_data.withdrawals))]
    for a, amt in other.b_data.withdrawals:
        o += [bytes(a), _u256(amt)]
    code_table = code_table or {}
    o.append(_u32(len(code_table)))
    for h, c in code_table.items():
        o += [bytes(h), _blob(c)]
    return b"".join(o)


def _parse_ir(r: Reader) -> GenerationInputs:
    n_before, g_before, g_after = r.u256(), r.u256(), r.u256()
    has_txn = r.u8()
    signed = r.blob()
    wd = [(r.take(20), r.u256()) for _ in range(r.u32())]
    state, txn, rec = r.trie(), r.trie(), r.trie()
    storage = []
    for _ in range(r.u32()):
        h = r.take(32)
        storage.append((h, r.trie()))
    roots = TrieRoots(r.take(32), r.take(32), r.take(32))
    checkpoint = r.take(32)
    code = {}
    for _ in range(r.u32()):
        h = r.take(32)
        code[h] = r.blob()
    meta, hashes = r.blob(), r.blob()
    return GenerationInputs(n_before, g_before, g_after, signed if has_txn else None, wd,
                            TrieInputs(state, txn, rec, storage), roots, checkpoint, code, meta, hashes)


def generation_inputs_bytes(bt: "tp.BlockTrace", other: OtherBlockData, code_table=None) -> bytes:
    """`BlockTrace::into_txn_proof_gen_ir` as the library emits it: the "BPGGENI1" buffer (include/bpg.h) that
    bp_generate_txn_proof_gi / bp_prove_shard_gi take as it is."""
    L = lib()
    L.bp_decode_block_trace.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    raw = trace_to_binary(bt, other, code_table)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    check(L.bp_decode_block_trace(raw, len(raw), C.byref(out), C.byref(n)))
    return take_buffer(out, n)


def parse_generation_inputs(geni: bytes, with_final_root=False):
    r = Reader(geni)
    if r.take(8) != b"BPGGENI1":
        raise ValueError("bad magic")
    irs = [_parse_ir(r) for _ in range(r.u32())]
    final_root = r.take(32)
    assert r.done()
    return (irs, final_root) if with_final_root else irs


def into_txn_proof_gen_ir(bt: "tp.BlockTrace", other: OtherBlockData, code_table=None, with_final_root=False):
    """`BlockTrace::into_txn_proof_gen_ir(p_meta, other_data)` -> Vec<TxnProofGenIR>.  Raises BpgError
    (code -2) where the reference returns a TraceParsingError or panics on a malformed payload."""
    L = lib()
    L.bp_decode_block_trace.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    raw = trace_to_binary(bt, other, code_table)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    check(L.bp_decode_block_trace(raw, len(raw), C.byref(out), C.byref(n)))
    r = Reader(take_buffer(out, n))
    if r.take(8) != b"BPGGENI1":
        raise ValueError("bad magic")
    irs = [_parse_ir(r) for _ in range(r.u32())]
    final_root = r.take(32)
    assert r.done()
    return (irs, final_root) if with_final_root else irs
