"""Host-side mirror of protocol_decoder::compact: decode an Erigon
compact block witness and compute the state-trie root.  Reference:
protocol_decoder/src/compact/compact_prestate_processing.rs:1240-1281 (process_compact_prestate)."""
import ctypes as C
from dataclasses import dataclass

from ._lib import check, lib, take_buffer

COMPATIBLE_HEADER_VERSION = 1  # processed_block_trace.rs:35


@dataclass(frozen=True)
class ProcessedCompactOutput:
    header_version: int
    state_root: bytes
    n_accounts: int
    n_storage_tries: int
    n_code: int
    n_accounts_missing_storage: int

    def version_is_compatible(self, target_ver=COMPATIBLE_HEADER_VERSION):
        return self.header_version == target_ver


def _bind():
    L = lib()
    L.bp_compact_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint8), C.c_char_p] + [C.POINTER(C.c_uint32)] * 4
    L.bp_compact_instructions.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.bp_compact_decode_full.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.bp_keccak256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.bp_keccak256.restype = None
    return L


def process_compact_prestate(witness: bytes) -> ProcessedCompactOutput:
    L = _bind()
    ver = C.c_uint8()
    root = C.create_string_buffer(32)
    na, ns, nc, miss = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(L.bp_compact_decode(witness, len(witness), C.byref(ver), root, C.byref(na), C.byref(ns), C.byref(nc),
                              C.byref(miss)))
    return ProcessedCompactOutput(ver.value, root.raw, na.value, ns.value, nc.value, miss.value)


@dataclass
class WitnessOutput:
    """`ProcessedCompactOutput{header, witness_out{tries{state, storage}, code}}`
    (compact_prestate_processing.rs:1243-1260): storage tries keyed by hashed account address."""
    header_version: int
    state: "object"      # partial_trie.PartialTrie
    storage: dict        # hashed account address (32 bytes) -> PartialTrie
    code: dict           # code hash -> bytes


def process_compact_prestate_full(witness: bytes) -> WitnessOutput:
    """The reference's whole output, not just the root: the decoded tries and the code map (bp_compact_decode_full)."""
    from .partial_trie import Reader
    L = _bind()
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    check(L.bp_compact_decode_full(witness, len(witness), C.byref(out), C.byref(n)))
    r = Reader(take_buffer(out, n))
    if r.take(8) != b"BPGCWIT1":
        raise ValueError("bad magic")
    ver = r.u8()
    state = r.trie()
    storage = {}
    for _ in range(r.u32()):
        h = r.take(32)
        storage[h] = r.trie()
    code = {}
    for _ in range(r.u32()):
        h = r.take(32)
        code[h] = r.blob()
    assert r.done()
    return WitnessOutput(ver, state, storage, code)


def parse_just_to_instructions(witness: bytes):
    """Instruction listing (one per line) -- compact_prestate_processing.rs:1283-1309."""
    L = _bind()
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    check(L.bp_compact_instructions(witness, len(witness), C.byref(out), C.byref(n)))
    return take_buffer(out, n).decode().splitlines()


def keccak256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    _bind().bp_keccak256(data, len(data), out)
    return out.raw
