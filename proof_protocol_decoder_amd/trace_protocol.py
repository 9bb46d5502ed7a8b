"""Host-side mirror of the trace-protocol payload: the JSON schema a
client sends to the prover scheduler, restated as dataclasses with the same field names and the same
serde conventions (externally tagged snake_case enums, hex byte strings).

Reference: protocol_decoder/src/trace_protocol.rs:39-204 (types), protocol_decoder/src/deserializers.rs
(ByteString: hex with an optional 0x / 0X prefix on input, always 0x-prefixed on output).
U256 / Address / H256 follow ethereum_types' serde (0x-prefixed; U256 minimal-length, the hashes
fixed-width) [UPSTREAM-UNVERIFIED: impl-serde is not vendored in the reference tree].

Only what the reference itself can process is wired to the native decoder: a `combined.compact`
pre-image goes to `compact.process_compact_prestate` (the reference's other pre-image variants are
`todo!()` -- processed_block_trace.rs:93-125 -- and raise NotImplementedError here too).
"""
import json
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

from . import compact as _compact


_HEX = re.compile(r"[0-9a-fA-F]*")


class TraceProtocolError(ValueError):
    """Malformed payload (serde's deserialisation error in the reference)."""


# --------------------------------------------------------------------------- scalars

def bytes_from_hex(s) -> bytes:
    """deserializers.rs:33-67.  The reference slices `&data[..2]` before looking for the prefix, so a
    string shorter than two characters panics there; here it is an ordinary error (empty -> b'')."""
    if not isinstance(s, str):
        raise TraceProtocolError("expected a hex encoded string with a prefix")
    body = s[2:] if s[:2] in ("0x", "0X") else s
    if len(body) % 2 or not _HEX.fullmatch(body):
        raise TraceProtocolError(f"invalid hex string {s!r}")
    return bytes.fromhex(body)


def bytes_to_hex(b: bytes) -> str:
    """deserializers.rs:70-79."""
    return "0x" + bytes(b).hex()


def _fixed(s, n, what) -> bytes:
    if not isinstance(s, str) or s[:2] != "0x":
        raise TraceProtocolError(f"{what}: expected a 0x-prefixed hex string")
    try:
        b = bytes.fromhex(s[2:])
    except ValueError:
        raise TraceProtocolError(f"{what}: invalid hex {s!r}") from None
    if len(b) != n:
        raise TraceProtocolError(f"{what}: expected {n} bytes, got {len(b)}")
    return b


def u256_from_json(s) -> int:
    if not isinstance(s, str) or s[:2] != "0x" or len(s) < 3 or len(s) > 66:
        raise TraceProtocolError(f"U256: expected 0x-prefixed hex of 1..64 digits, got {s!r}")
    try:
        return int(s[2:], 16)
    except ValueError:
        raise TraceProtocolError(f"U256: invalid hex {s!r}") from None


def u256_to_json(v: int) -> str:
    if not 0 <= v < 1 << 256:
        raise TraceProtocolError("U256 out of range")
    return hex(v)


# --------------------------------------------------------------------------- txn info

@dataclass
class ContractCodeUsage:
    """trace_protocol.rs:185-204: `read`(code hash) or `write`(new contract bytes)."""
    kind: str
    data: bytes

    def get_code_hash(self) -> bytes:
        return self.data if self.kind == "read" else _compact.keccak256(self.data)

    @staticmethod
    def from_json(o):
        if not isinstance(o, dict) or len(o) != 1:
            raise TraceProtocolError("code_usage: expected one of {read, write}")
        (k, v), = o.items()
        if k == "read":
            return ContractCodeUsage("read", _fixed(v, 32, "code hash"))
        if k == "write":
            return ContractCodeUsage("write", bytes_from_hex(v))
        raise TraceProtocolError(f"code_usage: unknown variant {k!r}")

    def to_json(self):
        return {self.kind: bytes_to_hex(self.data)}


@dataclass
class TxnTrace:
    """trace_protocol.rs:151-183; absent keys are None and are skipped on output."""
    balance: Optional[int] = None
    nonce: Optional[int] = None
    storage_read: Optional[List[bytes]] = None
    storage_written: Optional[Dict[bytes, int]] = None
    code_usage: Optional[ContractCodeUsage] = None
    self_destructed: Optional[bool] = None

    @staticmethod
    def from_json(o):
        _want_keys(o, set(), {"balance", "nonce", "storage_read", "storage_written", "code_usage",
                               "self_destructed"}, "TxnTrace")
        g = o.get
        sd = g("self_destructed")
        if sd is not None and not isinstance(sd, bool):
            raise TraceProtocolError("self_destructed: expected a bool")
        return TxnTrace(
            balance=None if g("balance") is None else u256_from_json(g("balance")),
            nonce=None if g("nonce") is None else u256_from_json(g("nonce")),
            storage_read=None if g("storage_read") is None else [_fixed(a, 32, "storage address")
                                                                 for a in _want_list(g("storage_read"), "storage_read")],
            storage_written=None if g("storage_written") is None else {
                _fixed(k, 32, "storage address"): u256_from_json(v)
                for k, v in _want_map(g("storage_written"), "storage_written").items()},
            code_usage=None if g("code_usage") is None else ContractCodeUsage.from_json(g("code_usage")),
            self_destructed=sd)

    def to_json(self):
        o = {}
        if self.balance is not None:
            o["balance"] = u256_to_json(self.balance)
        if self.nonce is not None:
            o["nonce"] = u256_to_json(self.nonce)
        if self.storage_read is not None:
            o["storage_read"] = [bytes_to_hex(a) for a in self.storage_read]
        if self.storage_written is not None:
            o["storage_written"] = {bytes_to_hex(k): u256_to_json(v) for k, v in self.storage_written.items()}
        if self.code_usage is not None:
            o["code_usage"] = self.code_usage.to_json()
        if self.self_destructed is not None:
            o["self_destructed"] = self.self_destructed
        return o


@dataclass
class TxnMeta:
    """trace_protocol.rs:124-145."""
    byte_code: bytes
    new_txn_trie_node_byte: bytes
    new_receipt_trie_node_byte: bytes
    gas_used: int

    @staticmethod
    def from_json(o):
        keys = {"byte_code", "new_txn_trie_node_byte", "new_receipt_trie_node_byte", "gas_used"}
        _want_keys(o, keys, keys, "TxnMeta")
        gas = o["gas_used"]
        if isinstance(gas, bool) or not isinstance(gas, int) or not 0 <= gas < 1 << 64:
            raise TraceProtocolError("gas_used: expected a u64")
        return TxnMeta(bytes_from_hex(o["byte_code"]), bytes_from_hex(o["new_txn_trie_node_byte"]),
                       bytes_from_hex(o["new_receipt_trie_node_byte"]), gas)

    def to_json(self):
        return {"byte_code": bytes_to_hex(self.byte_code),
                "new_txn_trie_node_byte": bytes_to_hex(self.new_txn_trie_node_byte),
                "new_receipt_trie_node_byte": bytes_to_hex(self.new_receipt_trie_node_byte),
                "gas_used": self.gas_used}


@dataclass
class TxnInfo:
    """trace_protocol.rs:110-122: per-address traces plus the txn-wide metadata."""
    traces: Dict[bytes, TxnTrace]
    meta: TxnMeta

    @staticmethod
    def from_json(o):
        _want_keys(o, {"traces", "meta"}, {"traces", "meta"}, "TxnInfo")
        return TxnInfo({_fixed(a, 20, "address"): TxnTrace.from_json(t)
                        for a, t in _want_map(o["traces"], "traces").items()}, TxnMeta.from_json(o["meta"]))

    def to_json(self):
        return {"traces": {bytes_to_hex(a): t.to_json() for a, t in self.traces.items()}, "meta": self.meta.to_json()}


# --------------------------------------------------------------------------- trie pre-images

@dataclass
class TrieCompact:
    """trace_protocol.rs:85-89: the compact witness bytes (a transparent ByteString)."""
    bytes: bytes


@dataclass
class TrieUncompressed:
    """trace_protocol.rs:80-83 (an empty struct in the reference)."""


@dataclass
class TrieDirect:
    """trace_protocol.rs:91-95: upstream's HashedPartialTrie serde form, carried opaquely."""
    raw: object


SeparateTriePreImage = Union[TrieUncompressed, TrieDirect]


def _separate_pre_image_from_json(o, what):
    k, v = _one_variant(o, what)
    if k == "uncompressed":
        _want_map(v, what)
        return TrieUncompressed()
    if k == "direct":
        return TrieDirect(v)
    raise TraceProtocolError(f"{what}: unknown variant {k!r}")


def _separate_pre_image_to_json(p):
    return {"uncompressed": {}} if isinstance(p, TrieUncompressed) else {"direct": p.raw}


@dataclass
class SeparateStorageTriesPreImage:
    """trace_protocol.rs:97-108: `single_trie` or `multiple_tries` keyed by hashed account address."""
    single_trie: Optional[TrieUncompressed] = None
    multiple_tries: Optional[Dict[bytes, SeparateTriePreImage]] = None


@dataclass
class SeparateTriePreImages:
    """trace_protocol.rs:62-67."""
    state: SeparateTriePreImage
    storage: SeparateStorageTriesPreImage


@dataclass
class CombinedPreImages:
    """trace_protocol.rs:78-82."""
    compact: TrieCompact


BlockTraceTriePreImages = Union[SeparateTriePreImages, CombinedPreImages]


def _pre_images_from_json(o):
    k, v = _one_variant(o, "trie_pre_images")
    if k == "combined":
        _want_keys(v, {"compact"}, {"compact"}, "combined")
        return CombinedPreImages(TrieCompact(bytes_from_hex(v["compact"])))
    if k == "separate":
        _want_keys(v, {"state", "storage"}, {"state", "storage"}, "separate")
        sk, sv = _one_variant(v["storage"], "storage")
        if sk == "single_trie":
            _want_map(sv, "single_trie")
            storage = SeparateStorageTriesPreImage(single_trie=TrieUncompressed())
        elif sk == "multiple_tries":
            storage = SeparateStorageTriesPreImage(multiple_tries={
                _fixed(h, 32, "hashed account address"): _separate_pre_image_from_json(p, "storage trie")
                for h, p in _want_map(sv, "multiple_tries").items()})
        else:
            raise TraceProtocolError(f"storage: unknown variant {sk!r}")
        return SeparateTriePreImages(_separate_pre_image_from_json(v["state"], "state"), storage)
    raise TraceProtocolError(f"trie_pre_images: unknown variant {k!r}")


def _pre_images_to_json(p):
    if isinstance(p, CombinedPreImages):
        return {"combined": {"compact": bytes_to_hex(p.compact.bytes)}}
    st = p.storage
    storage = {"single_trie": {}} if st.multiple_tries is None else {
        "multiple_tries": {bytes_to_hex(h): _separate_pre_image_to_json(t) for h, t in st.multiple_tries.items()}}
    return {"separate": {"state": _separate_pre_image_to_json(p.state), "storage": storage}}


# --------------------------------------------------------------------------- the payload

@dataclass
class BlockTrace:
    """trace_protocol.rs:39-48: everything needed to prove one block, as sent by the client."""
    trie_pre_images: BlockTraceTriePreImages
    txn_info: List[TxnInfo] = field(default_factory=list)

    @staticmethod
    def from_json(text: Union[str, bytes, dict]) -> "BlockTrace":
        if not isinstance(text, dict):
            try:
                text = json.loads(text)
            except json.JSONDecodeError as e:
                raise TraceProtocolError(f"invalid JSON: {e}") from None
        _want_keys(text, {"trie_pre_images", "txn_info"}, {"trie_pre_images", "txn_info"}, "BlockTrace")
        return BlockTrace(_pre_images_from_json(text["trie_pre_images"]),
                          [TxnInfo.from_json(t) for t in _want_list(text["txn_info"], "txn_info")])

    def to_json(self) -> dict:
        return {"trie_pre_images": _pre_images_to_json(self.trie_pre_images),
                "txn_info": [t.to_json() for t in self.txn_info]}

    def dumps(self) -> str:
        return json.dumps(self.to_json())

    def process_pre_images(self) -> _compact.ProcessedCompactOutput:
        """processed_block_trace.rs:84-125 (`process_block_trace_trie_pre_images`): only the combined
        compact form is implemented by the reference; it also insists on the witness header version
        (`:127-140`).  Everything else is `todo!()` there and NotImplementedError here."""
        if not isinstance(self.trie_pre_images, CombinedPreImages):
            raise NotImplementedError("separate trie pre-images are todo!() in the reference "
                                      "(processed_block_trace.rs:93-118)")
        out = _compact.process_compact_prestate(self.trie_pre_images.compact.bytes)
        if not out.version_is_compatible():
            raise TraceProtocolError(f"compact witness header version {out.header_version} is not the "
                                     f"supported version {_compact.COMPATIBLE_HEADER_VERSION}")
        return out

    def all_code_hashes(self) -> Dict[bytes, bytes]:
        """Contract code created inside the block, keyed by its hash: what the reference's
        `CodeHashResolving.extra_code_hash_mappings` collects (processed_block_trace.rs:40-60,170-180)."""
        found = {}
        for t in self.txn_info:
            for tr in t.traces.values():
                if tr.code_usage is not None and tr.code_usage.kind == "write":
                    found[tr.code_usage.get_code_hash()] = tr.code_usage.data
        return found


# --------------------------------------------------------------------------- helpers

def _want_map(o, what):
    if not isinstance(o, dict):
        raise TraceProtocolError(f"{what}: expected a map")
    return o


def _want_list(o, what):
    if not isinstance(o, list):
        raise TraceProtocolError(f"{what}: expected a sequence")
    return o


def _want_keys(o, required, allowed, what):
    _want_map(o, what)
    missing = required - o.keys()
    if missing:
        raise TraceProtocolError(f"{what}: missing field {sorted(missing)[0]!r}")
    # serde ignores unknown fields by default (no deny_unknown_fields in trace_protocol.rs)
    del allowed


def _one_variant(o, what):
    if isinstance(o, str):  # unit-like spelling is not valid for these newtype/struct variants
        raise TraceProtocolError(f"{what}: expected a map with a single variant key")
    _want_map(o, what)
    if len(o) != 1:
        raise TraceProtocolError(f"{what}: expected a map with a single variant key")
    (k, v), = o.items()
    return k, v
