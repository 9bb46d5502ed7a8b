"""Host-side view of the partial Merkle-Patricia tries that cross the C ABI (include/bpg.h, trie byte form):
what `eth_trie_utils::partial_trie::HashedPartialTrie` is to the reference's protocol_decoder.  The native
library (csrc/mpt.cpp) builds and mutates tries; this module only parses their byte form, walks them and
re-hashes them (the Yellow Paper node encoding, restated independently in Python so the tests can cross-check
the native root hashes)."""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

from .compact import keccak256

EMPTY_TRIE_HASH = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")
EMPTY_CODE_HASH = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")


def rlp_bytes(b: bytes) -> bytes:
    if len(b) == 1 and b[0] < 0x80:
        return bytes(b)
    return _len_prefix(len(b), 0x80) + bytes(b)


def rlp_list(items) -> bytes:
    body = b"".join(items)
    return _len_prefix(len(body), 0xC0) + body


def rlp_int(v: int) -> bytes:
    return rlp_bytes(v.to_bytes((v.bit_length() + 7) // 8, "big"))


def _len_prefix(n, base):
    if n < 56:
        return bytes([base + n])
    be = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([base + 55 + len(be)]) + be


def rlp_decode(b: bytes):
    """One item -> bytes or list (nested); raises ValueError on trailing bytes."""
    item, used = _rlp_item(b, 0)
    if used != len(b):
        raise ValueError("trailing bytes after the RLP item")
    return item


def _rlp_item(b, o):
    t = b[o]
    if t < 0x80:
        return bytes([t]), o + 1
    is_list, base = t >= 0xC0, 0xC0 if t >= 0xC0 else 0x80
    if t - base < 56:
        n, hdr = t - base, 1
    else:
        ll = t - base - 55
        n, hdr = int.from_bytes(b[o + 1:o + 1 + ll], "big"), 1 + ll
    body = b[o + hdr:o + hdr + n]
    if len(body) != n:
        raise ValueError("truncated RLP")
    if not is_list:
        return bytes(body), o + hdr + n
    out, p = [], 0
    while p < n:
        it, p = _rlp_item(body, p)
        out.append(it)
    return out, o + hdr + n


def hex_prefix(nibbles, leaf):
    flag = (2 if leaf else 0) + (len(nibbles) & 1)
    ns = ([flag, *nibbles] if len(nibbles) & 1 else [flag, 0, *nibbles])
    return bytes((ns[i] << 4) | ns[i + 1] for i in range(0, len(ns), 2))


def nibbles_of(b: bytes) -> Tuple[int, ...]:
    return tuple(x for byte in b for x in (byte >> 4, byte & 15))


@dataclass
class Node:
    kind: str                                   # "empty" | "hash" | "branch" | "extension" | "leaf"
    key: Tuple[int, ...] = ()
    value: bytes = b""
    hash: bytes = b""
    children: List[Optional["Node"]] = field(default_factory=list)

    def encode(self) -> bytes:
        if self.kind == "leaf":
            return rlp_list([rlp_bytes(hex_prefix(self.key, True)), rlp_bytes(self.value)])
        if self.kind == "extension":
            return rlp_list([rlp_bytes(hex_prefix(self.key, False)), _ref(self.children[0])])
        if self.kind == "branch":
            return rlp_list([_ref(c) for c in self.children] + [rlp_bytes(self.value) if self.value else b"\x80"])
        return b"\x80"


def _ref(n: Optional[Node]) -> bytes:
    if n is None or n.kind == "empty":
        return b"\x80"
    if n.kind == "hash":
        return rlp_bytes(n.hash)
    enc = n.encode()
    return enc if len(enc) < 32 else rlp_bytes(keccak256(enc))


def hashed_node_preimages(root: Optional[Node]) -> List[bytes]:
    """The RLP encodings a hasher of the (partial) trie runs Keccak-256 over: every node that its parent references by
    hash (an encoding of 32 bytes or more) and the root whatever its length; children before parents.  Hashed-out
    subtrees contribute nothing (their hash is already there)."""
    out: List[bytes] = []

    def walk(n, is_root):
        if n is None or n.kind in ("empty", "hash"):
            return
        for c in n.children:
            walk(c, False)
        enc = n.encode()
        if is_root or len(enc) >= 32:
            out.append(enc)
    walk(root, True)
    return out


@dataclass
class PartialTrie:
    root: Node

    @staticmethod
    def from_bytes(b: bytes) -> "PartialTrie":
        node, used = _parse(b, 0)
        if used != len(b):
            raise ValueError("trailing bytes after the trie")
        return PartialTrie(node)

    def hash(self) -> bytes:
        if self.root.kind == "empty":
            return EMPTY_TRIE_HASH
        if self.root.kind == "hash":
            return self.root.hash
        return keccak256(self.root.encode())

    def items(self):
        """(path nibbles, 'val' | 'hash', bytes) depth first -- PartialTrie::items()."""
        out = []

        def walk(n, path):
            if n is None or n.kind == "empty":
                return
            if n.kind == "hash":
                out.append((path, "hash", n.hash))
            elif n.kind == "leaf":
                out.append((path + n.key, "val", n.value))
            elif n.kind == "extension":
                walk(n.children[0], path + n.key)
            else:
                for i, c in enumerate(n.children):
                    walk(c, path + (i,))
        walk(self.root, ())
        return out

    def get(self, key_nibbles):
        n, k = self.root, tuple(key_nibbles)
        while True:
            if n is None or n.kind == "empty":
                return None
            if n.kind == "hash":
                raise KeyError("the key's path runs into a hashed-out node")
            if n.kind == "leaf":
                return n.value if n.key == k else None
            if n.kind == "extension":
                if k[:len(n.key)] != n.key:
                    return None
                n, k = n.children[0], k[len(n.key):]
            else:
                if not k:
                    return n.value or None
                n, k = n.children[k[0]], k[1:]


def _parse(b, o):
    tag = b[o]
    o += 1
    if tag == 0:
        return Node("empty"), o
    if tag == 1:
        return Node("hash", hash=bytes(b[o:o + 32])), o + 32
    if tag == 2:
        mask = int.from_bytes(b[o:o + 2], "little")
        vl = int.from_bytes(b[o + 2:o + 6], "little")
        value = bytes(b[o + 6:o + 6 + vl])
        o += 6 + vl
        ch = []
        for i in range(16):
            if mask >> i & 1:
                c, o = _parse(b, o)
                ch.append(c)
            else:
                ch.append(None)
        return Node("branch", value=value, children=ch), o
    if tag in (3, 4):
        nk = b[o]
        key = tuple(b[o + 1:o + 1 + nk])
        o += 1 + nk
        if tag == 3:
            c, o = _parse(b, o)
            return Node("extension", key=key, children=[c]), o
        vl = int.from_bytes(b[o:o + 4], "little")
        return Node("leaf", key=key, value=bytes(b[o + 4:o + 4 + vl])), o + 4 + vl
    raise ValueError("unknown trie node tag %d" % tag)


@dataclass
class AccountRlp:
    nonce: int
    balance: int
    storage_root: bytes
    code_hash: bytes

    @staticmethod
    def decode(b: bytes) -> "AccountRlp":
        f = rlp_decode(b)
        if not isinstance(f, list) or len(f) != 4:
            raise ValueError("not an account")
        return AccountRlp(int.from_bytes(f[0], "big"), int.from_bytes(f[1], "big"), f[2], f[3])

    def encode(self) -> bytes:
        return rlp_list([rlp_int(self.nonce), rlp_int(self.balance), rlp_bytes(self.storage_root), rlp_bytes(self.code_hash)])


class Reader:
    """Little-endian cursor over the byte layouts of include/bpg.h."""

    def __init__(self, b):
        self.b, self.o = bytes(b), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated buffer")
        out = self.b[self.o:self.o + n]
        self.o += n
        return out

    def u8(self): return self.take(1)[0]
    def u32(self): return int.from_bytes(self.take(4), "little")
    def u64(self): return int.from_bytes(self.take(8), "little")
    def blob(self): return self.take(self.u32())
    def u256(self): return int.from_bytes(self.take(32), "big")
    def trie(self): return PartialTrie.from_bytes(self.blob())
    def done(self): return self.o == len(self.b)
