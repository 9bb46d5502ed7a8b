"""SURVEY.md section 8(f) rows 1 (full output) and 2 (txn IR producer).  CPU only.

Row 1 is pinned by the reference's golden witnesses: the tries returned by bp_compact_decode_full are re-hashed
by an independent Python encoder (proof_protocol_decoder_amd/partial_trie.py) and must give the six roots of
complex_test_payloads.rs:14-30, every account's storage trie must hash to its storage_root (:73-90).

Row 2 has no test in the reference, so it is pinned by invariants on a synthetic block: the
block's accounts live in a Python dict model; the witness is ENCODED here from that model (compact opcodes of
compact_prestate_processing.rs:744-875), the traces are applied to the model in Python, and after every
transaction the state root the native replay reports (decoding.rs:129) must equal the root of a trie built FROM
SCRATCH over the model (insert-everything recomputation).  Also: each IR's partial tries hash to the roots before
the txn, contain what the txn touches, the asserts of decoding.rs:498-505 hold for dummies, withdrawals land on
the right entry (:356-402).
"""
import json
import os

import pytest

from proof_protocol_decoder_amd import compact, decoding, partial_trie as pt, trace_protocol as tp
from proof_protocol_decoder_amd import BpgError

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "compact_witness_vectors.json")))
K = compact.keccak256


# ----------------------------------------------------------------------------- row 1: full output
@pytest.mark.parametrize("i", range(6))
def test_full_decode_rehashes_to_the_golden_roots(i):
    v = VEC["complex"][i]
    out = compact.process_compact_prestate_full(bytes.fromhex(v["witness_hex"]))
    assert out.header_version == 1
    assert out.state.hash().hex() == v["state_root"]                       # Python re-hash of the returned nodes
    assert out.state.hash() == compact.process_compact_prestate(bytes.fromhex(v["witness_hex"])).state_root
    n_acc = 0
    for path, kind, val in out.state.items():
        if kind != "val":
            continue
        acc = pt.AccountRlp.decode(val)
        n_acc += 1
        h_addr = bytes((path[j] << 4) | path[j + 1] for j in range(0, 64, 2))
        assert len(path) == 64
        if acc.storage_root != pt.EMPTY_TRIE_HASH:
            # complex_test_payloads.rs:73-90: the storage trie is there, keyed by the hashed address, and is the right one
            assert h_addr in out.storage and out.storage[h_addr].hash() == acc.storage_root
    assert n_acc > 0 and set(out.storage) <= {bytes((p[j] << 4) | p[j + 1] for j in range(0, 64, 2))
                                               for p, k, _ in out.state.items() if k == "val"}
    for h, code in out.code.items():
        assert K(code) == h


# ----------------------------------------------------------------------------- a from-scratch trie builder
def build_trie(items):
    """items: {nibble tuple: value bytes} -> pt.Node, built top down over the sorted keys (no inserts)."""
    def rec(keys, depth):
        if not keys:
            return None
        if len(keys) == 1:
            return pt.Node("leaf", key=keys[0][depth:], value=items[keys[0]])
        first, last = keys[0], keys[-1]
        c = 0
        while first[depth + c] == last[depth + c]:
            c += 1
        if c:
            return pt.Node("extension", key=first[depth:depth + c], children=[rec(keys, depth + c)])
        ch = [rec([k for k in keys if k[depth] == n], depth + 1) for n in range(16)]
        return pt.Node("branch", children=ch)
    root = rec(sorted(items), 0)
    return pt.PartialTrie(root if root is not None else pt.Node("empty"))


def u256_bytes(v):
    return v.to_bytes((v.bit_length() + 7) // 8, "big")


class Model:
    """addr -> [nonce, balance, {slot int: value int}, code bytes]"""

    def __init__(self, accounts):
        self.acc = {a: [n, b, dict(s), c] for a, (n, b, s, c) in accounts.items()}

    def storage_trie(self, a):
        st = {pt.nibbles_of(K(slot.to_bytes(32, "big"))): pt.rlp_int(v) for slot, v in self.acc[a][2].items() if v}
        return build_trie(st)

    def account_rlp(self, a):
        n, b, _, code = self.acc[a]
        return pt.AccountRlp(n, b, self.storage_trie(a).hash(), K(code) if code else pt.EMPTY_CODE_HASH).encode()

    def state_trie(self):
        return build_trie({pt.nibbles_of(K(a)): self.account_rlp(a) for a in self.acc})


# ----------------------------------------------------------------------------- compact witness encoder
def cbor_bytes(b):
    n = len(b)
    head = bytes([0x40 + n]) if n < 24 else bytes([0x58, n]) if n < 256 else bytes([0x59]) + n.to_bytes(2, "big")
    return head + bytes(b)


def cbor_uint(v):
    if v < 24:
        return bytes([v])
    for ai, w in ((24, 1), (25, 2), (26, 4), (27, 8)):
        if v < 1 << (8 * w):
            return bytes([ai]) + v.to_bytes(w, "big")
    raise ValueError


def key_bytes(nibs):
    nibs = list(nibs)
    assert nibs
    odd = len(nibs) & 1
    padded = nibs + [0] if odd else nibs
    return bytes([odd]) + bytes((padded[i] << 4) | padded[i + 1] for i in range(0, len(padded), 2))


def emit_node(n, leaf_op, hashed=()):
    """post-order opcodes of a trie node; leaf_op(node) emits the leaf's own operands + operator"""
    if n is None or n.kind == "empty":
        return b""
    if n.kind == "leaf":
        return leaf_op(n)
    if n.kind == "extension":
        return emit_node(n.children[0], leaf_op) + b"\x01" + cbor_bytes(key_bytes(n.key))
    mask, body = 0, b""
    for i, c in enumerate(n.children):            # earliest node -> lowest set bit
        if c is not None:
            mask |= 1 << i
            body += emit_node(c, leaf_op)
    return body + b"\x02" + cbor_uint(mask)


def encode_witness(model, hash_out_storage_of=()):
    by_hash = {pt.nibbles_of(K(a)): a for a in model.acc}

    def storage_leaf(n):
        raw = pt.rlp_decode(n.value)               # stored value is rlp(raw); the opcode carries the raw bytes
        return b"\x00" + cbor_bytes(key_bytes(n.key)) + cbor_bytes(raw)

    def account_leaf(path):
        def op(n):
            a = by_hash[path(n)]
            nonce, bal, storage, code = model.acc[a]
            out, flags = b"", 0
            if code:
                out += b"\x04" + cbor_bytes(code)
                flags |= 1
            st = model.storage_trie(a)
            if st.root.kind != "empty":
                flags |= 2
                out += (b"\x03" + st.hash()) if a in hash_out_storage_of else emit_node(st.root, storage_leaf)
            flags |= 4 | 8
            out += b"\x05" + cbor_bytes(key_bytes(n.key)) + bytes([flags]) + cbor_uint(nonce) + cbor_bytes(u256_bytes(bal))
            if code:
                out += cbor_uint(len(code))
            return out
        return op
    # the leaf's full path is needed to find its account: walk with the path
    state = model.state_trie()

    def walk(n, path):
        if n is None or n.kind == "empty":
            return b""
        if n.kind == "leaf":
            return account_leaf(lambda _n: path + _n.key)(n)
        if n.kind == "extension":
            return walk(n.children[0], path + n.key) + b"\x01" + cbor_bytes(key_bytes(n.key))
        mask, body = 0, b""
        for i, c in enumerate(n.children):
            if c is not None:
                mask |= 1 << i
                body += walk(c, path + (i,))
        return body + b"\x02" + cbor_uint(mask)
    return b"\x01" + walk(state.root, ())


# ----------------------------------------------------------------------------- the synthetic block
def addr(i):
    return bytes([0xA0 + i]) * 20


A, B, Cc, D, E, F = (addr(i) for i in range(6))
CODE_C, CODE_D, CODE_F = b"\x60\x01\x60\x02\x01" * 3, b"\xfe" * 40, b"\x60\x00\x80\xfd"


def fresh_model():
    return Model({
        A: (7, 10**18, {}, b""),
        B: (0, 5, {}, b""),
        Cc: (1, 0, {1: 11, 2: 22, 3: 33, 9: 99}, CODE_C),
        D: (1, 77, {5: 55, 6: 66}, CODE_D),
        E: (3, 1234, {1: 1, 2: 2, 3: 3}, b""),
    })


def receipt(i):
    # a legacy receipt: rlp([status, cumulative gas, bloom(256), logs])
    return pt.rlp_list([pt.rlp_int(1), pt.rlp_int(21000 * (i + 1)), pt.rlp_bytes(bytes(256)), pt.rlp_list([])])


def block(model):
    """[(TxnInfo, python delta applied to the model)] -- three transactions"""
    txns = []
    # txn 0: A -> B transfer
    t0 = tp.TxnInfo({A: tp.TxnTrace(balance=10**18 - 1000 - 21000, nonce=8), B: tp.TxnTrace(balance=1005)},
                    tp.TxnMeta(b"\x02\xf8\x70", b"\x01", receipt(0), 21000))

    def d0(m):
        m.acc[A][0], m.acc[A][1] = 8, 10**18 - 1000 - 21000
        m.acc[B][1] = 1005
    txns.append((t0, d0))
    # txn 1: A calls C: reads slot 1, writes slot 2, clears slot 3 (a delete), reads C's code
    s = lambda i: i.to_bytes(32, "big")
    t1 = tp.TxnInfo({A: tp.TxnTrace(balance=10**18 - 1000 - 71000, nonce=9),
                     Cc: tp.TxnTrace(storage_read=[s(1)], storage_written={s(2): 5, s(3): 0},
                                     code_usage=tp.ContractCodeUsage("read", K(CODE_C)))},
                    tp.TxnMeta(b"\x02\xf8\x71\x01", b"\x02", pt.rlp_bytes(b"\x02" + receipt(1)), 50000))

    def d1(m):
        m.acc[A][0], m.acc[A][1] = 9, 10**18 - 1000 - 71000
        m.acc[Cc][2][2] = 5
        del m.acc[Cc][2][3]
    txns.append((t1, d1))
    # txn 2: A deploys F (code write, one slot), D self-destructs
    t2 = tp.TxnInfo({A: tp.TxnTrace(nonce=10),
                     F: tp.TxnTrace(nonce=1, balance=3, storage_written={s(1): 7}, code_usage=tp.ContractCodeUsage("write", CODE_F)),
                     D: tp.TxnTrace(self_destructed=True)},
                    tp.TxnMeta(b"\x02\xf8\x72\x02\x03", b"\x03", receipt(2), 90000))

    def d2(m):
        m.acc[A][0] = 10
        m.acc[F] = [1, 3, {1: 7}, CODE_F]
        del m.acc[D]
    txns.append((t2, d2))
    return txns


def make_trace(model, infos, **kw):
    return tp.BlockTrace(tp.CombinedPreImages(tp.TrieCompact(encode_witness(model, **kw))), infos)


def txn_key(i):
    return pt.nibbles_of(pt.rlp_int(i))


def test_witness_encoder_round_trips_through_the_decoder():
    m = fresh_model()
    w = encode_witness(m, hash_out_storage_of=(E,))
    full = compact.process_compact_prestate_full(w)
    assert full.state.hash() == m.state_trie().hash()
    assert set(full.storage) == {K(Cc), K(D), K(E)} and full.storage[K(E)].root.kind == "hash"
    assert full.storage[K(Cc)].hash() == m.storage_trie(Cc).hash()
    assert full.code == {K(CODE_C): CODE_C, K(CODE_D): CODE_D}


def test_delta_replay_matches_from_scratch_recomputation():
    m = fresh_model()
    txns = block(m)
    other = decoding.OtherBlockData(decoding.BlockLevelData(b"meta-bytes", b"hashes-bytes", [(B, 100), (E, 1)]), b"\x11" * 32)
    irs, final_root = decoding.into_txn_proof_gen_ir(make_trace(m, [t for t, _ in txns], hash_out_storage_of=(E,)), other,
                                                     with_final_root=True)
    assert len(irs) == 4                                   # three transactions + the withdrawal dummy (decoding.rs:369-386)
    txn_trie, rec_trie, gas = {}, {}, 0
    for i, (info, delta) in enumerate(txns):
        ir = irs[i]
        before_root = m.state_trie().hash()
        assert ir.tries.state_trie.hash() == before_root                       # the sub-trie is a view of the state before
        assert ir.tries.transactions_trie.hash() == build_trie(txn_trie).hash()
        assert ir.tries.receipts_trie.hash() == build_trie(rec_trie).hash()
        for a, tr in info.traces.items():                                      # every touched account is reachable
            want = m.account_rlp(a) if a in m.acc else None
            assert ir.tries.state_trie.get(pt.nibbles_of(K(a))) == want
        st_by = dict(ir.tries.storage_tries)
        assert set(st_by) == {K(a) for a in info.traces}
        for a, tr in info.traces.items():
            if a in m.acc:
                assert st_by[K(a)].hash() == m.storage_trie(a).hash()
            for slot in list(tr.storage_read or []) + list((tr.storage_written or {}).keys()):
                v = m.acc.get(a, [0, 0, {}, b""])[2].get(int.from_bytes(slot, "big"))
                assert st_by[K(a)].get(pt.nibbles_of(K(slot))) == (pt.rlp_int(v) if v else None)
        assert (ir.txn_number_before, ir.gas_used_before) == (i, gas)
        gas += info.meta.gas_used
        assert ir.gas_used_after == gas and ir.signed_txn == info.meta.byte_code and ir.withdrawals == []
        delta(m)
        txn_trie[txn_key(i)] = info.meta.byte_code
        rec_trie[txn_key(i)] = receipt(i) if i != 1 else b"\x02" + receipt(1)    # the typed receipt is unwrapped (:335-343)
        assert ir.trie_roots_after.state_root == m.state_trie().hash(), "state root after txn %d" % i
        assert ir.trie_roots_after.transactions_root == build_trie(txn_trie).hash()
        assert ir.trie_roots_after.receipts_root == build_trie(rec_trie).hash()
        assert ir.checkpoint_state_trie_root == b"\x11" * 32 and ir.block_metadata == b"meta-bytes" and ir.block_hashes == b"hashes-bytes"
        assert pt.EMPTY_CODE_HASH in ir.contract_code and ir.contract_code[pt.EMPTY_CODE_HASH] == b""
    assert irs[1].contract_code[K(CODE_C)] == CODE_C and irs[2].contract_code[K(CODE_F)] == CODE_F
    # the withdrawal dummy: fully hashed-out tries of the state after the last txn, then the balances move
    wd = irs[3]
    assert wd.signed_txn is None and wd.tries.state_trie.root.kind == "hash" and wd.tries.state_trie.hash() == m.state_trie().hash()
    assert (wd.txn_number_before, wd.gas_used_before, wd.gas_used_after) == (3, gas, gas)     # decoding.rs:498-505
    assert wd.withdrawals == [(B, 100), (E, 1)]
    m.acc[B][1] += 100
    m.acc[E][1] += 1
    assert wd.trie_roots_after.state_root == m.state_trie().hash() == final_root
    assert wd.trie_roots_after.transactions_root == build_trie(txn_trie).hash()


def test_dummy_padding_of_short_blocks():
    other = decoding.OtherBlockData()
    m = fresh_model()
    irs = decoding.into_txn_proof_gen_ir(make_trace(m, []), other)             # empty block: two dummies (:315-325)
    assert len(irs) == 2 and all(ir.signed_txn is None and ir.tries.state_trie.root.kind == "hash" for ir in irs)
    assert all(ir.trie_roots_after.state_root == m.state_trie().hash() == ir.tries.state_trie.hash() for ir in irs)
    assert all((ir.txn_number_before, ir.gas_used_before, ir.gas_used_after) == (0, 0, 0) for ir in irs)
    assert irs[0].trie_roots_after.transactions_root == pt.EMPTY_TRIE_HASH
    t0, d0 = block(m)[0]
    before = m.state_trie().hash()
    irs = decoding.into_txn_proof_gen_ir(make_trace(m, [t0]), other)           # one txn, no withdrawals: dummy first (:334-338)
    assert len(irs) == 2 and irs[0].signed_txn is None and irs[1].signed_txn == t0.meta.byte_code
    assert irs[0].tries.state_trie.hash() == before == irs[0].trie_roots_after.state_root
    d0(m)
    assert irs[1].trie_roots_after.state_root == m.state_trie().hash()
    m2 = fresh_model()
    other_w = decoding.OtherBlockData(decoding.BlockLevelData(withdrawals=[(A, 9)]))
    irs = decoding.into_txn_proof_gen_ir(make_trace(m2, [t0]), other_w)        # with withdrawals: dummy last, carries them (:339-343, 388-398)
    assert len(irs) == 2 and irs[0].signed_txn == t0.meta.byte_code and irs[1].signed_txn is None
    d0(m2)
    assert irs[1].tries.state_trie.hash() == m2.state_trie().hash() and irs[1].withdrawals == [(A, 9)]
    m2.acc[A][1] += 9
    assert irs[1].trie_roots_after.state_root == m2.state_trie().hash()


def test_errors_are_statuses_not_crashes():
    m = fresh_model()
    other = decoding.OtherBlockData()
    s = lambda i: i.to_bytes(32, "big")
    # E's storage is hashed out in the witness: touching a slot of it cannot be served (MissingKeysCreatingSubPartialTrie)
    bad = tp.TxnInfo({E: tp.TxnTrace(storage_read=[s(1)])}, tp.TxnMeta(b"\x01", b"", receipt(0), 1))
    with pytest.raises(BpgError) as e:
        decoding.into_txn_proof_gen_ir(make_trace(m, [bad], hash_out_storage_of=(E,)), other)
    assert e.value.code == -2 and "storage" in str(e.value)
    # a code hash nobody can resolve
    bad = tp.TxnInfo({B: tp.TxnTrace(code_usage=tp.ContractCodeUsage("read", b"\x77" * 32))}, tp.TxnMeta(b"\x01", b"", receipt(0), 1))
    with pytest.raises(BpgError) as e:
        decoding.into_txn_proof_gen_ir(make_trace(m, [bad]), other)
    assert e.value.code == -2 and "code" in str(e.value)
    irs = decoding.into_txn_proof_gen_ir(make_trace(m, [bad]), other, code_table={b"\x77" * 32: b"\xaa"})
    assert irs[1].contract_code[b"\x77" * 32] == b"\xaa"
    # withdrawal to an account that is not in the state
    with pytest.raises(BpgError) as e:
        decoding.into_txn_proof_gen_ir(make_trace(m, []), decoding.OtherBlockData(decoding.BlockLevelData(withdrawals=[(F, 1)])))
    assert e.value.code == -2
    # receipt bytes that are neither a legacy receipt nor an RLP string
    bad = tp.TxnInfo({}, tp.TxnMeta(b"\x01", b"", b"\xc1\x80", 1))
    with pytest.raises(BpgError):
        decoding.into_txn_proof_gen_ir(make_trace(m, [bad]), other)
    # malformed byte form
    import ctypes as C
    from proof_protocol_decoder_amd._lib import lib
    raw = decoding.trace_to_binary(make_trace(m, [block(m)[0][0]]), other)
    L = lib()
    L.bp_decode_block_trace.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    for cut in (raw[:-1], raw[:40], b"XXXXXXXX" + raw[8:], raw + b"\x00"):
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        assert L.bp_decode_block_trace(cut, len(cut), C.byref(out), C.byref(n)) == -2


def test_trie_operations_against_from_scratch_builds():
    """insert / delete of csrc/mpt.cpp reach the same root as building the final key set from scratch, in any order
    (exercised through storage writes: insert new slots, overwrite, delete down to an empty trie)."""
    import random
    rng = random.Random(7)
    s = lambda i: i.to_bytes(32, "big")
    slots = {i: rng.randrange(1, 1 << 200) for i in rng.sample(range(1, 5000), 60)}
    m = Model({A: (1, 1, {}, b""), Cc: (1, 0, slots, CODE_C)})
    other = decoding.OtherBlockData()
    keys = list(slots)
    infos, expect = [], []
    cur = dict(slots)
    for rnd in range(6):
        writes = {}
        for k in rng.sample(keys, 15):
            writes[k] = 0 if rnd % 2 else rng.randrange(1, 1 << 64)
        for k in rng.sample(range(5000, 6000), 5):
            writes[k] = rng.randrange(1, 1 << 30)
            keys.append(k)
        if rnd == 5:
            writes = {k: 0 for k in cur}                    # clear everything: the trie must end up empty
        infos.append(tp.TxnInfo({Cc: tp.TxnTrace(storage_written={s(k): v for k, v in writes.items()})},
                                tp.TxnMeta(b"\x01", b"", receipt(rnd), 1)))
        for k, v in writes.items():
            if v:
                cur[k] = v
            else:
                cur.pop(k, None)
        expect.append(dict(cur))
    irs = decoding.into_txn_proof_gen_ir(make_trace(m, infos), other)
    for ir, st in zip(irs, expect):
        m.acc[Cc][2] = st
        assert ir.trie_roots_after.state_root == m.state_trie().hash()
    assert m.storage_trie(Cc).hash() == pt.EMPTY_TRIE_HASH


def test_generation_inputs_map_to_prover_irs():
    """decoded GenerationInputs -> the prover's IRs: counters from the decoded entries, dummies do not advance,
    the seed commits to the decoded roots (a different state transition = a different witness seed)."""
    from proof_protocol_decoder_amd.block_driver import irs_from_generation_inputs
    m = fresh_model()
    txns = block(m)
    other = decoding.OtherBlockData(decoding.BlockLevelData(withdrawals=[(B, 100)]))
    gis = decoding.into_txn_proof_gen_ir(make_trace(m, [t for t, _ in txns]), other)
    irs = irs_from_generation_inputs(gis, 17, (6, 5, 7, 7, 5, 6, 9), (16, 8, 24, 40, 16, 24, 8))
    assert [ir.dummy for ir in irs] == [False, False, False, True]
    assert [ir.txn_number_before for ir in irs] == [0, 1, 2, 3]
    assert [(ir.gas_used_before, ir.gas_used_after) for ir in irs] == [(0, 21000), (21000, 71000), (71000, 161000), (161000, 161000)]
    assert all(len(ir.to_bytes()) == 200 for ir in irs)
    # same payload but a different balance in txn 0: every seed from there on changes
    m2 = fresh_model()
    t2 = block(m2)
    t2[0][0].traces[B].balance += 1
    irs2 = irs_from_generation_inputs(decoding.into_txn_proof_gen_ir(make_trace(m2, [t for t, _ in t2]), other), 17,
                                      (6, 5, 7, 7, 5, 6, 9), (16, 8, 24, 40, 16, 24, 8))
    assert irs2[0].state_root_before == irs[0].state_root_before and all(a.seed != b.seed for a, b in zip(irs, irs2))


def test_the_library_derives_the_same_irs_from_generation_inputs_as_the_python_mirror():
    """csrc/gi.cpp (bp_gi_chain_start / bp_gi_entry_ir, the host half of bp_generate_txn_proof_gi): counters, state-root
    chain, seeds, grown table heights and AIR flags of every entry of a decoded block, for every combination of options,
    against block_driver.irs_from_generation_inputs (host only)."""
    from proof_protocol_decoder_amd.block_driver import GiOptions, gi_irs, irs_from_generation_inputs
    base_log, base_w = (6, 5, 7, 7, 5, 6, 9), (16, 8, 24, 40, 16, 24, 8)
    for wd in ([], [(B, 100)]):
        m = fresh_model()
        infos = [t for t, _ in block(m)]
        other = decoding.OtherBlockData(decoding.BlockLevelData(b"meta", b"hashes", wd), b"\x22" * 32)
        trace = make_trace(m, infos, hash_out_storage_of=(E,))
        geni = decoding.generation_inputs_bytes(trace, other)
        gis = decoding.parse_generation_inputs(geni)
        for kw in ({}, dict(keccak_air=True), dict(keccak_air=True, keccak_trie_nodes=True),
                   dict(keccak_air=True, memory_air=True), dict(keccak_air=True, byte_packing_air=True, keccak_sponge_air=True),
                   dict(keccak_air=True, keccak_trie_nodes=True, memory_air=True, byte_packing_air=True, keccak_sponge_air=True),
                   dict(keccak_air=True, keccak_sponge_air=True, logic_air=True),
                   dict(keccak_air=True, keccak_trie_nodes=True, memory_air=True, byte_packing_air=True, keccak_sponge_air=True, logic_air=True)):
            want = [ir.to_bytes() for ir in irs_from_generation_inputs(gis, 17, base_log, base_w, **kw)]
            assert gi_irs(geni, GiOptions.make(17, base_log, base_w, **kw)) == want, kw
    # a block of one transaction: the prepended dummy carries the counters of its position
    m = fresh_model()
    infos = [t for t, _ in block(m)][:1]
    geni = decoding.generation_inputs_bytes(make_trace(m, infos), decoding.OtherBlockData(decoding.BlockLevelData()))
    gis = decoding.parse_generation_inputs(geni)
    assert gis[0].signed_txn is None and gis[1].signed_txn is not None
    assert gi_irs(geni, GiOptions.make(3, base_log, base_w)) == [ir.to_bytes() for ir in irs_from_generation_inputs(gis, 3, base_log, base_w)]
    # options that make no sense, and buffers that are not generation inputs, are statuses
    import ctypes as C
    with pytest.raises(BpgError, match="needs BP_GI_KECCAK_AIR"):
        gi_irs(geni, GiOptions.make(3, base_log, base_w, memory_air=True))
    with pytest.raises(BpgError, match="needs BP_GI_KECCAK_SPONGE_AIR"):
        gi_irs(geni, GiOptions.make(3, base_log, base_w, keccak_air=True, logic_air=True))
    for bad in (geni[:-1], geni + b"\0", b"BPGGENI2" + geni[8:], geni[:40]):
        with pytest.raises(BpgError):
            gi_irs(bad, GiOptions.make(3, base_log, base_w))


def test_mutated_payloads_never_crash():
    """Client-supplied bytes: every mutation of a valid "BPGTRAC1" payload must come back as BP_OK or
    BP_ERR_INVALID_INPUT (bpg.h: nothing aborts across the ABI) -- 3000 random byte flips, truncations and
    length-field corruptions, including inside the compact witness."""
    import ctypes as C
    import random
    from proof_protocol_decoder_amd._lib import lib
    m = fresh_model()
    raw = decoding.trace_to_binary(make_trace(m, [t for t, _ in block(m)], hash_out_storage_of=(E,)),
                                   decoding.OtherBlockData(decoding.BlockLevelData(b"m", b"h", [(B, 5)])))
    L = lib()
    L.bp_decode_block_trace.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.bp_compact_decode_full.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    rng = random.Random(11)
    seen = {0: 0, -2: 0}
    wit = raw[12:12 + int.from_bytes(raw[8:12], "little")]
    for it in range(3000):
        b = bytearray(raw)
        kind = it % 4
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(8, len(b))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            b = b[:rng.randrange(0, len(b))]
        elif kind == 2:
            o = rng.randrange(8, len(b) - 4)
            b[o:o + 4] = rng.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\x00\x00\x00\x00"])
        else:
            o = rng.randrange(12, 12 + len(wit))              # inside the compact witness
            b[o] = rng.randrange(256)
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = L.bp_decode_block_trace(bytes(b), len(b), C.byref(out), C.byref(n))
        assert rc in (0, -2), rc
        seen[rc] += 1
        if rc == 0:
            L.bp_free_buffer(out)
        w = bytearray(wit)
        w[rng.randrange(len(w))] = rng.randrange(256)
        rc = L.bp_compact_decode_full(bytes(w), len(w), C.byref(out), C.byref(n))
        assert rc in (0, -2), rc
        if rc == 0:
            L.bp_free_buffer(out)
    assert seen[0] > 0 and seen[-2] > 1000


@pytest.mark.parametrize("i", range(6))
def test_golden_witnesses_as_pre_images_of_an_empty_block(i):
    """The reference's own golden witnesses through the IR producer: an empty block over each of them is padded to two
    dummy entries (decoding.rs:315-325) whose hashed-out state trie carries the GOLDEN state root, before and after."""
    v = VEC["complex"][i]
    bt = tp.BlockTrace(tp.CombinedPreImages(tp.TrieCompact(bytes.fromhex(v["witness_hex"]))), [])
    irs, final_root = decoding.into_txn_proof_gen_ir(bt, decoding.OtherBlockData(), with_final_root=True)
    assert len(irs) == 2 and final_root.hex() == v["state_root"]
    full = compact.process_compact_prestate_full(bytes.fromhex(v["witness_hex"]))
    for ir in irs:
        assert ir.signed_txn is None and ir.tries.state_trie.hash().hex() == v["state_root"] == ir.trie_roots_after.state_root.hex()
        assert ir.trie_roots_after.transactions_root == pt.EMPTY_TRIE_HASH and ir.contract_code == {}
        assert {h for h, _ in ir.tries.storage_tries} == set(full.storage)
        assert all(t.hash() == full.storage[h].hash() for h, t in ir.tries.storage_tries)


def test_keccak_work_of_a_decoded_transaction():
    """block_driver.keccak_inputs_of_generation_inputs: the permutation inputs of Keccak-256(signed_txn) and of every
    contract code, checked by running the sponge again in Python on top of the oracle's permutation (host only)."""
    import numpy as np
    from oracle import pyoracle
    from proof_protocol_decoder_amd import compact
    from proof_protocol_decoder_amd.block_driver import irs_from_generation_inputs, keccak_inputs_of_generation_inputs
    pyoracle.build()
    m = fresh_model()
    infos = [t for t, _ in block(m)]
    other = decoding.OtherBlockData(decoding.BlockLevelData(b"meta", b"hashes", []), b"\x22" * 32)
    gis = decoding.into_txn_proof_gen_ir(make_trace(m, infos), other)
    seen_code = False
    for g in gis:
        states = keccak_inputs_of_generation_inputs(g)
        msgs = ([bytes(g.signed_txn)] if g.signed_txn else []) + [bytes(g.contract_code[h]) for h in sorted(g.contract_code)]
        seen_code |= bool(g.contract_code)
        assert len(states) == sum(len(x) // 136 + 1 for x in msgs)
        k = 0
        for msg in msgs:
            n = len(msg) // 136 + 1
            out = pyoracle.keccak_f(np.array(states[k + n - 1], dtype=np.uint64))     # the last permutation's output
            assert out[:4].astype("<u8").tobytes() == compact.keccak256(msg)
            k += n
    irs = irs_from_generation_inputs(gis, 17, (6, 5, 7, 7, 5, 6, 9), (16, 8, 24, 40, 16, 24, 8), keccak_air=True)
    assert all(ir.keccak_air and ir.table_width[3] == 2431 and ir.table_log_n[3] >= 7 for ir in irs)
    assert all(24 * len(ir.keccak_inputs) <= (1 << ir.table_log_n[3]) for ir in irs)


def test_keccak_work_includes_the_hashing_of_the_partial_tries():
    """keccak_inputs_of_generation_inputs(trie_nodes=True): after the transaction and the code come the nodes of the
    entry's partial tries, children before parents; replaying the sponge over the listed permutations reproduces
    every node hash, and each trie's last digest is its root (host only)."""
    import numpy as np
    from oracle import pyoracle
    from proof_protocol_decoder_amd import compact
    from proof_protocol_decoder_amd.block_driver import irs_from_generation_inputs, keccak_inputs_of_generation_inputs
    from proof_protocol_decoder_amd.partial_trie import hashed_node_preimages
    pyoracle.build()
    m = fresh_model()
    infos = [t for t, _ in block(m)]
    other = decoding.OtherBlockData(decoding.BlockLevelData(b"meta", b"hashes", []), b"\x22" * 32)
    gis = decoding.into_txn_proof_gen_ir(make_trace(m, infos), other)
    grew = False
    for g in gis:
        base = keccak_inputs_of_generation_inputs(g)
        states = keccak_inputs_of_generation_inputs(g, trie_nodes=True)
        assert states[:len(base)] == base
        tries = [g.tries.state_trie, g.tries.transactions_trie, g.tries.receipts_trie] + [t for _, t in g.tries.storage_tries]
        k = len(base)
        for trie in tries:
            pre = hashed_node_preimages(trie.root)
            assert all(len(e) >= 32 for e in pre[:-1])          # only the root may be shorter
            for enc in pre:
                n = len(enc) // 136 + 1
                out = pyoracle.keccak_f(np.array(states[k + n - 1], dtype=np.uint64))
                assert out[:4].astype("<u8").tobytes() == compact.keccak256(enc)
                k += n
            if pre:
                assert compact.keccak256(pre[-1]) == trie.hash()
        assert k == len(states)
        grew |= len(states) > len(base)
    assert grew
    irs = irs_from_generation_inputs(gis, 17, (6, 5, 7, 7, 5, 6, 9), (16, 8, 24, 40, 16, 24, 8), keccak_air=True,
                                     keccak_trie_nodes=True)
    assert all(24 * len(ir.keccak_inputs) <= (1 << ir.table_log_n[3]) for ir in irs)
    assert max(ir.table_log_n[3] for ir in irs) > 7           # the tables grew to hold the trie hashing


def test_memory_and_byte_packing_work_of_a_decoded_transaction():
    """irs_from_generation_inputs(..., memory_air=True, byte_packing_air=True): the memory log and the byte-packing
    sequences of an entry are the traffic of exactly the bytes its Keccak table hashes -- the sequences spell the
    32-byte chunks big-endian, the log replays as a memory whose reads return the words those chunks spell, each read
    being the operation (address, timestamp) its sequence names (the lookup byte_packing -> memory), and the oracle's
    witnesses of both (padded to the table heights the IR asks for) satisfy what tests/test_memory_air.py and
    tests/test_byte_packing_air.py check (host only)."""
    import numpy as np
    from oracle import pyoracle
    from proof_protocol_decoder_amd.block_driver import (hashed_preimages_of_generation_inputs, irs_from_generation_inputs,
                                                         memory_and_byte_packing_work_of_preimages)
    import test_memory_air as tm
    pyoracle.build()
    m = fresh_model()
    infos = [t for t, _ in block(m)]
    other = decoding.OtherBlockData(decoding.BlockLevelData(b"meta", b"hashes", []), b"\x22" * 32)
    gis = decoding.into_txn_proof_gen_ir(make_trace(m, infos), other)
    irs = irs_from_generation_inputs(gis, 17, (6, 5, 7, 7, 5, 6, 9), (16, 8, 24, 40, 16, 24, 8), keccak_air=True,
                                     keccak_trie_nodes=True, memory_air=True, byte_packing_air=True)
    busy = 0
    for g, ir in zip(gis, irs):
        assert ir.memory_air and ir.byte_packing_air and ir.table_width[6] == 45 and ir.table_width[1] == 299
        pre = hashed_preimages_of_generation_inputs(g, trie_nodes=True)
        blob = b"".join(pre)
        wit = dict(ir.witness)
        log, seqs = np.array(wit[6], dtype=np.uint64).reshape(-1, 11), np.array(wit[1], dtype=np.uint64).reshape(-1, 6)
        chunks = [p[o:o + 32] for p in pre for o in range(0, len(p), 32)]
        assert len(log) == 2 * len(chunks) <= (1 << ir.table_log_n[6]) and len(seqs) <= (1 << ir.table_log_n[1])
        if not len(blob):
            continue
        busy += 1
        # sorted by (address, timestamp); the reads, in address order, return the words the chunks spell
        keys = [(int(r[1]), int(r[2])) for r in log]
        assert keys == sorted(keys) and len(set(keys)) == len(keys)
        words = [sum(int(r[3 + k]) << (32 * k) for k in range(8)) for r in log if r[0] == 1]
        assert words == [int.from_bytes(c, "big") for c in chunks]
        assert b"".join(w.to_bytes(len(c), "big") for w, c in zip(words, chunks)) == blob
        # the sequences are the strings cut into 32-byte chunks, and each names its read: (is_read, address, timestamp)
        assert len(seqs) == len(chunks)
        reads = {(int(r[1]), int(r[2])) for r in log if r[0] == 1}
        for s, c in zip(seqs, chunks):
            assert int(s[0]) & 1 == 1 and int(s[1]) & 0xFF == len(c)
            assert (int(s[1]) >> 8, int(s[0]) >> 8) in reads
            assert b"".join(int(w).to_bytes(8, "little") for w in s[2:])[:len(c)] == c
        # the oracle's witnesses of the padded inputs are a memory / spell the chunks
        n_rows = 1 << ir.table_log_n[6]
        padded = np.zeros((n_rows, 11), dtype=np.uint64)
        padded[:len(log)] = log
        last = log[-1].copy()
        last[0] = 1
        for i in range(len(log), n_rows):
            last[2] += np.uint64(1)
            padded[i] = last
        tm.check_trace_is_a_memory(pyoracle.memory_trace(ir.table_log_n[6], inputs=padded))
        bp_rows = 1 << ir.table_log_n[1]
        bp = np.zeros((bp_rows, 6), dtype=np.uint64)
        bp[:len(seqs)] = seqs
        t = pyoracle.byte_packing_trace(ir.table_log_n[1], inputs=bp)
        for r, c in enumerate(chunks):
            assert sum(int(t[289 + k, r]) << (32 * k) for k in range(8)) == int.from_bytes(c, "big")
            assert (int(t[297, r]), int(t[298, r])) == (r, r + 2)
    assert busy >= 2
