"""AIR 6 (the absorbing side of Keccak-256: one 136-byte block per row) on the CPU: the rows of a message against
hashlib's Keccak... the oracle's permutation, its witness against the rows, its constraint list against the witness, and
its proofs against the PRODUCT's CPU verifier (csrc/air.hpp over the extension field) -- two independent statements of
the same 2587 constraints (the oracle states pad10*1 as three cases per byte, the product through a running sum of the
length flags).  GPU side: tests/test_gpu_keccak_sponge_air.py."""
import ctypes as C

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_FULL, COL_FINAL, COL_LEN, COL_BLOCK, COL_RATE, COL_CAP, COL_XORED, COL_UPDATED, N_COLS = 0, 1, 2, 138, 1226, 2314, 2330, 2364, 2414
MESSAGES = [b"", b"abc", b"q" * 135, b"r" * 136, b"s" * 137, b"The quick brown fox jumps over the lazy dog" * 9]


def rows_of(oracle, msgs, log_n):
    rows = np.zeros((1 << log_n, 44), dtype=np.uint64)
    k, digests = 0, []
    for m in msgs:
        d, r = oracle.keccak_sponge_rows(m)
        rows[k:k + len(r)] = r
        k += len(r)
        digests.append(d)
    assert k <= (1 << log_n)
    return rows, k, digests


def lanes(t, r, col0, n):
    return [int(t[col0 + 2 * l, r]) | (int(t[col0 + 2 * l + 1, r]) << 32) for l in range(n)]


def test_rows_of_messages_end_in_their_keccak256(oracle):
    """Known answers: the product's and the oracle's row builders agree, chain through the permutation, pad correctly at
    every boundary length, and the last row of a message leaves its Keccak-256 (compact.keccak256, itself pinned by the
    reference's golden MPT roots) in the updated state."""
    from proof_protocol_decoder_amd import compact, proof_gen as pg
    for m in MESSAGES:
        d, r = oracle.keccak_sponge_rows(m)
        d2, r2 = pg.keccak256_sponge_rows(m)
        assert d == d2 == compact.keccak256(m) and r.tolist() == r2
        assert [int(x) for x in r[:, 0]] == [1] * (len(m) // 136) + [2] and int(r[-1, 1]) == len(m) % 136
        assert not r[0, 19:].any()                                   # a message starts from the zero state
        blocks = b"".join(int(w).to_bytes(8, "little") for row in r for w in row[2:19])
        padded = bytearray(m) + b"\x00" * (136 - len(m) % 136)
        padded[len(m)] ^= 0x01
        padded[-1] ^= 0x80
        assert blocks == bytes(padded)
    rows, k, digests = rows_of(oracle, MESSAGES, 4)
    t = oracle.keccak_sponge_trace(4, inputs=rows)
    assert t.shape == (N_COLS, 16) and (t[:COL_CAP] <= 1).all() and (t[COL_CAP:] < np.uint64(1 << 32)).all()
    r = 0
    for m, d in zip(MESSAGES, digests):
        n = len(m) // 136 + 1
        last = r + n - 1
        out = lanes(t, last, COL_UPDATED, 4)
        assert b"".join(x.to_bytes(8, "little") for x in out) == d
        for i in range(r, last):                                     # chaining inside the message
            assert lanes(t, i, COL_UPDATED, 25) == [sum(int(t[COL_RATE + 64 * l + z, i + 1]) << z for z in range(64)) for l in range(17)] + \
                [int(t[COL_CAP + 2 * l, i + 1]) | (int(t[COL_CAP + 2 * l + 1, i + 1]) << 32) for l in range(8)]
        assert int(t[COL_LEN + len(m) % 136, last]) == 1 and int(t[COL_LEN:COL_BLOCK, last].sum()) == 1
        r += n
    assert not t[:, k:].any()                                        # the rest: padding rows
    # the (xored rate, capacity) of a row is what its Keccak-f permutation takes in: the link to AIR 1
    inp = lanes(t, 0, COL_XORED, 17) + [int(t[COL_CAP + 2 * l, 0]) | (int(t[COL_CAP + 2 * l + 1, 0]) << 32) for l in range(8)]
    assert lanes(t, 0, COL_UPDATED, 25) == [int(x) for x in oracle.keccak_f(np.array(inp, dtype=np.uint64))]


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.KECCAK_SPONGE_COLS, air_id=oracle.AIR_KECCAK_SPONGE, **dict(dict(num_queries=6, pow_bits=6), **kw))


def prove(oracle, cfg, trace):
    tc = oracle.Committed.from_values(trace, cfg.rate_bits, cfg.cap_height)
    ch = oracle.PyChallenger()
    ch.observe(tc.cap())
    ctl = np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
    chv = ch.clone()
    return oracle.stark_prove(cfg, trace, ctl, ch, None, tc), ctl, chv


def product_verify(cfg, proof):
    """The product's CPU verifier through the C ABI (bp_stark_verify_air): host only, no GPU."""
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    pc = pkg.ops.stark_cfg(cfg.log_n, cfg.n_cols, n_const=cfg.n_const, deg_pow=cfg.deg_pow, rate_bits=cfg.rate_bits,
                           cap_height=cfg.cap_height, num_queries=cfg.num_queries, pow_bits=cfg.pow_bits,
                           arity_bits=cfg.arity_bits, final_poly_bits=cfg.final_poly_bits)
    raw = np.ascontiguousarray(proof, dtype="<u8").tobytes()
    return L.bp_stark_verify_air(cfg.air_id, C.byref(pc), None, raw, len(raw))


@pytest.mark.parametrize("seeded", [True, False], ids=["seeded", "messages"])
def test_oracle_proof_is_accepted_by_both_verifiers_and_tampering_is_not(oracle, seeded):
    log_n = 5
    cfg = small_cfg(oracle, log_n)
    trace = oracle.keccak_sponge_trace(log_n, seed=0x5907) if seeded else oracle.keccak_sponge_trace(log_n, inputs=rows_of(oracle, MESSAGES * 2, log_n)[0])
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 6
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with keccak_sponge_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
    syn = oracle.make_cfg(log_n, oracle.KECCAK_SPONGE_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


def _breaks():
    """(what, function that spoils a valid trace of MESSAGES in place).  Row map of the trace: 0 "", 1 "abc", 2 q*135,
    3-4 r*136, 5-6 s*137, 7-9 the fox (387 bytes: two full blocks and 115 bytes)."""
    def both_flags(t): t[COL_FINAL, 3] = 1
    def two_lengths(t): t[COL_LEN + 9, 1] = 1
    def no_length(t): t[COL_LEN + 3, 1] = 0
    def block_bit(t): t[COL_BLOCK + 50, 7] = 2
    def rate_bit(t): t[COL_RATE + 70, 8] = 2
    def first_pad_byte(t):                       # "abc": byte 3 must be 0x01
        t[COL_BLOCK + 8 * 3 + 1, 1] = 1
    def zero_pad_byte(t):                        # "abc": byte 60 must be zero
        t[COL_BLOCK + 8 * 60 + 5, 1] = 1
    def last_pad_bit(t):                         # byte 135 must carry 0x80
        t[COL_BLOCK + 8 * 135 + 7, 1] = 0
    def xored(t): t[COL_XORED + 11, 8] = int(t[COL_XORED + 11, 8]) ^ 4
    def chain(t):                                # row 8 continues row 7: its state before must be row 7's updated state
        t[COL_RATE + 200, 8] = 1 - int(t[COL_RATE + 200, 8])
    def chain_cap(t): t[COL_CAP + 5, 8] = int(t[COL_CAP + 5, 8]) ^ 1
    def fresh(t):                                # row 5 starts a message: its state before must be zero
        t[COL_CAP + 2, 5] = 7
    def full_then_nothing(t):                    # row 9 final -> full: the message would end on a full block
        t[COL_FINAL, 9], t[COL_FULL, 9] = 0, 1
    def first_row(t): t[COL_RATE + 1, 0] = 1
    return [("K1 both flags", both_flags), ("K3 two lengths", two_lengths), ("K3 no length", no_length), ("K4 block bit", block_bit),
            ("K5 rate bit", rate_bit), ("K6 the first pad byte", first_pad_byte), ("K6 a zero pad byte", zero_pad_byte),
            ("K6 the last pad bit", last_pad_bit), ("K7 the xored rate", xored), ("K8 chaining (rate)", chain),
            ("K8 chaining (capacity)", chain_cap), ("K8 a fresh message", fresh), ("K10 / K3 a message ends on a full block", full_then_nothing),
            ("K9 the first row", first_row)]


@pytest.mark.parametrize("what,spoil", _breaks(), ids=[w for w, _ in _breaks()])
def test_a_witness_that_breaks_one_family_yields_a_rejected_proof(oracle, what, spoil):
    cfg = small_cfg(oracle, 4)
    trace = oracle.keccak_sponge_trace(4, inputs=rows_of(oracle, MESSAGES, 4)[0])
    spoil(trace)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_the_keccak_sponge_air():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(6)
    assert d.name == b"keccak_sponge" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (2414, 2414, 12, 2)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (2587, 24, 36)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert sum(c for _, c, _, _ in fams[:11]) == 2587 and fams[8] == (2486, 50, 1, 2) and fams[10] == (2586, 1, 1, 2)
    # the twelve looking products (two into the Keccak-f table, ten into the logic table), interleaved transition / last row
    assert fams[11:] == [(2587, 12, 1, 3), (2588, 12, 3, 2)]
