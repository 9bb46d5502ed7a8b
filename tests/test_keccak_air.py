"""AIR 1 (Keccak-f[1600], one round per row) on the CPU: the oracle's permutation against hashlib, its witness
against the permutation, its constraint list against the witness, and its proofs against the PRODUCT's CPU verifier
(csrc/air.hpp instantiated over the extension field) -- two independent statements of the same 2826 constraints,
cross-checked before any GPU is involved.  The GPU side of the same AIR is in tests/test_gpu_keccak_air.py."""
import ctypes as C
import hashlib

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_STEP, COL_A, COL_C, COL_CP, COL_AP, COL_APP, COL_APP0, COL_APPP = 0, 24, 74, 394, 714, 2314, 2364, 2428


def sha3_256_by_hand(permute, msg):
    """The SHA3-256 sponge around a given Keccak-f implementation (rate 136 bytes, domain bits 0x06, final 0x80)."""
    rate = 136
    m = bytearray(msg) + b"\x06"
    m += b"\x00" * (-len(m) % rate)
    m[-1] |= 0x80
    st = np.zeros(25, dtype=np.uint64)
    blocks = []
    for off in range(0, len(m), rate):
        st[:17] ^= np.frombuffer(bytes(m[off:off + rate]), dtype="<u8")
        blocks.append(st.copy())
        st = permute(st)
    return st[:4].astype("<u8").tobytes(), blocks


@pytest.mark.parametrize("msg", [b"", b"abc", b"q" * 135, b"r" * 136, b"The quick brown fox" * 20])
def test_oracle_permutation_reproduces_sha3_256(oracle, msg):
    digest, _ = sha3_256_by_hand(oracle.keccak_f, msg)
    assert digest == hashlib.sha3_256(msg).digest()


def test_oracle_permutation_known_answer(oracle):
    # Keccak-f[1600] of the all-zero state (the Keccak team's KeccakF-1600 intermediate values, first two lanes)
    z = oracle.keccak_f(np.zeros(25, dtype=np.uint64))
    assert (int(z[0]), int(z[1])) == (0xF1258F7940E1DDE7, 0x84D5CCF933C0478A)


def lanes_of(trace, row, col0):
    return np.array([int(trace[col0 + 2 * l, row]) | (int(trace[col0 + 2 * l + 1, row]) << 32) for l in range(25)],
                    dtype=np.uint64)


def test_trace_rows_are_the_rounds_of_the_permutation(oracle):
    """Known answers through the witness: the trace of the sponge blocks of a message ends, 24 rows later, in the
    state whose first four lanes are hashlib's SHA3-256."""
    msg = b"proof-protocol-decoder" * 9          # two blocks
    digest, blocks = sha3_256_by_hand(oracle.keccak_f, msg)
    log_n = 6                                     # 64 rows: permutations 0, 1 and 16 rows of a third
    inputs = np.zeros((3, 25), dtype=np.uint64)
    inputs[0], inputs[1] = blocks[0], blocks[1]
    t = oracle.keccak_trace(log_n, inputs=inputs)
    assert t.shape == (2431, 64) and (t < np.uint64(P)).all()
    for p, blk in enumerate(blocks):
        assert (lanes_of(t, 24 * p, COL_A) == blk).all()
        out = lanes_of(t, 24 * p + 23, COL_APP)
        out[0] = int(t[COL_APPP, 24 * p + 23]) | (int(t[COL_APPP + 1, 24 * p + 23]) << 32)
        assert (out == oracle.keccak_f(blk)).all()
    assert out[:4].astype("<u8").tobytes() == hashlib.sha3_256(msg).digest() == digest
    # one-hot flags, bits are bits, rows chain inside a permutation
    assert (t[COL_STEP:COL_STEP + 24].sum(axis=0) == 1).all()
    assert all(int(t[COL_STEP + r % 24, r]) == 1 for r in range(64))
    assert (t[COL_C:COL_APP] <= 1).all() and (t[COL_APP0:COL_APPP] <= 1).all()
    for r in range(63):
        if r % 24 != 23:
            nxt = lanes_of(t, r + 1, COL_A)
            cur = lanes_of(t, r, COL_APP)
            cur[0] = int(t[COL_APPP, r]) | (int(t[COL_APPP + 1, r]) << 32)
            assert (nxt == cur).all()


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.KECCAK_COLS, air_id=oracle.AIR_KECCAK_F,
                           **dict(dict(num_queries=6, pow_bits=6), **kw))


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


@pytest.mark.parametrize("log_n", [5, 7])
def test_oracle_proof_is_accepted_by_both_verifiers_and_tampering_is_not(oracle, log_n):
    cfg = small_cfg(oracle, log_n)
    trace = oracle.keccak_trace(log_n, seed=0xBEEF + log_n)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 1
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with keccak_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
        ch = oracle.PyChallenger()
        ch.observe(bad[16:16 + 64])
        c2 = np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
        assert oracle.stark_verify(cfg, bad, c2, ch, None) != 0
    # a proof claiming another AIR is refused outright
    syn = oracle.make_cfg(log_n, oracle.KECCAK_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


def test_queries_checked_by_several_threads_report_the_first_failing_query(oracle):
    """stark_verify checks the queries of a proof with >= 16 of them on up to four threads when the process is not busy
    proving; what it reports must be what the sequential loop reports: acceptance, and for a proof with TWO corrupted
    queries the failure of the lower one."""
    import proof_protocol_decoder_amd as pkg
    log_n = 5
    cfg = small_cfg(oracle, log_n, num_queries=20)
    trace = oracle.keccak_trace(log_n, seed=0xBEEF)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert product_verify(cfg, proof) == 0
    # the layout: the query section is the tail of the proof, 20 records of equal length
    q_words = None
    for cand in range(1, proof.size):
        if (proof.size - cand * 20) > 0 and int(proof[proof.size - cand * 20]) < (1 << (log_n + 1)) and cand > 2431:
            q_words = cand
            break
    assert q_words is not None
    bad = proof.copy()
    first = proof.size - 20 * q_words
    bad[first + 13 * q_words + 5] ^= np.uint64(1)       # a trace value of query 13
    bad[first + 6 * q_words + 9] ^= np.uint64(1)        # ... and of query 6
    assert product_verify(cfg, bad) != 0
    assert b"query 6:" in pkg.lib().bp_last_error()


# one wrong cell per constraint family: (column, row, what it breaks)
BREAKS = [(COL_STEP + 3, 3, "F1 flags rotate"), (COL_C + 64 * 2 + 17, 9, "F3/F5 theta"), (COL_CP + 64 * 4 + 63, 30, "F3/F4"),
          (COL_AP + 64 * 13 + 5, 12, "F4/F5/F6"), (COL_A + 2 * 7 + 1, 25, "F5/F9 input limb"),
          (COL_APP + 2 * 11, 40, "F6/F9 chi limb"), (COL_APP0 + 31, 2, "F7/F8"), (COL_APPP + 1, 7, "F8/F9 iota"),
          (2430, 23, "lookup filter is a bit"), (2430, 5, "lookup filter only on last-round rows")]


@pytest.mark.parametrize("col,row,what", BREAKS, ids=[b[2] for b in BREAKS])
def test_a_witness_that_breaks_one_family_yields_a_rejected_proof(oracle, col, row, what):
    """The prover does not check its witness; the verifier must.  Flip ONE cell of a valid trace (0 <-> 1 for a bit, a
    different limb value otherwise): the proof made from it is rejected by both verifiers at the constraint check."""
    log_n = 6
    cfg = small_cfg(oracle, log_n)
    trace = oracle.keccak_trace(log_n, seed=0x5EED)
    v = int(trace[col, row])
    trace[col, row] = (1 - v) if v <= 1 and COL_C <= col < COL_APP or COL_APP0 <= col < COL_APPP or col < 24 else (v ^ 0x40)
    if col == 2430 and row % 24 != 23:
        trace[col, row] = 1     # a well-formed bit, on a row that is not a permutation's last round
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_both_airs():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(1)
    assert d.name == b"keccak_f" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (2431, 2431, 4, 3)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (2826, 10, 11)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert sum(c for _, c, _, _ in fams[:10]) == 2826 and fams[0] == (0, 24, 2, 1) and fams[9] == (2776, 50, 1, 2)
    assert max(deg for _, _, _, deg in fams) == 3
    # the table's lookups follow (csrc/air.hpp, namespace ctl): the filter g (two constraints), the carried input h_c of
    # both challenge sets (fixed on first-round rows, carried on transitions), two filtered running products
    assert fams[10:] == [(2826, 2, 0, 2), (2828, 1, 0, 2), (2829, 1, 1, 2), (2830, 1, 0, 2), (2831, 1, 1, 2),
                         (2832, 1, 1, 3), (2833, 1, 3, 2), (2834, 1, 1, 3), (2835, 1, 3, 2)]
    s = pkg.ops.air_describe(0, n_cols=135, n_const=82, deg_pow=3)
    assert s.name == b"synthetic" and (s.fixed_n_cols, s.n_cols, s.degree, s.n_air_constraints) == (0, 135, 9, 99)
    from proof_protocol_decoder_amd._lib import BpgError
    with pytest.raises(BpgError):
        pkg.ops.air_describe(99)


def test_oracle_txn_with_the_keccak_flag_differs_only_through_table_3(oracle):
    """The IR flag 0x100 (Keccak table = AIR 1) in the oracle's generate_txn_proof: accepted, different from the
    all-synthetic proof of the same IR, refused when table 3 is not 2431 columns wide."""
    from pg_common import LOG_N, SMALL, WIDTH, ir_words
    st = oracle.PgState(**SMALL)
    width = list(WIDTH)
    width[3] = 2431
    iw = ir_words(5, 0, 0x5EED0042, width=tuple(width))
    plain = st.txn(iw)
    iw[1] = 0x101
    flagged = st.txn(iw)
    assert plain.shape == flagged.shape and (plain != flagged).any()
    assert st.verify(flagged) == 0
    bad = ir_words(5, 0, 0x5EED0042)
    bad[1] = 0x101
    with pytest.raises(Exception):
        st.txn(bad)
