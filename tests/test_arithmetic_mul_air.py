"""AIR 7 (x * y = z + 2^256 w on 256-bit words: a 32-column schoolbook product over 16-bit limbs with 21-bit carries) on
the CPU: the oracle's witness (product computed on 64-bit words) against Python's integers, its constraint list against
the witness, and its proofs against the PRODUCT's CPU verifier (csrc/air.hpp over the extension field).  GPU side:
tests/test_gpu_arithmetic_mul_air.py."""
import ctypes as C

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_MUL, COL_X, COL_Y, COL_Z, COL_W, COL_CARRY, N_COLS = 0, 1, 17, 33, 289, 545, 1217
M = 1 << 256


def check_row(t, r, mul, x, y):
    assert int(t[COL_MUL, r]) == mul
    assert sum(int(t[COL_X + k, r]) << (16 * k) for k in range(16)) == x
    assert sum(int(t[COL_Y + k, r]) << (16 * k) for k in range(16)) == y
    z = sum(int(t[COL_Z + i, r]) << i for i in range(256))
    w = sum(int(t[COL_W + i, r]) << i for i in range(256))
    assert z + (w << 256) == (x * y if mul else 0)
    carries = [sum(int(t[COL_CARRY + 21 * k + j, r]) << j for j in range(21)) for k in range(32)]
    assert carries[31] == 0 and (mul or not any(carries))


def words(v):
    return [(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)]


def random_inputs(n, seed):
    rng = np.random.default_rng(seed)
    inp = rng.integers(0, 1 << 63, size=(n, 9), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 9), dtype=np.uint64)
    inp[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint64)
    return inp


def test_trace_rows_are_products_of_python_integers(oracle):
    log_n = 6
    inp = random_inputs(1 << log_n, 41)
    edge = [(1, M - 1, M - 1), (1, 0, M - 1), (1, 1, 1), (1, 1 << 255, 2), (1, 0xFFFF, 0xFFFF), (0, M - 1, M - 1),
            (1, (1 << 128) - 1, (1 << 128) + 1)]
    for r, (m, x, y) in enumerate(edge):
        inp[r] = [m] + words(x) + words(y)
    t = oracle.arithmetic_mul_trace(log_n, inputs=inp)
    assert t.shape == (N_COLS, 64) and (t[COL_X:COL_Z] < np.uint64(1 << 16)).all() and (t[COL_Z:] <= 1).all()
    for r in range(64):
        x = sum(int(inp[r, 1 + w]) << (64 * w) for w in range(4))
        y = sum(int(inp[r, 5 + w]) << (64 * w) for w in range(4))
        check_row(t, r, int(inp[r, 0]) & 1, x, y)
    s1 = oracle.arithmetic_mul_trace(7, seed=0x77AA)
    assert (oracle.arithmetic_mul_trace(7, seed=0x77AA) == s1).all() and 0 < int(s1[COL_MUL].sum()) < 128
    for r in range(0, 128, 11):
        check_row(s1, r, int(s1[COL_MUL, r]), sum(int(s1[COL_X + k, r]) << (16 * k) for k in range(16)),
                  sum(int(s1[COL_Y + k, r]) << (16 * k) for k in range(16)))


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.ARITHMETIC_MUL_COLS, air_id=oracle.AIR_ARITHMETIC_MUL, **dict(dict(num_queries=6, pow_bits=6), **kw))


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
    trace = oracle.arithmetic_mul_trace(log_n, seed=0x3141 + log_n)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 7
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with arithmetic_mul_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
    syn = oracle.make_cfg(log_n, oracle.ARITHMETIC_MUL_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


# one wrong cell per constraint family: (column, row, new value or None = flip the bit, what it breaks)
BREAKS = [(COL_MUL, 3, 2, "U0 is_mul not a bit"), (COL_Z + 77, 5, 2, "U1 z bit not a bit"), (COL_W + 200, 7, 2, "U2 w bit not a bit"),
          (COL_CARRY + 21 * 9 + 4, 9, 2, "U3 carry bit not a bit"), (COL_X + 6, 11, None, "U4 an x limb"), (COL_Z + 0, 13, None, "U4 the lowest bit of z"),
          (COL_W + 255, 15, None, "U4 the top bit of w"), (COL_CARRY + 21 * 20 + 3, 17, None, "U4 a carry"), (COL_CARRY + 21 * 31, 19, None, "U5 / U4 the top carry")]


@pytest.mark.parametrize("col,row,val,what", BREAKS, ids=[b[3] for b in BREAKS])
def test_a_witness_that_breaks_one_family_yields_a_rejected_proof(oracle, col, row, val, what):
    log_n = 5
    cfg = small_cfg(oracle, log_n)
    inp = random_inputs(1 << log_n, 6)
    inp[:, 0] = 1
    trace = oracle.arithmetic_mul_trace(log_n, inputs=inp)
    v = int(trace[col, row])
    trace[col, row] = val if val is not None else ((1 - v) if v <= 1 else v ^ 0x40)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_the_multiplication_air():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(7)
    assert d.name == b"arithmetic_mul" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (1217, 1217, 1, 3)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (1218, 2, 8)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert sum(c for _, c, _, _ in fams[:6]) == 1218 and fams[4] == (1185, 32, 0, 3)


def test_a_transaction_s_arithmetic_table_proven_by_the_multiplication_air(oracle):
    """IR flag 0x4000: the arithmetic table (index 0, prover_state.rs:85-93) of a transaction is proven with AIR 7 instead
    of AIR 4 -- seeded products or the caller's.  The oracle's table proofs are accepted by its own verifier and by the
    product's CPU verifier (which takes the statement from the IR); both AIR flags at once are refused."""
    from pg_common import LOG_N, SMALL, WIDTH, ir_words
    from proof_protocol_decoder_amd import proof_gen as pg
    st = oracle.PgState(**SMALL)
    width = list(WIDTH)
    width[0] = 1217
    ir = ir_words(9, 0, 0x5EED0717, width=tuple(width))
    ir[1] |= 0x4000
    b = pg.ProverStateBuilder()
    for t, name in enumerate(pg.TABLES):
        getattr(b, "set_%s_circuit_size" % name)(range(SMALL["table_log_lo"][t], SMALL["table_log_hi"][t]))
    b.set(**{k: v for k, v in SMALL.items() if not k.startswith("table_")})
    tp = st.txn_tables(ir)
    assert st.verify_tables(tp) == 0
    pg.verify_txn_table_proofs(b.cfg, tp.tobytes(), np.array(ir, dtype=np.uint64).tobytes())
    assert int(tp[2 + 13 + 4]) == 7                            # the first table's header: AIR 7
    ops = [[1, 3, 0, 0, 0, 5, 0, 0, 0], [1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1, 2**64 - 1]]
    tp2 = st.txn_tables(ir, witness={0: ops})
    assert st.verify_tables(tp2) == 0 and (tp2 != tp).any()
    pg.verify_txn_table_proofs(b.cfg, tp2.tobytes(), np.array(ir, dtype=np.uint64).tobytes())
    both = list(ir)
    both[1] |= 0x800
    with pytest.raises(Exception):
        st.txn_tables(both)
    full = st.txn(ir)                                          # ... and the whole transaction proof on top of it
    assert st.verify(full) == 0
