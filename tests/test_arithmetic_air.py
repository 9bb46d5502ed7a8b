"""AIR 4 (ADD / SUB / LT / GT on 256-bit words, sixteen 16-bit limbs, one carry chain) on the CPU: the oracle's witness
against Python's integers, its constraint list against the witness, and its proofs against the PRODUCT's CPU verifier
(csrc/air.hpp over the extension field) -- two independent statements of the same 294 constraints (the oracle writes
one limb equation per operation, the product collects the coefficients of x, y and z).  GPU side:
tests/test_gpu_arithmetic_air.py."""
import ctypes as C

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_OP, COL_X, COL_Y, COL_Z, COL_CARRY, COL_RES, N_COLS = 0, 4, 20, 36, 292, 308, 309
M = 1 << 256


def limbs_word(t, r, col0):
    return sum(int(t[col0 + k, r]) << (16 * k) for k in range(16))


def bits_word(t, r):
    return sum(int(t[COL_Z + i, r]) << i for i in range(256))


def check_row(t, r, op, x, y):
    assert limbs_word(t, r, COL_X) == x and limbs_word(t, r, COL_Y) == y
    assert [int(t[COL_OP + i, r]) for i in range(4)] == [int(op == i + 1) for i in range(4)]
    z, res = bits_word(t, r), int(t[COL_RES, r])
    if op == 1:
        assert z == (x + y) % M and res == 0 and int(t[COL_CARRY + 15, r]) == (x + y) // M
    elif op == 2:
        assert z == (x - y) % M and res == 0
    elif op == 3:
        assert res == int(x < y) and z == (x - y) % M
    elif op == 4:
        assert res == int(x > y) and z == (y - x) % M
    else:
        assert z == 0 and res == 0 and not t[COL_CARRY:COL_CARRY + 16, r].any()


def random_inputs(n, seed):
    rng = np.random.default_rng(seed)
    inp = rng.integers(0, 1 << 63, size=(n, 9), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 9), dtype=np.uint64)
    inp[:, 0] = rng.integers(0, 5, size=n, dtype=np.uint64)
    return inp


def words(v):
    return [(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)]


def test_trace_rows_are_the_operations_on_python_integers(oracle):
    log_n = 6
    inp = random_inputs(1 << log_n, 21)
    edge = [(1, M - 1, 1), (1, M - 1, M - 1), (2, 0, 1), (2, 5, 5), (3, 7, 7), (3, 0, M - 1), (3, M - 1, 0), (4, 7, 7),
            (4, 1 << 255, (1 << 255) - 1), (4, 0xFFFF, 0x10000), (1, 0xFFFF, 1), (0, M - 1, M - 1), (9, 3, 4)]
    for r, (op, x, y) in enumerate(edge):
        inp[r] = [op] + words(x) + words(y)
    t = oracle.arithmetic_trace(log_n, inputs=inp)
    assert t.shape == (N_COLS, 64) and (t[COL_X:COL_Z] < np.uint64(1 << 16)).all()
    assert (t[COL_Z:COL_RES + 1] <= 1).all() and (t[:COL_X] <= 1).all()
    for r in range(64):
        op = int(inp[r, 0]) if int(inp[r, 0]) <= 4 else 0
        x = sum(int(inp[r, 1 + w]) << (64 * w) for w in range(4))
        y = sum(int(inp[r, 5 + w]) << (64 * w) for w in range(4))
        check_row(t, r, op, x, y)
    s1 = oracle.arithmetic_trace(8, seed=0x1234)
    assert (oracle.arithmetic_trace(8, seed=0x1234) == s1).all() and (oracle.arithmetic_trace(8, seed=0x1235) != s1).any()
    codes = s1[COL_OP] + 2 * s1[COL_OP + 1] + 3 * s1[COL_OP + 2] + 4 * s1[COL_OP + 3]
    assert set(int(c) for c in codes) == {0, 1, 2, 3, 4}
    for r in range(0, 256, 13):
        check_row(s1, r, int(codes[r]), limbs_word(s1, r, COL_X), limbs_word(s1, r, COL_Y))


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.ARITHMETIC_COLS, air_id=oracle.AIR_ARITHMETIC, **dict(dict(num_queries=6, pow_bits=6), **kw))


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


@pytest.mark.parametrize("log_n", [5, 8])
def test_oracle_proof_is_accepted_by_both_verifiers_and_tampering_is_not(oracle, log_n):
    cfg = small_cfg(oracle, log_n)
    trace = oracle.arithmetic_trace(log_n, seed=0xA11CE + log_n)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 4
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with arithmetic_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
    syn = oracle.make_cfg(log_n, oracle.ARITHMETIC_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


# one wrong cell per constraint family: (column, row, new value or None = flip the bit, what it breaks)
BREAKS = [(COL_OP + 2, 3, 2, "A0 flag not a bit"), (COL_OP, 5, None, "A1 two operations / A4"), (COL_Z + 100, 9, 2, "A2 z bit not a bit"),
          (COL_CARRY + 7, 11, None, "A3-A4 a carry flipped"), (COL_X + 3, 13, None, "A4 an x limb changed"),
          (COL_Z + 255, 17, None, "A4 the top bit of z"), (COL_RES, 19, None, "A5 the comparison result")]


@pytest.mark.parametrize("col,row,val,what", BREAKS, ids=[b[3] for b in BREAKS])
def test_a_witness_that_breaks_one_family_yields_a_rejected_proof(oracle, col, row, val, what):
    log_n = 6
    cfg = small_cfg(oracle, log_n)
    inp = random_inputs(1 << log_n, 5)
    inp[:, 0] = 1 + (np.arange(1 << log_n) % 4)          # every row a real operation
    inp[5, 0] = 2                                        # row 5: sub; setting is_add as well makes two operations
    trace = oracle.arithmetic_trace(log_n, inputs=inp)
    v = int(trace[col, row])
    trace[col, row] = val if val is not None else ((1 - v) if v <= 1 else v ^ 0x40)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_the_arithmetic_air():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(4)
    assert d.name == b"arithmetic" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (309, 309, 1, 2)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (294, 2, 4)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert sum(c for _, c, _, _ in fams[:6]) == 294 and fams[4] == (277, 16, 0, 2)
