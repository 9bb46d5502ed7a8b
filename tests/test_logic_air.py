"""AIR 2 (logic: one AND / OR / XOR of two 256-bit words per row) on the CPU: the oracle's witness against Python's big
integers, its constraint list against the witness, and its proofs against the PRODUCT's CPU verifier (csrc/air.hpp
instantiated over the extension field) -- two independent statements of the same 524 constraints (the oracle selects
one polynomial per operation, the product folds them into p (a + b) + q ab), cross-checked before any GPU is
involved.  The GPU side of the same AIR is in tests/test_gpu_logic_air.py."""
import ctypes as C

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_OP, COL_IN0, COL_IN1, COL_RES, N_COLS = 0, 3, 259, 515, 524
OPS = {0: lambda a, b: 0, 1: lambda a, b: a & b, 2: lambda a, b: a | b, 3: lambda a, b: a ^ b}


def word_of(trace, row, col0):
    return sum(int(trace[col0 + i, row]) << i for i in range(256))


def result_of(trace, row):
    return sum(int(trace[COL_RES + k, row]) << (32 * k) for k in range(8))


def random_inputs(n, seed):
    rng = np.random.default_rng(seed)
    inp = rng.integers(0, 1 << 63, size=(n, 9), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 9), dtype=np.uint64)
    inp[:, 0] = rng.integers(0, 4, size=n, dtype=np.uint64)
    return inp


def test_trace_rows_are_the_operations_on_python_integers(oracle):
    log_n = 6
    inp = random_inputs(1 << log_n, 7)
    # edge operands: all zero, all one, equal, complementary
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    for r, (op, a, b) in enumerate([(1, 0, 0), (2, ones, ones), (3, ones, ones), (3, ones, 0), (1, ones, 0), (0, ones, ones)]):
        inp[r] = [op] + [a] * 4 + [b] * 4
    t = oracle.logic_trace(log_n, inputs=inp)
    assert t.shape == (N_COLS, 64) and (t[:COL_RES] <= 1).all() and (t[COL_RES:] < np.uint64(1 << 32)).all()
    for r in range(64):
        op = int(inp[r, 0])
        a = sum(int(inp[r, 1 + w]) << (64 * w) for w in range(4))
        b = sum(int(inp[r, 5 + w]) << (64 * w) for w in range(4))
        assert word_of(t, r, COL_IN0) == a and word_of(t, r, COL_IN1) == b
        assert [int(t[COL_OP + i, r]) for i in range(3)] == [int(op == 1), int(op == 2), int(op == 3)]
        assert result_of(t, r) == OPS[op](a, b), (r, op)
    # the seeded witness draws all four codes and is reproducible
    s1, s2 = oracle.logic_trace(8, seed=0xABCD), oracle.logic_trace(8, seed=0xABCD)
    assert (s1 == s2).all() and (oracle.logic_trace(8, seed=0xABCE) != s1).any()
    codes = s1[COL_OP] + 2 * s1[COL_OP + 1] + 3 * s1[COL_OP + 2]
    assert set(int(c) for c in codes) == {0, 1, 2, 3}
    for r in range(0, 256, 17):
        assert result_of(s1, r) == OPS[int(codes[r])](word_of(s1, r, COL_IN0), word_of(s1, r, COL_IN1))


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.LOGIC_COLS, air_id=oracle.AIR_LOGIC, **dict(dict(num_queries=6, pow_bits=6), **kw))


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
    trace = oracle.logic_trace(log_n, seed=0xFEED + log_n)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 2
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with logic_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
    # a proof claiming another AIR is refused outright
    syn = oracle.make_cfg(log_n, oracle.LOGIC_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


# one wrong cell per constraint family: (column, row, new value or None = flip the bit, what it breaks)
BREAKS = [(COL_OP, 3, 2, "L0 flag not a bit"), (COL_OP + 1, 5, None, "L1 two operations / L3"),
          (COL_IN0 + 77, 9, 2, "L2 operand bit not a bit"), (COL_IN1 + 200, 11, None, "L3 result of limb 6"),
          (COL_RES + 4, 20, None, "L3 result limb"), (523, 7, 2, "lookup filter is a bit")]


@pytest.mark.parametrize("col,row,val,what", BREAKS, ids=[b[3] for b in BREAKS])
def test_a_witness_that_breaks_one_family_yields_a_rejected_proof(oracle, col, row, val, what):
    """The prover does not check its witness; the verifier must.  Change ONE cell of a valid trace: the proof made from
    it is rejected by both verifiers at the constraint check."""
    log_n = 6
    cfg = small_cfg(oracle, log_n)
    inp = random_inputs(1 << log_n, 99)
    inp[:, 0] = 1 + (np.arange(1 << log_n) % 3)    # every row a real operation: flipping an operand bit changes a result
    inp[11, 0] = 3
    trace = oracle.logic_trace(log_n, inputs=inp)
    v = int(trace[col, row])
    trace[col, row] = val if val is not None else ((1 - v) if v <= 1 else v ^ 0x40)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_the_logic_air():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(2)
    assert d.name == b"logic" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (524, 524, 2, 3)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (524, 5, 8)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert fams[:4] == [(0, 3, 0, 2), (3, 1, 0, 2), (4, 512, 0, 2), (516, 8, 0, 3)]
    # the lookup keccak_sponge -> logic: the filter is a bit, two filtered running products
    assert fams[4:] == [(524, 1, 0, 2), (525, 1, 1, 3), (526, 1, 3, 2), (527, 1, 1, 3), (528, 1, 3, 2)]
    # a table of the wrong width is refused
    cfg = pkg.ops.stark_cfg(6, 523)
    assert L.bp_stark_verify_air(2, C.byref(cfg), None, b"\0" * 8, 8) != 0
