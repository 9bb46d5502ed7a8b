"""AIR 3 (a memory log sorted by address, then timestamp) on the CPU: the oracle's witness against a Python model of a
memory, its constraint list against the witness, and its proofs against the PRODUCT's CPU verifier (csrc/air.hpp over
the extension field) -- two independent statements of the same 60 constraints.  GPU side: tests/test_gpu_memory_air.py."""
import ctypes as C

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_READ, COL_ADDR, COL_TS, COL_VAL, COL_CHG, COL_GAP, N_COLS = 0, 1, 2, 3, 11, 12, 45


def random_log(n, seed, n_addr=None):
    """A consistent log: random operations on a few addresses, replayed on a Python dict, then sorted."""
    rng = np.random.default_rng(seed)
    n_addr = n_addr or max(2, n // 5)
    addrs = np.sort(rng.choice(1 << 20, size=n_addr, replace=False))
    mem, ops, ts = {}, [], 0
    for _ in range(n):
        a = int(addrs[rng.integers(0, n_addr)])
        ts += int(rng.integers(1, 1000))
        if rng.integers(0, 2):
            ops.append((1, a, ts, mem.get(a, (0,) * 8)))
        else:
            v = tuple(int(x) for x in rng.integers(0, 1 << 32, size=8))
            mem[a] = v
            ops.append((0, a, ts, v))
    ops.sort(key=lambda o: (o[1], o[2]))
    return np.array([[o[0], o[1], o[2], *o[3]] for o in ops], dtype=np.uint64)


def check_trace_is_a_memory(t):
    n = t.shape[1]
    assert (t[COL_READ] <= 1).all() and (t[COL_CHG] <= 1).all() and (t[COL_GAP:] <= 1).all()
    mem = {}
    for i in range(n):
        a, rd, v = int(t[COL_ADDR, i]), int(t[COL_READ, i]), tuple(int(x) for x in t[COL_VAL:COL_VAL + 8, i])
        if rd:
            assert v == mem.get(a, (0,) * 8), i
        else:
            mem[a] = v
        if i + 1 < n:
            gap = sum(int(t[COL_GAP + z, i]) << z for z in range(32))
            if int(t[COL_CHG, i]):
                assert int(t[COL_ADDR, i + 1]) == a + 1 + gap
            else:
                assert int(t[COL_ADDR, i + 1]) == a and int(t[COL_TS, i + 1]) == int(t[COL_TS, i]) + 1 + gap


def test_traces_are_consistent_memories(oracle):
    log = random_log(64, 3)
    t = oracle.memory_trace(6, inputs=log)
    assert t.shape == (N_COLS, 64) and (t < np.uint64(P)).all()
    assert (t[:COL_CHG].T == log).all()
    check_trace_is_a_memory(t)
    s1 = oracle.memory_trace(8, seed=0x77)
    check_trace_is_a_memory(s1)
    assert (oracle.memory_trace(8, seed=0x77) == s1).all() and (oracle.memory_trace(8, seed=0x78) != s1).any()
    assert 0 < int(s1[COL_READ].sum()) < 256 and int(s1[COL_CHG].sum()) == 63   # four operations per address


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.MEMORY_COLS, air_id=oracle.AIR_MEMORY, **dict(dict(num_queries=6, pow_bits=6), **kw))


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


@pytest.mark.parametrize("log_n,seeded", [(5, True), (7, False), (9, True)])
def test_oracle_proof_is_accepted_by_both_verifiers_and_tampering_is_not(oracle, log_n, seeded):
    cfg = small_cfg(oracle, log_n)
    trace = oracle.memory_trace(log_n, seed=0xFACE + log_n) if seeded else oracle.memory_trace(log_n, inputs=random_log(1 << log_n, log_n))
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 3
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with memory_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
    syn = oracle.make_cfg(log_n, oracle.MEMORY_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


def _broken_logs():
    """Logs that are NOT a memory: each breaks one thing the AIR exists to enforce."""
    base = random_log(64, 11, n_addr=6)
    out = []
    b = base.copy()                       # a read returns something else than what was written
    r = next(i for i in range(1, 64) if b[i, 0] == 1 and b[i, 1] == b[i - 1, 1])
    b[r, 3 + 2] ^= np.uint64(5)
    out.append(("M5 stale read", b))
    b = base.copy()                       # the first access of an address is a read of non-zero memory
    r = next(i for i in range(1, 64) if b[i, 1] != b[i - 1, 1])
    b[r, 0], b[r, 3] = 1, 7
    out.append(("M6 first read not zero", b))
    b = base.copy()                       # timestamps go backwards inside an address
    r = next(i for i in range(1, 64) if b[i, 1] == b[i - 1, 1])
    b[r, 2] = b[r - 1, 2] - np.uint64(1) if b[r, 0] == 0 else b[r - 1, 2]
    if b[r, 0] == 1:
        b[r, 0] = 0
    out.append(("M4 time does not advance", b))
    b = base.copy()                       # addresses not sorted
    r = next(i for i in range(1, 64) if b[i, 1] != b[i - 1, 1])
    b[r:, 1] = b[r - 1, 1] - np.uint64(1)
    b[r:, 0] = 0
    out.append(("M4 addresses go down", b))
    b = base.copy()
    b[0, 0], b[0, 3 + 7] = 1, 9           # the very first row reads a non-zero value
    out.append(("M7 first row", b))
    return out


@pytest.mark.parametrize("what,log", _broken_logs(), ids=[w for w, _ in _broken_logs()])
def test_a_log_that_is_not_a_memory_yields_a_rejected_proof(oracle, what, log):
    """The witness generator follows the log it is given; the verifier must refuse what is not a memory."""
    cfg = small_cfg(oracle, 6)
    trace = oracle.memory_trace(6, inputs=log)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


@pytest.mark.parametrize("col,row,val", [(COL_READ, 3, 2), (COL_CHG, 9, 2), (COL_GAP + 5, 20, 3), (COL_CHG, 17, None)],
                         ids=["M0", "M1", "M2", "M3-M4 flag flipped"])
def test_a_witness_cell_out_of_range_is_rejected(oracle, col, row, val):
    cfg = small_cfg(oracle, 6)
    trace = oracle.memory_trace(6, seed=5)
    trace[col, row] = val if val is not None else 1 - int(trace[col, row])
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_the_memory_air():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(3)
    assert d.name == b"memory" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (45, 45, 2, 3)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (60, 5, 1)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert sum(c for _, c, _, _ in fams[:8]) == 60 and fams[5] == (36, 8, 1, 3) and fams[7] == (52, 8, 2, 2)
