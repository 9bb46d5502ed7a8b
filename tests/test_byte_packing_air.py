"""AIR 5 (byte packing: a big-endian sequence of 1..32 bytes and the 256-bit word it spells) on the CPU: the oracle's
witness against int.from_bytes, its constraint list against the witness, and its proofs against the PRODUCT's CPU
verifier (csrc/air.hpp over the extension field) -- two independent statements of the same 330 constraints (the oracle
writes the value length by length, the product regroups it by byte slot).  GPU side: tests/test_gpu_byte_packing_air.py."""
import ctypes as C

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
COL_READ, COL_LEN, COL_BITS, COL_VAL, N_COLS = 0, 1, 33, 289, 299


def slots_of(t, r):
    return bytes(sum(int(t[COL_BITS + 8 * s + b, r]) << b for b in range(8)) for s in range(32))


def check_row(t, r, rd, ln, data):
    assert int(t[COL_READ, r]) == rd
    assert [int(t[COL_LEN + j - 1, r]) for j in range(1, 33)] == [int(ln == j) for j in range(1, 33)]
    s = slots_of(t, r)
    assert s[:ln] == data[:ln] and s[ln:] == bytes(32 - ln)
    value = sum(int(t[COL_VAL + k, r]) << (32 * k) for k in range(8))
    assert value == int.from_bytes(data[:ln], "big")


def random_inputs(n, seed):
    rng = np.random.default_rng(seed)
    inp = rng.integers(0, 1 << 63, size=(n, 6), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 6), dtype=np.uint64)
    inp[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint64)
    inp[:, 1] = rng.integers(0, 33, size=n, dtype=np.uint64)
    return inp


def test_trace_rows_spell_big_endian_words(oracle):
    log_n = 7
    inp = random_inputs(1 << log_n, 31)
    for r, ln in enumerate([0, 1, 2, 3, 4, 5, 8, 16, 31, 32, 40]):     # every boundary; 40 is clipped to 32
        inp[r, 1] = ln
    inp[11, 2:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    inp[11, 1] = 32
    t = oracle.byte_packing_trace(log_n, inputs=inp)
    assert t.shape == (N_COLS, 128) and (t[:COL_VAL] <= 1).all() and (t[COL_VAL:] < np.uint64(1 << 32)).all()
    for r in range(128):
        data = b"".join(int(inp[r, 2 + w]).to_bytes(8, "little") for w in range(4))
        check_row(t, r, int(inp[r, 0]) & 1, min(int(inp[r, 1]), 32), data)
    s1 = oracle.byte_packing_trace(9, seed=0xB17E)
    assert (oracle.byte_packing_trace(9, seed=0xB17E) == s1).all() and (oracle.byte_packing_trace(9, seed=0xB17F) != s1).any()
    lens = sum(j * s1[COL_LEN + j - 1] for j in range(1, 33))
    assert set(int(x) for x in lens) == set(range(33))                  # every length occurs among 512 seeded rows
    for r in range(0, 512, 29):
        ln = int(lens[r])
        check_row(s1, r, int(s1[COL_READ, r]), ln, slots_of(s1, r))


def small_cfg(oracle, log_n, **kw):
    return oracle.make_cfg(log_n, oracle.BYTE_PACKING_COLS, air_id=oracle.AIR_BYTE_PACKING, **dict(dict(num_queries=6, pow_bits=6), **kw))


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


@pytest.mark.parametrize("log_n", [5, 9])
def test_oracle_proof_is_accepted_by_both_verifiers_and_tampering_is_not(oracle, log_n):
    cfg = small_cfg(oracle, log_n)
    trace = oracle.byte_packing_trace(log_n, seed=0xBEEF00 + log_n)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert int(proof[14]) == 5
    assert oracle.stark_verify(cfg, proof, ctl, chv.clone(), None) == 0
    assert product_verify(cfg, proof) == 0          # air.hpp over the extension field agrees with byte_packing_air.c at zeta
    for word in (20, proof.size // 2, proof.size - 5):
        bad = proof.copy()
        bad[word] ^= np.uint64(1 << 9)
        assert product_verify(cfg, bad) != 0
    syn = oracle.make_cfg(log_n, oracle.BYTE_PACKING_COLS, num_queries=6, pow_bits=6)
    assert oracle.stark_verify(syn, proof, ctl, chv.clone(), None) != 0


def _breaks():
    """(what, function that spoils a valid trace in place) -- one per constraint family"""
    def flag2(t): t[COL_LEN + 4, 3] = 2
    def two_lengths(t):
        ln = next(j for j in range(1, 33) if t[COL_LEN + j - 1, 5])
        t[COL_LEN + (ln % 32), 5] = 1
    def bit2(t): t[COL_BITS + 70, 9] = 2
    def byte_beyond(t):                         # row 11 has len = 5: a byte in slot 7
        t[COL_BITS + 8 * 7 + 1, 11] = 1
    def limb(t): t[COL_VAL + 0, 13] = int(t[COL_VAL + 0, 13]) ^ 0x100
    def byte_inside(t):                         # row 17 has len = 20: flipping a bit of slot 3 changes limb 4
        t[COL_BITS + 8 * 3 + 2, 17] = 1 - int(t[COL_BITS + 8 * 3 + 2, 17])
    def read3(t): t[COL_READ, 2] = 3
    return [("P1 flag not a bit", flag2), ("P2 two lengths", two_lengths), ("P3 slot bit not a bit", bit2),
            ("P4 a byte beyond the length", byte_beyond), ("P5 a value limb", limb), ("P5 a byte of the sequence", byte_inside),
            ("P0 is_read", read3)]


@pytest.mark.parametrize("what,spoil", _breaks(), ids=[w for w, _ in _breaks()])
def test_a_witness_that_breaks_one_family_yields_a_rejected_proof(oracle, what, spoil):
    log_n = 6
    cfg = small_cfg(oracle, log_n)
    inp = random_inputs(1 << log_n, 8)
    inp[:, 1] = 1 + (np.arange(1 << log_n) % 32)
    inp[11, 1], inp[17, 1] = 5, 20
    trace = oracle.byte_packing_trace(log_n, inputs=inp)
    spoil(trace)
    proof, ctl, chv = prove(oracle, cfg, trace)
    assert oracle.stark_verify(cfg, proof, ctl, chv, None) != 0
    assert product_verify(cfg, proof) != 0


def test_air_registry_describes_the_byte_packing_air():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    assert L.bp_air_count() == 9
    d = pkg.ops.air_describe(5)
    assert d.name == b"byte_packing" and (d.fixed_n_cols, d.n_cols, d.n_aux, d.degree) == (299, 299, 2, 2)
    assert (d.n_air_constraints, d.n_ctl_constraints, d.n_units) == (330, 4, 9)
    fams = [(f.first_index, f.count, f.kind, f.degree) for f in d.families[:d.n_families]]
    assert sum(c for _, c, _, _ in fams[:6]) == 330 and fams[5] == (322, 8, 0, 2)
