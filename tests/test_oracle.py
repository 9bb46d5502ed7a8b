"""CPU tests of the oracle itself: what pins it.  No GPU needed."""
import numpy as np
import pytest

from pg_common import SMALL, ir_words
from util import P, bitrev_perm, rand_field


def test_field_against_python_bigints(oracle):
    rng = np.random.default_rng(1)
    a = rand_field(rng, 64)
    # multiplication / inversion through the NTT of a delta and through poseidon is indirect; check
    # the primitive directly with a length-2 NTT: (a0 + a1, a0 - a1)
    for x, y in zip(a[::2], a[1::2]):
        out = oracle.ntt(np.array([x, y], dtype=np.uint64))
        assert int(out[0]) == (int(x) + int(y)) % P and int(out[1]) == (int(x) - int(y)) % P


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8, 10])
def test_ntt_matches_naive_dft_and_inverts(oracle, log_n):
    rng = np.random.default_rng(log_n)
    a = rand_field(rng, 1 << log_n)
    assert (oracle.ntt(a) == oracle.dft_naive(a)).all()
    assert (oracle.ntt(a, inverse=True) == oracle.dft_naive(a, inverse=True)).all()
    assert (oracle.ntt(oracle.ntt(a), inverse=True) == a).all()


def test_lde_agrees_with_direct_evaluation(oracle):
    rng = np.random.default_rng(5)
    log_n, r = 5, 2
    vals = rand_field(rng, (3, 1 << log_n))
    coeffs, lde = oracle.lde_batch(vals, r)
    m = 1 << (log_n + r)
    w = pow(7, (P - 1) >> (log_n + r), P)
    for c in range(3):
        cs = [int(x) for x in coeffs[c]]
        for i in (0, 1, 5, m - 1):
            x = 7 * pow(w, i, P) % P
            assert int(lde[c][i]) == sum(cf * pow(x, j, P) for j, cf in enumerate(cs)) % P
    # the interpolant reproduces the trace on the subgroup
    assert (oracle.ntt_batch(coeffs) == vals).all()


def test_poseidon_known_answers(oracle):
    """Upstream test vectors (recalled; see tools/gen_poseidon_constants.py for provenance)."""
    z = oracle.poseidon(np.zeros(12, dtype=np.uint64))[0]
    assert [int(x) for x in z] == [
        0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca, 0xd7709673896996dc,
        0x46a84e87642f44ed, 0xd032648251ee0b3c, 0x1c687363b207df62, 0xdf8565563e8045fe, 0x40f5b37ff4254dae,
        0xd070f637b431067c, 0x1792b1c4342109d7]
    s = oracle.poseidon(np.arange(12, dtype=np.uint64))[0]
    assert [int(x) for x in s] == [
        0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2, 0x414754bfebd051f0,
        0xcb1f8980294a023f, 0x6eb2a9e4d54a9d0f, 0x1902bc3af467e056, 0xf045d5eafdc6021f, 0xe4150f77caaa3be5,
        0xc9bfd01d39b50cce, 0x5c0a27fcb0e1459b]


def test_poseidon_matches_generator_script(oracle):
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_poseidon_constants.py")
    spec = importlib.util.spec_from_file_location("gen_poseidon_constants", path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rc = gen.round_constants()
    gen.self_check(rc)
    rng = np.random.default_rng(3)
    st = rand_field(rng, (5, 12))
    got = oracle.poseidon(st)
    for i in range(5):
        assert [int(x) for x in got[i]] == gen.permute([int(x) for x in st[i]], rc)


def test_sponge_and_noop(oracle):
    rng = np.random.default_rng(4)
    x = rand_field(rng, 19)
    assert (oracle.hash_or_noop(x[:3]) == np.array(list(x[:3]) + [0], dtype=np.uint64)).all()
    # overwrite-mode sponge by hand
    st = np.zeros(12, dtype=np.uint64)
    for off in range(0, 19, 8):
        chunk = x[off:off + 8]
        st[:len(chunk)] = chunk
        st = oracle.poseidon(st)[0]
    assert (oracle.hash_no_pad(x) == st[:4]).all()


def test_merkle_paths_verify(oracle):
    rng = np.random.default_rng(6)
    cols = rand_field(rng, (7, 64))
    dig, cap = oracle.merkle_commit(cols, 2, bitrev_rows=True)
    br = bitrev_perm(6)
    L = oracle.lib()
    flat = np.ascontiguousarray(dig.reshape(-1))
    for leaf in (0, 1, 17, 63):
        path = np.empty(4 * 4, dtype=np.uint64)
        L.orc_merkle_path(flat, 6, 2, leaf, path)
        row = np.ascontiguousarray(cols[:, br[leaf]])
        assert L.orc_merkle_verify(row, 7, leaf, path, 6, 2, np.ascontiguousarray(cap.reshape(-1))) == 0
        row[0] ^= np.uint64(1)
        assert L.orc_merkle_verify(row, 7, leaf, path, 6, 2, np.ascontiguousarray(cap.reshape(-1))) != 0


def test_fri_fold_equals_coefficient_fold(oracle):
    """Evaluation-domain fold == upstream's reduce_with_powers on coefficients + coset FFT."""
    rng = np.random.default_rng(8)
    log_m, ab = 8, 4
    m = 1 << log_m
    coeffs = rand_field(rng, (2, m))  # ext polynomial: c0 plane, c1 plane
    beta = (int(rand_field(rng, 1, edge=False)[0]), int(rand_field(rng, 1, edge=False)[0]))
    L = oracle.lib()
    vals = coeffs.copy()
    for k in range(2):
        L.orc_coset_ntt(vals[k], log_m, 7)
    br = bitrev_perm(log_m)
    layer = np.ascontiguousarray(np.stack([vals[0][br], vals[1][br]], axis=1))
    folded = oracle.fri_fold(layer, ab, 7, beta)

    def emul(a, b):
        return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    new = np.zeros((2, m >> ab), dtype=np.uint64)
    for c in range(m >> ab):
        acc = (0, 0)
        for j in reversed(range(1 << ab)):
            acc = emul(acc, beta)
            acc = ((acc[0] + int(coeffs[0][c * 16 + j])) % P, (acc[1] + int(coeffs[1][c * 16 + j])) % P)
        new[0][c], new[1][c] = acc
    shift = pow(7, 16, P)
    for k in range(2):
        L.orc_coset_ntt(new[k], log_m - ab, shift)
    br2 = bitrev_perm(log_m - ab)
    assert (folded[:, 0] == new[0][br2]).all() and (folded[:, 1] == new[1][br2]).all()


def test_challenger_duplex_order(oracle):
    ch = oracle.PyChallenger()
    ch.observe(np.arange(1, 4, dtype=np.uint64))
    st = np.zeros(12, dtype=np.uint64)
    st[:3] = [1, 2, 3]
    st = oracle.poseidon(st)[0]
    assert [ch.challenge() for _ in range(3)] == [int(st[7]), int(st[6]), int(st[5])]


@pytest.mark.parametrize("case", [(6, 16, 0, 1, 1, 10), (9, 24, 0, 1, 1, 12), (7, 19, 5, 3, 3, 8)])
def test_stark_accepts_and_rejects_single_bit_flips(oracle, case):
    log_n, C, K, e, r, nq = case
    cfg = oracle.make_cfg(log_n, C, n_const=K, deg_pow=e, rate_bits=r, num_queries=nq, pow_bits=8)
    consts = oracle.synth_constants(7, log_n, K) if K else None
    cc = oracle.Committed.from_values(consts, r, 4) if K else None
    tr = oracle.synth_trace(1234, cfg, consts)
    tc = oracle.Committed.from_values(tr, r, 4)

    cap_words = 4 << 4

    def prologue(trace_cap):
        # a real verifier reads the trace cap from the proof it is checking (16-word header first)
        ch = oracle.PyChallenger()
        if K:
            ch.observe(cc.cap())
        ch.observe(trace_cap)
        return ch, np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
    ch, ctl = prologue(tc.cap())
    proof = oracle.stark_prove(cfg, tr, ctl, ch, cc, tc)
    cap = cc.cap() if K else None
    chv, ctlv = prologue(proof[16:16 + cap_words])
    assert oracle.stark_verify(cfg, proof, ctlv, chv, cap) == 0
    rng = np.random.default_rng(0)
    idx = list(rng.integers(16, proof.size, size=12)) + [16, 16 + cap_words, 16 + 2 * cap_words]
    for i in idx:
        bad = proof.copy()
        bad[i] ^= np.uint64(1 << int(rng.integers(0, 63)))
        chv, ctlv = prologue(bad[16:16 + cap_words])
        assert oracle.stark_verify(cfg, bad, ctlv, chv, cap) != 0, "flip at word %d accepted" % i
    # a witness that violates the AIR yields a proof the verifier rejects (the quotient is no longer
    # consistent with the constraints at zeta)
    tr2 = tr.copy()
    tr2[2, 5] ^= np.uint64(1)
    tc2 = oracle.Committed.from_values(tr2, r, 4)
    ch2, ctl2 = prologue(tc2.cap())
    bad_proof = oracle.stark_prove(cfg, tr2, ctl2, ch2, cc, tc2)
    chv, ctlv = prologue(bad_proof[16:16 + cap_words])
    assert oracle.stark_verify(cfg, bad_proof, ctlv, chv, cap) != 0


def test_txn_agg_block_chain_on_cpu(oracle):
    st = oracle.PgState(**SMALL)
    t0 = st.txn(ir_words(7, 0, 0x5EED0001))
    pv0 = t0[4 + 84:4 + 97]
    t1 = st.txn(ir_words(7, 1, 0x5EED0002, root_before=tuple(int(x) for x in pv0[8:12]), gas=(121, 150)))
    assert st.verify(t0) == 0 and st.verify(t1) == 0
    agg = st.agg(t0, False, t1, False)
    assert st.verify(agg) == 0
    with pytest.raises(RuntimeError):
        st.agg(t1, False, t0, False)  # not contiguous
    blk = st.block(None, agg)
    assert st.verify(blk) == 0
    bad = blk.copy()
    bad[-3] ^= np.uint64(4)
    assert st.verify(bad) != 0


def test_dummy_entry_does_not_advance_public_values(oracle):
    """IR version 2 = a padding entry (decoding.rs:484-520: txn numbers and gas before/after equal, tries
    untouched): proven and verified like a txn, aggregates with its neighbours, public values stand still."""
    st = oracle.PgState(**SMALL)
    real = ir_words(7, 0, 0x5EED0001)
    dummy = ir_words(7, 0, 0x44554D4D59000000, gas=(100, 100))
    dummy[1] = 2
    d, t = st.txn(dummy), st.txn(real)
    pv_d, pv_t = d[4 + 84:4 + 97], t[4 + 84:4 + 97]
    assert int(pv_d[0]) == int(pv_d[1]) == 0 and int(pv_d[2]) == int(pv_d[3]) == 100
    assert (pv_d[4:8] == pv_d[8:12]).all() and int(pv_t[1]) == 1 and (pv_t[4:8] != pv_t[8:12]).any()
    assert st.verify(d) == 0
    agg = st.agg(d, False, t, False)            # dummy first, then the only real txn (decoding.rs:336-340)
    assert st.verify(agg) == 0 and st.verify(st.block(None, agg)) == 0
    bad = ir_words(7, 0, 1, gas=(100, 121))
    bad[1] = 2
    with pytest.raises(RuntimeError):           # a dummy that uses gas is refused (decoding.rs:503-506)
        st.txn(bad)
