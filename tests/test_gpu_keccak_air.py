"""AIR 1 (Keccak-f[1600], csrc/air.hpp) on the GPU against the oracle's independent statement of it
(oracle/keccak_air.c): the witness generator, K5 alone through bp_quotient_eval(air_id = 1, ...), and whole table
proofs byte for byte -- BASELINE configs[3] ("wide Keccak-f STARK trace, 2^20 rows") is the last one."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from util import P, coset_major_to_natural, rand_field, to_dev, to_host

pytestmark = pytest.mark.gpu
SEED = 0x5EED000000000004


@pytest.mark.parametrize("log_n", [4, 7, 12])
def test_witness_matches_oracle(bpg, oracle, log_n):
    want = oracle.keccak_trace(log_n, seed=SEED + log_n)
    got = to_host(bpg.ops.keccak_trace(log_n, seed=SEED + log_n))
    assert got.shape == want.shape == (2431, 1 << log_n) and (got == want).all()
    rng = np.random.default_rng(log_n)
    inputs = rng.integers(0, 1 << 64, size=(((1 << log_n) + 23) // 24, 25), dtype=np.uint64)
    inputs[0] = 0                                            # the all-zero state: the Keccak team's known answer
    want = oracle.keccak_trace(log_n, inputs=inputs)
    got = to_host(bpg.ops.keccak_trace(log_n, inputs=to_dev(inputs)))
    assert (got == want).all()
    if log_n >= 5:
        assert int(got[2428, 23]) | (int(got[2429, 23]) << 32) == 0xF1258F7940E1DDE7


@pytest.mark.parametrize("log_n", [5, 9, 12])
def test_quotient_eval_matches_oracle(bpg, oracle, log_n):
    """K5 alone on AIR 1: random LDE matrices (the constraints are evaluated on whatever is there -- on the coset the
    'bit' columns are arbitrary field elements), fixed challenges.  These heights spread the units over grid.y and
    sum the partial results; the one-pass form (grid.y = 1) runs in the 2^17 / 2^20-row proofs below."""
    rng = np.random.default_rng(700 + log_n)
    rows = (1 << log_n) << 1
    trace = rand_field(rng, (2431, rows))
    aux = rand_field(rng, (4, rows))
    ctl, alphas = rand_field(rng, (4,)), rand_field(rng, (2,))
    want = oracle.quotient_values(oracle.make_cfg(log_n, 2431, air_id=1), None, trace, aux, ctl, alphas[0], alphas[1])
    idx = coset_major_to_natural(log_n, 1)

    def to_cm(mat):
        cm = np.empty_like(mat)
        cm[:, idx] = mat
        return to_dev(cm)
    got = bpg.ops.quotient_eval(bpg.ops.stark_cfg(log_n, 2431), to_cm(trace), to_cm(aux), None, ctl, alphas, air_id=1)
    assert (to_host(got)[:, idx] == want).all()


def oracle_proof(oracle, log_n, nq, pb, seed):
    cfg = oracle.make_cfg(log_n, 2431, num_queries=nq, pow_bits=pb, air_id=1)
    tr = oracle.keccak_trace(log_n, seed=seed)
    tc = oracle.Committed.from_values(tr, 1, 4)
    ch = oracle.PyChallenger()
    ch.observe(tc.cap())
    ctl = np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
    chv = ch.clone()
    return cfg, oracle.stark_prove(cfg, tr, ctl, ch, None, tc), ctl, chv


def product_verify(bpg, pc, proof):
    raw = np.ascontiguousarray(proof, dtype="<u8").tobytes()
    return bpg.lib().bp_stark_verify_air(1, C.byref(pc), None, raw, len(raw))


@pytest.mark.parametrize("log_n,nq,pb,loaded", [(5, 6, 6, 0), (8, 20, 10, 1), (11, 84, 16, 0), (11, 84, 16, 1), (14, 84, 16, 0)])
def test_table_proof_bit_exact(bpg, oracle, log_n, nq, pb, loaded):
    """prove -> verify, bit-flip rejection, HIP bytes == oracle bytes.  2^14 rows is the S1 Keccak table height
    (constants.rs:12: the range starts at 14).  loaded: K5 in ONE pass (all six units and the CTL part by one
    workgroup row), as the library runs it while several provers share the device."""
    cfg, want, ctl, chv = oracle_proof(oracle, log_n, nq, pb, SEED)
    pc = bpg.ops.stark_cfg(log_n, 2431, num_queries=nq, pow_bits=pb)
    bpg.lib().bp_tune_assume_loaded(loaded)
    try:
        got = bpg.ops.stark_prove_air(1, pc, SEED)
    finally:
        bpg.lib().bp_tune_assume_loaded(-1)
    assert got.shape == want.shape and int(got[14]) == 1
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at word %d of %d" % (bad[0], want.size)
    assert oracle.stark_verify(cfg, got, ctl, chv, None) == 0
    assert product_verify(bpg, pc, got) == 0
    flipped = got.copy()
    flipped[got.size // 2] ^= np.uint64(1 << 21)
    assert product_verify(bpg, pc, flipped) != 0


def test_wrong_shapes_for_the_air_are_refused(bpg):
    from proof_protocol_decoder_amd._lib import BpgError
    for kw in (dict(n_cols=2432), dict(n_cols=2431, n_const=2), dict(n_cols=2431, deg_pow=3, rate_bits=3)):
        cfg = bpg.ops.stark_cfg(6, kw.pop("n_cols"), num_queries=6, pow_bits=6, **kw)
        with pytest.raises(BpgError, match="keccak_f"):
            bpg.ops.stark_prove_air(1, cfg, 1)
    with pytest.raises(BpgError, match="unknown air_id"):
        bpg.ops.stark_prove_air(9, bpg.ops.stark_cfg(6, 16), 1)


@pytest.mark.parametrize("log_n", [17, 20])
def test_keccak_f_trace_at_baseline_size_matches_the_oracle(bpg, oracle, log_n):
    """BASELINE configs[3] as a REAL Keccak-f trace: 2^20 rows (43690 permutations and a cut one) x 2431 columns,
    rate 2, standard_fast_config, one GPU.  The oracle's proof of the same table was made once on the GPU box's host
    cores (tools/gen_cfg4_golden.py 20 keccak_f) and its sha256 is committed; the GPU proof must have it, verify
    under both verifiers, and stop verifying after a bit flip.  At this height the quotient is ONE pass: 8192
    workgroups, the alpha fold never leaves the registers."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    need = 8 * (1 << log_n) * 2431 * 6.5
    if free < need:
        pytest.skip("needs ~%.0f GB of free device memory, %.0f GB free" % (need / 1e9, free / 1e9))
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.json")))["tables"]
    key = "logn%d_keccak_f" % log_n
    if key not in gold:
        pytest.skip("no oracle digest for %s yet (tools/gen_cfg4_golden.py %d keccak_f)" % (key, log_n))
    pc = bpg.ops.stark_cfg(log_n, 2431)
    try:
        got = bpg.ops.stark_prove_air(1, pc, SEED)
    finally:
        bpg.lib().bp_release_cached_memory()
    want = gold[key]
    assert got.size == want["n_words"] and [int(x) for x in got[:6]] == want["head"] and [int(x) for x in got[-2:]] == want["tail"]
    assert hashlib.sha256(np.ascontiguousarray(got, dtype="<u8").tobytes()).hexdigest() == want["sha256"]
    assert product_verify(bpg, pc, got) == 0
    cfg = oracle.make_cfg(log_n, 2431, air_id=1)
    ch = oracle.PyChallenger()
    ch.observe(got[16:16 + 64])
    ctl = np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
    assert oracle.stark_verify(cfg, got, ctl, ch, None) == 0
    bad = got.copy()
    bad[got.size // 3] ^= np.uint64(1 << 17)
    assert product_verify(bpg, pc, bad) != 0
