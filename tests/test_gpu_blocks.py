"""BASELINE configs[2] and configs[4] on one GPU: the 256-txn and the 1024-txn synthetic S1 block through
BlockDriver.prove_block_distributed at bp_config_default parameters (the reference's default table ranges,
constants.rs:6-18; standard_fast_config; recursion shape 2^13 x 135), then `python3 bench.py --gpus 2` as the
driver starts it (no launcher), two ranks sharing this box's one GPU.

The 8-GPU sharding of these blocks changes who proves which contiguous slice (proof_types.rs:23-24,
docs/usage_seq_diagrams.md:12-18), not a single proof byte: the aggregation tree is fixed by the block, so what is
pinned here on one GPU -- oracle digests of txn proofs, verifier acceptance of the block proof, the public-value
chain -- is what N ranks produce too (tests/test_host_cpu.py covers the N > 1 control flow over gloo)."""
import hashlib
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S1_LOG_N = (16, 9, 12, 14, 9, 12, 17)
S1_WIDTH = (128, 128, 192, 2432, 512, 320, 16)
DEFAULT_ORACLE_CFG = dict(table_log_lo=list(S1_LOG_N), table_log_hi=[x + 1 for x in S1_LOG_N], stark_rate_bits=1,
                          stark_cap_height=4, stark_num_queries=84, stark_pow_bits=16, arity_bits=4, final_poly_bits=5,
                          rec_log_n=13, rec_n_cols=135, rec_n_const=85, rec_rate_bits=3, rec_num_queries=28,
                          rec_pow_bits=16, shrink_depth=3, rec_air_id=8)


def words(b):
    return np.frombuffer(b, dtype=np.uint64)


@pytest.fixture(scope="module")
def pg(bpg):
    return bpg.proof_gen


@pytest.fixture(scope="module")
def default_state(pg):
    """ProverStateBuilder::default() with bench.py's 24 prover streams (126 GiB of device state)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    st = pg.ProverStateBuilder().set(n_workers=24, arena_bytes=5 << 30).build()
    yield st
    st.close()


def prove_block(pg, st, irs):
    """-> (block proof, {index: txn proof}) through prove_block_distributed, keeping the txn proofs it makes."""
    from proof_protocol_decoder_amd.block_driver import BlockDriver
    kept = {}

    def prove_txn(ir):
        p = pg.generate_txn_proof(st, ir)
        kept[ir.txn_number_before] = p
        return p

    drv = BlockDriver(st, n_threads=24, prove_txn=prove_txn)
    try:
        return drv.prove_block_distributed(irs), kept
    finally:
        drv.close()


def check_block(pg, st, oracle, blk, irs, kept, block_number):
    n = len(irs)
    pv, kind = pg.public_values_of(blk.intern)
    assert kind == 2 and blk.b_height == block_number
    assert (pv.txn_number_before, pv.txn_number_after) == (0, n)
    assert (pv.gas_used_before, pv.gas_used_after) == (0, n * 21000)
    assert pv.state_root_before == (1, 2, 3, 4)
    # every txn's public values chain into the next one's IR, and the block ends where the last txn ended
    for i in range(n - 1):
        assert kept[i].p_vals.state_root_after == irs[i + 1].state_root_before, i
    assert pv.state_root_after == kept[n - 1].p_vals.state_root_after
    pg.VerifierState.from_prover_state(st).verify(blk)                        # VerifierState::verify, product C++
    ost = oracle.PgState(**DEFAULT_ORACLE_CFG)
    assert ost.verify(words(blk.intern)) == 0                                 # the oracle's verifier
    bad = bytearray(blk.intern)
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(pg.ProofGenError):
        pg.VerifierState.from_prover_state(st).verify(pg.GeneratedBlockProof(blk.b_height, bytes(bad)))
    return ost


def test_block256_at_default_config(pg, default_state, oracle):
    """BASELINE configs[2]'s block on one GPU.  Byte parity with the oracle on transactions 0, 127 and 255 (digests
    by tools/gen_block256_golden.py), both verifiers accept the block proof, the public values chain 0 -> 256."""
    from proof_protocol_decoder_amd.block_driver import synthetic_block_irs
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "hotpath_golden.json")))["block256"]
    irs = synthetic_block_irs(gold["block_number"], 256, S1_LOG_N, S1_WIDTH)
    assert list(struct.unpack("<25Q", irs[255].to_bytes())) == gold["ir255"]   # the chain the oracle was given
    blk, kept = prove_block(pg, default_state, irs)
    assert sorted(kept) == list(range(256))
    for i in (0, 127, 255):
        assert words(kept[i].intern).size == gold["txn%d" % i]["n_words"]
        assert hashlib.sha256(kept[i].intern).hexdigest() == gold["txn%d" % i]["sha256"], \
            "txn %d differs from the oracle's proof" % i
    ost = check_block(pg, default_state, oracle, blk, irs, kept, gold["block_number"])
    assert ost.verify(words(kept[200].intern)) == 0


def test_txn_with_six_real_tables_at_baseline_sizes_matches_the_oracle(pg, default_state, oracle):
    """One transaction of the `--real-airs` workload at bp_config_default parameters and the S1 table heights: the
    arithmetic, byte-packing, Keccak-f, Keccak-sponge, logic and memory tables proven with their AIRs, both cross-table
    lookups active (keccak_sponge -> keccak_f over 2^14 / 2^9 rows, byte_packing -> memory over 2^9 / 2^17 rows), the
    recursion layer on the PLONK-shaped circuit.  The oracle proves the same transaction on the box's host cores (about
    half a minute); the containers are equal byte for byte."""
    import ctypes as C
    from proof_protocol_decoder_amd.block_driver import synthetic_block_irs
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:  # OpenMP would otherwise start one thread per host core, not per core this process may run on
        C.CDLL("libgomp.so.1").omp_set_num_threads(min(n, 64))
    except OSError:
        pass
    ir = synthetic_block_irs(4100, 2, S1_LOG_N, S1_WIDTH, keccak_air=True, logic_air=True, memory_air=True, arithmetic_air=True,
                             byte_packing_air=True, keccak_sponge_air=True)[1]
    assert ir.table_width == (309, 299, 192, 2431, 2414, 524, 45)
    got = pg.generate_txn_proof(default_state, ir)
    want = oracle.PgState(**DEFAULT_ORACLE_CFG).txn(list(struct.unpack("<25Q", ir.to_bytes())))
    assert words(got.intern).shape == want.shape and (words(got.intern) == want).all()
    tables = pg.generate_txn_table_proofs(default_state, ir)
    pg.verify_txn_table_proofs(default_state.cfg, tables)
    assert oracle.PgState(**DEFAULT_ORACLE_CFG).verify_tables(words(tables)) == 0


def test_block1024_properties(pg, default_state, oracle):
    """BASELINE configs[4]: 1024 transactions, 1023 aggregations (depth 10), one block proof.  No oracle run of this
    size exists (about six CPU hours); checked by properties: acceptance by both verifiers, rejection after a bit
    flip, the public-value chain 0 -> 1024, and determinism of the tree (proving transaction 1023 again gives the
    same bytes)."""
    from proof_protocol_decoder_amd.block_driver import synthetic_block_irs
    irs = synthetic_block_irs(3024, 1024, S1_LOG_N, S1_WIDTH)
    blk, kept = prove_block(pg, default_state, irs)
    assert sorted(kept) == list(range(1024))
    check_block(pg, default_state, oracle, blk, irs, kept, 3024)
    assert pg.generate_txn_proof(default_state, irs[1023]).intern == kept[1023].intern


def test_state_that_does_not_fit_is_refused_with_a_sizing_message(pg):
    """bp_state_build sizes the whole state against hipMemGetInfo before the first allocation."""
    with pytest.raises(pg.ProofGenError, match="does not fit device"):
        pg.ProverStateBuilder().set(n_workers=64, arena_bytes=8 << 30).build()
