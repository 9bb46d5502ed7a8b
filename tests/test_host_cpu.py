"""CPU-only tests of the product's host side: the C ABI surface, the CPU verifier, the block
driver (sharding, trees, the N>1 gather over gloo).  No GPU compute is called here."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

from pg_common import LOG_N, SMALL, WIDTH, ir_words

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_every_declared_symbol():
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    header = open(os.path.join(ROOT, "include", "bpg.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, "declared in include/bpg.h but not exported: %s" % missing
    assert L.bp_version().startswith(b"bpg")


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from proof_protocol_decoder_amd import proof_gen as pg
    with pytest.raises(pg.ProofGenError) as e:
        pg.ProverStateBuilder().build()
    assert e.value.code == -4 and "no CPU fallback" in e.value.message


def test_product_never_touches_the_oracle_directory():
    """The product path must not import, link or execute anything under oracle/."""
    pkg_dir = os.path.join(ROOT, "proof_protocol_decoder_amd")
    pat = re.compile(r"import\s+oracle|from\s+oracle|oracle/|liboracle|pyoracle|orc_[a-z]")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".cuh", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(src), "%s references the oracle" % os.path.join(dirpath, f)


def test_ir_encoding_and_public_values_roundtrip():
    from proof_protocol_decoder_amd import proof_gen as pg
    ir = pg.TxnProofGenIR(7, 3, 100, 121, (1, 2, 3, 4), 0x5EED, LOG_N, WIDTH)
    assert list(struct.unpack("<25Q", ir.to_bytes())) == ir_words(7, 3, 0x5EED)
    with pytest.raises(pg.ProofGenError):
        pg.TxnProofGenIR(7, 3, 100, 121, (1, 2, 3, 0xFFFFFFFFFFFFFFFF), 1, LOG_N, WIDTH).to_bytes()


@pytest.fixture(scope="module")
def cpu_chain(oracle):
    st = oracle.PgState(**SMALL)
    t0 = st.txn(ir_words(7, 0, 0x5EED0001))
    root1 = tuple(int(x) for x in t0[4 + 84 + 8:4 + 84 + 12])
    t1 = st.txn(ir_words(7, 1, 0x5EED0002, root_before=root1, gas=(121, 150)))
    agg = st.agg(t0, False, t1, False)
    blk = st.block(None, agg)
    return st, t0, t1, agg, blk


def make_verifier(oracle_state):
    from proof_protocol_decoder_amd import proof_gen as pg
    b = pg.ProverStateBuilder()
    for t, name in enumerate(pg.TABLES):
        getattr(b, "set_%s_circuit_size" % name)(range(SMALL["table_log_lo"][t], SMALL["table_log_hi"][t]))
    b.set(**{k: v for k, v in SMALL.items() if not k.startswith("table_")})
    return pg.VerifierState.from_caps(b.cfg, oracle_state.circuit_caps().reshape(-1))


def test_product_cpu_verifier_accepts_oracle_proofs_and_rejects_corruption(cpu_chain):
    """VerifierState::verify (verifier_state.rs:56-71) is host code: it must agree with the oracle's."""
    from proof_protocol_decoder_amd import proof_gen as pg
    st, t0, t1, agg, blk = cpu_chain
    v = make_verifier(st)
    v.verify(blk.tobytes())
    v.verify_any(t0.tobytes())
    v.verify_any(agg.tobytes())
    with pytest.raises(pg.ProofGenError):
        v.verify(agg.tobytes())            # not a block proof
    rng = np.random.default_rng(11)
    for i in list(rng.integers(4, blk.size, size=16)) + [4 + 17, 4 + 30 + 16, blk.size - 1]:
        bad = blk.copy()
        bad[i] ^= np.uint64(1 << int(rng.integers(0, 60)))
        with pytest.raises(pg.ProofGenError) as e:
            v.verify(bad.tobytes())
        assert e.value.code in (-5, -2)
    pv, kind = pg.public_values_of(blk.tobytes())
    assert kind == 2 and pv.block_number == 7 and pv.txn_number_after == 2
    # a container whose list has another length than its kind's circuit hashes is refused before any hashing: here the
    # block proof's list with one more word (and the header saying so)
    longer = np.concatenate([blk[:4 + 30], np.zeros(1, dtype=np.uint64), blk[4 + 30:]])
    longer[2] = 31
    with pytest.raises(pg.ProofGenError, match="public inputs"):
        v.verify(longer.tobytes())


def test_host_transcript_permutation_matches_the_oracle_in_both_forms(oracle):
    """poseidon_host (prover.cpp: the challenger's CPU permutation, MDS over 32-bit halves, AVX2 where the CPU has it)
    against the oracle's permutation on random, edge and non-canonical states, in the automatic and the scalar form."""
    import ctypes as C
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    L.bp_debug_poseidon_host.argtypes = [C.c_void_p, C.c_size_t]
    L.bp_tune_host_poseidon.argtypes = [C.c_int]
    P = 0xFFFFFFFF00000001
    rng = np.random.default_rng(5)
    states = rng.integers(0, 2**64, size=(257, 12), dtype=np.uint64)
    edge = np.array([0, 1, P - 1, P, P + 1, 2**32 - 1, 2**32, 2**64 - 1, 2**63, 0xFFFFFFFF, 0xFFFFFFFF00000000, 7], dtype=np.uint64)
    states[0] = 0
    states[1] = edge
    states[2] = np.uint64(2**64 - 1)
    states[3] = np.uint64(P - 1)
    want = oracle.poseidon(states % np.uint64(P))
    try:
        for mode in (0, 1):
            L.bp_tune_host_poseidon(mode)
            got = states.copy()
            assert L.bp_debug_poseidon_host(got.ctypes.data, got.shape[0]) == 0
            assert (got == want).all(), "host permutation differs from the oracle (form %d)" % mode
    finally:
        L.bp_tune_host_poseidon(0)
    assert [hex(int(x)) for x in want[0][:2]] == ["0x3c18a9786cb0b359", "0xc4055e3364a246c3"]   # SURVEY.md appendix A, recalled KAT


def test_shard_bounds_are_contiguous_and_balanced():
    from proof_protocol_decoder_amd.block_driver import shard_bounds
    for n, w in [(256, 8), (256, 1), (10, 4), (7, 7), (1024, 8), (5, 2)]:
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_tree_reduce_keeps_order_and_counts():
    from concurrent.futures import ThreadPoolExecutor
    from proof_protocol_decoder_amd.block_driver import aggregation_plan, tree_reduce
    calls = []

    def agg(a, b):
        assert a[1] == b[0]                 # contiguous ranges only
        calls.append((a, b))
        return (a[0], b[1])
    for shape in ("balanced", "pairs_then_chain"):
        for n in (1, 2, 3, 5, 8, 13, 32):
            calls.clear()
            with ThreadPoolExecutor(4) as pool:
                assert tree_reduce([(i, i + 1) for i in range(n)], agg, pool, shape) == (0, n)
            assert len(calls) == n - 1
            plan = aggregation_plan(n, shape)
            assert len(plan) == n - 1 and all(l < n + k and r < n + k for k, (l, r) in enumerate(plan))
    # what is left to do once the LAST leaf exists: log2(n) aggregations one after the other in the balanced tree,
    # its pair and one chain step in the other shape -- whatever n is
    def tail(n, shape):
        plan, depth = aggregation_plan(n, shape), {n - 1: 0}
        for k, (l, r) in enumerate(plan):
            if l in depth or r in depth:
                depth[n + k] = 1 + max(depth.get(l, 0), depth.get(r, 0))
        return depth[n + len(plan) - 1]
    assert [tail(n, "balanced") for n in (16, 32, 256)] == [4, 5, 8]
    assert [tail(n, "pairs_then_chain") for n in (16, 32, 256)] == [2, 2, 2]


def test_prove_shard_runs_aggregations_ahead_of_waiting_transactions():
    """The shard's tree is scheduled with the proving: an aggregation whose children exist is taken before any
    transaction that has not started, and a failure in any task surfaces."""
    import threading
    import time
    from proof_protocol_decoder_amd.block_driver import BlockDriver
    order, lock = [], threading.Lock()

    def txn(i):
        time.sleep(0.01)
        with lock:
            order.append(("txn", i))
        return (i, i + 1)

    def agg(a, b):
        assert a[1] == b[0]
        with lock:
            order.append(("agg", a[0], b[1]))
        return (a[0], b[1])
    drv = BlockDriver(None, n_threads=2, prove_txn=txn, prove_agg=agg)
    try:
        top, leaves = drv.prove_shard(list(range(12)))
        assert top == (0, 12) and leaves == [(i, i + 1) for i in range(12)]
        # the pair (0, 1) is aggregated long before the last transactions are proven
        assert order.index(("agg", 0, 2)) < order.index(("txn", 6))
        assert sum(1 for o in order if o[0] == "agg") == 11

        def bad(i):
            if i == 5:
                raise RuntimeError("txn 5 failed")
            return txn(i)
        drv.prove_txn = bad
        with pytest.raises(RuntimeError, match="txn 5 failed"):
            drv.prove_shard(list(range(12)))
    finally:
        drv.close()


DRIVER_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import ctypes as C
import numpy as np, torch, torch.distributed as dist
from pg_common import SMALL, ir_words
from oracle import pyoracle
from proof_protocol_decoder_amd import proof_gen as pg
from proof_protocol_decoder_amd.block_driver import BlockDriver, TorchGather
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
st = pyoracle.PgState(**SMALL)          # the checker stands in for the GPU prover in this CPU test
def wrap(words):
    raw = words.tobytes(); pv, k = pg.public_values_of(raw)
    return (pg.GeneratedAggProof if k == 1 else pg.GeneratedTxnProof)(pv, raw)
def prove_txn(ir): return wrap(st.txn(ir))
def prove_agg(a, b):
    a, b = pg._as_aggregatable(a), pg._as_aggregatable(b)
    return wrap(st.agg(np.frombuffer(a.intern(), dtype=np.uint64), a.is_agg(),
                       np.frombuffer(b.intern(), dtype=np.uint64), b.is_agg()))
def prove_block(parent, agg):
    raw = st.block(None, np.frombuffer(agg.intern, dtype=np.uint64)).tobytes()
    return pg.GeneratedBlockProof(pg.public_values_of(raw)[0].block_number, raw)
# a chained 5-txn block: every rank derives the same inputs
L = pg._bind()
L.bp_state_root_after.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
irs, root, gas = [], (1, 2, 3, 4), 0
N_TXNS, FAIL_RANK, TOP_TREE = {n_txns}, {fail_rank}, {top_tree!r}
for i in range(N_TXNS):
    seed = 0x5EED1000 + i
    irs.append(ir_words(11, i, seed, root_before=root, gas=(gas, gas + 7)))
    out = (C.c_uint64 * 4)()
    assert L.bp_state_root_after((C.c_uint64 * 4)(*root), seed, i, out) == 0
    root, gas = tuple(out), gas + 7
FAIL_AGG = {fail_agg}
if rank == FAIL_RANK:
    def prove_txn(ir): raise RuntimeError("injected failure on rank %d" % rank)
if rank == FAIL_AGG:
    def prove_agg(a, b): raise RuntimeError("injected aggregation failure on rank %d" % rank)
drv = BlockDriver(n_threads=2, prove_txn=prove_txn, prove_agg=prove_agg, prove_block=prove_block)
if FAIL_AGG >= 0:
    # a failure INSIDE the pairwise tree (one transaction per rank: every aggregation is a tree step): the failing rank
    # raises its own error, the ranks above it on the way to rank 0 raise ShardFailed, the ranks that had already handed
    # their proof over return None -- and nobody is left waiting in a recv
    from proof_protocol_decoder_amd.block_driver import ShardFailed
    try:
        got = drv.prove_block_distributed(irs, rank, world, TorchGather(torch.device("cpu")), top_tree="pairwise")
    except ShardFailed as e:
        assert rank == 0 and rank != FAIL_AGG, (rank, str(e))
    except RuntimeError as e:
        assert rank == FAIL_AGG and "injected aggregation" in str(e)
    else:
        assert got is None and rank not in (0, FAIL_AGG)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
if FAIL_RANK >= 0:
    from proof_protocol_decoder_amd.block_driver import ShardFailed
    try:
        drv.prove_block_distributed(irs, rank, world, TorchGather(torch.device("cpu")), top_tree=TOP_TREE)
    except ShardFailed as e:
        assert rank != FAIL_RANK and "failed on rank" in str(e)
    except RuntimeError as e:
        assert rank == FAIL_RANK and "injected" in str(e)
    else:
        raise SystemExit("rank %d: no exception although rank %d failed" % (rank, FAIL_RANK))
    dist.barrier()                     # every rank got here: nobody is stuck in the gather
    dist.destroy_process_group()
    sys.exit(0)
blk = drv.prove_block_distributed(irs, rank, world, TorchGather(torch.device("cpu")), top_tree=TOP_TREE)
if rank == 0:
    w = np.frombuffer(blk.intern, dtype=np.uint64)
    assert st.verify(w) == 0
    pv, kind = pg.public_values_of(blk.intern)
    assert kind == 2 and (pv.txn_number_before, pv.txn_number_after) == (0, N_TXNS) and pv.state_root_after == root
    np.save({out!r}, w)
else:
    assert blk is None
dist.destroy_process_group()
'''


def test_block_driver_world_size_2_over_gloo(tmp_path, oracle):
    """N>1 path: contiguous slices, local trees, gather to rank 0, top tree + block proof; rank 0's
    block proof verifies and carries the same public values as the single-process run."""
    out2, out1 = str(tmp_path / "w2.npy"), str(tmp_path / "w1.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    script = tmp_path / "drv.py"
    for world, out, port in ((2, out2, 29611), (1, out1, 29612)):
        script.write_text(DRIVER_SCRIPT.format(root=ROOT, out=out, n_txns=5, fail_rank=-1, top_tree="pairwise", fail_agg=-1))
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a, b = np.load(out2), np.load(out1)
    # the tree shape differs (2 slices vs 1) so the proofs differ, but the public values must not
    assert a.shape == b.shape and (a[4 + 17:4 + 30] == b[4 + 17:4 + 30]).all()


def _run_driver(tmp_path, world, n_txns, port, fail_rank=-1, omp="1", top_tree="pairwise", fail_agg=-1):
    out = str(tmp_path / ("w%d_%d_%s.npy" % (world, n_txns, top_tree)))
    script = tmp_path / ("drv_%d_%d_%d_%s_%d.py" % (world, n_txns, fail_rank, top_tree, fail_agg))
    script.write_text(DRIVER_SCRIPT.format(root=ROOT, out=out, n_txns=n_txns, fail_rank=fail_rank, top_tree=top_tree, fail_agg=fail_agg))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=omp)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return out


def test_block_driver_world_size_8_uneven_split(tmp_path, oracle):
    """The 8-rank shape of the driver's scaling run, rehearsed over gloo: 21 transactions over 8 ranks (slices of
    3,3,3,3,3,2,2,2), local trees of different depth, gather, top tree of 8 sub-proofs, block proof; public values
    equal the single-process run's."""
    a = np.load(_run_driver(tmp_path, 8, 21, 29621))
    b = np.load(_run_driver(tmp_path, 1, 21, 29622, omp="8"))
    assert a.shape == b.shape and (a[4 + 17:4 + 30] == b[4 + 17:4 + 30]).all()
    # the pairwise top tree (rank 2k+1 -> 2k, 4k+2 -> 4k, 4 -> 0: rank 0 makes three aggregations, not seven) is the
    # balanced tree over ranks the gather form builds on rank 0: the same block proof, byte for byte
    c = np.load(_run_driver(tmp_path, 8, 21, 29626, top_tree="gather"))
    assert (a == c).all()


def test_block_driver_fewer_entries_than_ranks(tmp_path, oracle):
    """A block of 0 or 1 transactions is padded to exactly two entries (decoding.rs:304-347): on more than two
    ranks the rest have an empty slice.  They must join the gather with an empty payload, not raise and leave
    ranks 0 and 1 waiting in the collective."""
    a = np.load(_run_driver(tmp_path, 4, 2, 29623))
    b = np.load(_run_driver(tmp_path, 1, 2, 29624, omp="4"))
    assert (a == b).all()          # two ranks with one txn each = the same tree as one rank with two
    c = np.load(_run_driver(tmp_path, 4, 2, 29627, top_tree="gather"))
    assert (a == c).all()
    # five ranks, three entries: ranks 3 and 4 are empty, rank 4's nothing travels 4 -> 0 at the last level
    d = np.load(_run_driver(tmp_path, 5, 3, 29628))
    e = np.load(_run_driver(tmp_path, 5, 3, 29629, top_tree="gather"))
    assert (d == e).all()


def test_block_driver_failure_on_one_rank_raises_everywhere(tmp_path, oracle):
    """A rank whose shard raises reports it through the gather's length exchange; every rank raises (ShardFailed on
    the healthy ones, the original error on the failed one) and all of them reach the next barrier."""
    _run_driver(tmp_path, 3, 6, 29625, fail_rank=1)
    _run_driver(tmp_path, 3, 6, 29630, fail_rank=1, top_tree="gather")
    _run_driver(tmp_path, 5, 10, 29631, fail_rank=4)      # the rank that only joins the tree at its last level
    _run_driver(tmp_path, 4, 8, 29632, fail_rank=0)
    _run_driver(tmp_path, 4, 4, 29633, fail_agg=2)        # rank 2 fails merging rank 3's proof: 0 hears of it, 1 and 3 are done
    _run_driver(tmp_path, 4, 4, 29634, fail_agg=0)


def test_config_bounds_are_checked_before_any_device_work():
    """check_cfg (csrc/prover.cpp): a final polynomial longer than 256 points would be interpolated on the host in
    O(len^2), and 2 << pow_bits must not overflow: both are refused as invalid input, without a GPU."""
    from proof_protocol_decoder_amd import proof_gen as pg
    for kw in (dict(final_poly_bits=9), dict(final_poly_bits=31), dict(rec_pow_bits=33), dict(rec_n_const=5000),
               dict(arity_bits=3), dict(rec_num_queries=0)):
        with pytest.raises(pg.ProofGenError) as e:
            pg.ProverStateBuilder().set(**kw).build()
        assert e.value.code in (-2, -6), (kw, e.value.code)


def test_bench_spawns_its_own_ranks_when_started_without_a_launcher(monkeypatch):
    """`python3 bench.py --gpus 8` as the driver's scaling run starts it (no WORLD_SIZE): the parent must start
    torch.distributed.run with the same arguments BEFORE importing torch or touching the library, and exit with the
    launcher's code."""
    import importlib
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        seen["torch_loaded_by_bench"] = "torch" in bench.__dict__
        return subprocess.CompletedProcess(cmd, 7)

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "127.0.0.1" in cmd
    assert cmd[-7:] == [os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert not seen["torch_loaded_by_bench"]
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--txns", "0"])
    with pytest.raises(SystemExit, match="--txns must be >= 2"):
        bench.main()


def test_witness_entry_points_check_their_arguments_before_any_device_work():
    """bp_keccak_trace / bp_logic_trace / bp_memory_trace / bp_arithmetic_trace / bp_byte_packing_trace and the IR flag setters: null outputs,
    heights out of range and tables of the wrong width are refused with BP_ERR_INVALID_INPUT (-2) and a message, without a
    GPU."""
    import ctypes as C
    import proof_protocol_decoder_amd as pkg
    from proof_protocol_decoder_amd import proof_gen as pg
    L = pkg.lib()
    L.bp_last_error.restype = C.c_char_p
    for name in ("bp_keccak_trace", "bp_logic_trace", "bp_memory_trace", "bp_arithmetic_trace", "bp_byte_packing_trace",
                 "bp_keccak_sponge_trace", "bp_arithmetic_mul_trace"):
        f = getattr(L, name)
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        assert f(None, 1, 8, None, None) == -2 and name.encode() in L.bp_last_error()
        assert f(None, 1, 2, C.c_void_p(8), None) == -2 and b"log_n" in L.bp_last_error()
        assert f(None, 1, 40, C.c_void_p(8), None) == -2
    pg._bind()
    ir = (C.c_uint64 * 25)(*struct_ir())
    for setter, width in (("bp_ir_set_keccak_air", b"2431"), ("bp_ir_set_logic_air", b"524"), ("bp_ir_set_memory_air", b"45"),
                          ("bp_ir_set_arithmetic_air", b"309"), ("bp_ir_set_byte_packing_air", b"299"),
                          ("bp_ir_set_keccak_sponge_air", b"2414")):
        f = getattr(L, setter)
        f.argtypes = [C.POINTER(C.c_uint64), C.c_int]
        assert f(ir, 1) == -2 and width in L.bp_last_error()       # the synthetic widths do not fit the AIR
        assert f(ir, 0) == 0 and ir[1] == 1                        # switching it off is always possible
        junk = (C.c_uint64 * 25)()
        assert f(junk, 1) == -2 and b"not an IR" in L.bp_last_error()
    # all four flags together, each on a table of the right width
    w = list(struct_ir())
    w[18 + 0], w[18 + 1], w[18 + 3], w[18 + 4], w[18 + 5], w[18 + 6] = 309, 299, 2431, 2414, 524, 45
    ir = (C.c_uint64 * 25)(*w)
    for setter in ("bp_ir_set_arithmetic_air", "bp_ir_set_byte_packing_air", "bp_ir_set_keccak_air", "bp_ir_set_keccak_sponge_air",
                   "bp_ir_set_logic_air", "bp_ir_set_memory_air"):
        assert getattr(L, setter)(ir, 1) == 0
    assert ir[1] == 0x3F01
    assert L.bp_ir_set_logic_air(ir, 0) == 0 and ir[1] == 0x3D01
    # the arithmetic table is proven by ONE AIR: the multiplication AIR (flag 0x4000, 1217 columns) instead of AIR 4
    f = L.bp_ir_set_arithmetic_mul_air
    f.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    assert f(ir, 1) == -2 and b"1217" in L.bp_last_error()
    w[18 + 0] = 1217
    ir2 = (C.c_uint64 * 25)(*w)
    assert f(ir2, 1) == 0 and ir2[1] == 0x4001 and f(ir2, 0) == 0 and ir2[1] == 1
    ir2[1] |= 0x800
    assert f(ir2, 1) == -2 and b"ONE AIR" in L.bp_last_error()


def struct_ir():
    from pg_common import IR_MAGIC, LOG_N, WIDTH
    return [IR_MAGIC, 1, 7, 0, 100, 121, 1, 2, 3, 4, 0x5EED, *LOG_N, *WIDTH]
