"""Host time of VerifierState::verify on proofs of the default shape (what rank 0 of a multi-GPU run pays per foreign
child at every level of the top tree): python tools/verify_time_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
from bench import S1_LOG_N, S1_WIDTH  # noqa: E402
from proof_protocol_decoder_amd import proof_gen as pg  # noqa: E402
from proof_protocol_decoder_amd.block_driver import synthetic_block_irs  # noqa: E402

b = pg.ProverStateBuilder()
for t, name in enumerate(pg.TABLES):
    getattr(b, "set_%s_circuit_size" % name)(range(S1_LOG_N[t], S1_LOG_N[t] + 1))
b.set(n_workers=2, arena_bytes=5 << 30)
st = b.build()
irs = synthetic_block_irs(4000, 2, S1_LOG_N, S1_WIDTH)
t0, t1 = (pg.generate_txn_proof(st, ir) for ir in irs)
agg = pg.generate_agg_proof(st, t0, t1)
v = pg.VerifierState.from_prover_state(st)
for name, p in (("txn proof", t0), ("aggregation proof", agg)):
    ms = []
    for _ in range(8):
        a = time.perf_counter()
        v.verify_any(p.intern)
        ms.append((time.perf_counter() - a) * 1e3)
    print("verify %s (%d bytes): %s ms; median %.2f ms" % (name, len(p.intern), [round(x, 2) for x in ms], sorted(ms)[4]))
