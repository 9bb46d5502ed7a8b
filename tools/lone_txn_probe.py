"""A transaction alone on the chip: wall time of generate_txn_proof, nothing else running (what the last txn of a
shard, or a shard of one, costs).  python tools/lone_txn_probe.py [reps] [--real-airs] [--one-worker]
By default the state has four workers: the lone prover borrows the three idle ones' streams as side lanes (the
seven trace commitments overlap on them); --one-worker: a state of one prover stream, the single-stream reference.
Under `rocprofv3 --kernel-trace --stats -- python tools/lone_txn_probe.py 3` the kernel stats are the lone txn's own."""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_protocol_decoder_amd as pkg
from proof_protocol_decoder_amd import proof_gen as pg
from proof_protocol_decoder_amd.block_driver import synthetic_block_irs

S1_LOG_N = (16, 9, 12, 14, 9, 12, 17)
S1_WIDTH = (128, 128, 192, 2432, 512, 320, 16)
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
real = "--real-airs" in sys.argv
L = pkg.lib()
L.bp_use_blocking_sync(0)
one = "--one-worker" in sys.argv
for a in sys.argv:   # --witness-threads=N: bp_tune_witness_threads (default 7; 1 = the prover's own thread)
    if a.startswith("--witness-threads="):
        L.bp_tune_witness_threads(int(a.split("=")[1]))
st = pg.ProverStateBuilder().set(device=0, n_workers=1 if one else 4, arena_bytes=6 << 30).build()
irs = synthetic_block_irs(2000, reps + 1, S1_LOG_N, S1_WIDTH, keccak_air=real, logic_air=real, memory_air=real,
                          arithmetic_air=real, byte_packing_air=real, keccak_sponge_air=real)
pg.generate_txn_proof(st, irs[0])
ms = []
for ir in irs[1:]:
    t0 = time.perf_counter()
    pg.generate_txn_proof(st, ir)
    ms.append((time.perf_counter() - t0) * 1e3)
print("lone txn%s%s: %s ms; median %.1f ms" % (" (six real tables)" if real else "", " (one worker, no side lanes)" if one else "", [round(x, 1) for x in ms], sorted(ms)[len(ms) // 2]))
st.close()
