"""What rank 0 does after the gather on an 8-GPU node, measured on one GPU: 8 sub-block proofs made by ANOTHER prover
state (so every one is a foreign child: host verification, proof_gen.rs:66-75 done on the host) -> 7 aggregation
proofs in 3 levels -> the block proof.  This tail is serial to the step and bounds strong scaling at N = 8 (a 32-txn
shard is ~0.92 s).  Also: one aggregation proof alone on the chip, children known / foreign."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg
from proof_protocol_decoder_amd import proof_gen as pg
from proof_protocol_decoder_amd.block_driver import BlockDriver, synthetic_block_irs, tree_reduce

S1_LOG_N = (16, 9, 12, 14, 9, 12, 17)
S1_WIDTH = (128, 128, 192, 2432, 512, 320, 16)
torch.cuda.set_device(0)
n_sub = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a = pg.ProverStateBuilder().set(device=0, n_workers=8, arena_bytes=5 << 30).build()
da = BlockDriver(a, n_threads=8)
irs = synthetic_block_irs(7, 2 * n_sub, S1_LOG_N, S1_WIDTH)
subs = [da.prove_shard(irs[2 * k:2 * k + 2])[0] for k in range(n_sub)]
raws = [s.intern for s in subs]
# one aggregation, children known to the state
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    da.prove_agg(subs[0], subs[1])
    print("agg proof, children produced here: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
b = pg.ProverStateBuilder().set(device=0, n_workers=8, arena_bytes=5 << 30).build()
db = BlockDriver(b, n_threads=8)
fs = [db.decode_proof(r) for r in raws]
db.prove_agg(fs[2], fs[3])  # warm
for rep in range(3):
    t0 = time.perf_counter()
    db.prove_agg(fs[0], fs[1])
    print("agg proof, both children foreign: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for rep in range(4):
    t0 = time.perf_counter()
    fs = [db.decode_proof(r) for r in raws]
    top = tree_reduce(fs, db.prove_agg, db.pool)
    t1 = time.perf_counter()
    blk = db.prove_block(None, top)
    t2 = time.perf_counter()
    print("tail after the gather, %d foreign sub-block proofs: tree %.1f ms + block proof %.1f ms = %.1f ms"
          % (n_sub, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), flush=True)
pg.VerifierState.from_prover_state(b).verify(blk)
print("block proof verifies", flush=True)
da.close(); db.close(); a.close(); b.close()
