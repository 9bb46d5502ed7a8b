"""Aggregate Poseidon rate of many concurrent mid-sized Merkle commits (one per stream), the shape that
dominates a txn proof: 2^16 rows x 135 columns.  Is stream-level concurrency filling the chip?"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg
bpg.lib().bp_use_blocking_sync(0)
# the automatic kernel-form choice follows the number of L1 provers at work; this probe drives L0 directly,
# so it states the loaded-chip choice itself (one lane per state above 2^13 rows)
bpg.lib().bp_tune_quad_threshold(1 << 13)
torch.cuda.set_device(0)
log_n, r, C = 13, 3, 135
rows = 1 << (log_n + r)
perms = rows * ((C + 7) // 8) + rows
grouped = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bpg.lib().bp_tune_poseidon_grouped(grouped)
print("bp_tune_poseidon_grouped(%d)" % grouped, flush=True)
for n_streams in (1, 2, 4, 8, 16, 24):
    mats = [torch.randint(0, 2**62, (C, rows), dtype=torch.int64, device="cuda") for _ in range(n_streams)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    reps = 20
    def work(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                bpg.ops.merkle_commit(mats[i], log_n, r, 4)
        streams[i].synchronize()
    for i in range(n_streams): work(i)   # warm
    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("%2d streams: %.3f Gperm/s aggregate" % (n_streams, n_streams * reps * perms / dt / 1e9), flush=True)
