"""What the host pays for waiting.  python tools/wait_mode_probe.py lib|torch [txns] [host_wait_knob]
  lib    the library decides the device's wait mode before anything else touches it (bp_host_wait_mode 1: the
         runtime's waits sleep)
  torch  the process uses the device through torch FIRST, as a host application with another HIP library would: the
         library must leave the mode alone (2) and its own poll-and-sleep wait takes over (prover.cpp, Worker::wait)
Prints the block rate and the CPU time the process burned per second of wall time (cores kept busy) over the timed
blocks; with N prover threads waiting most of the time, cores / N is the cost of one waiting thread."""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as pkg
from proof_protocol_decoder_amd import proof_gen as pg
from proof_protocol_decoder_amd.block_driver import BlockDriver, synthetic_block_irs

first = sys.argv[1] if len(sys.argv) > 1 else "lib"
txns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
knob = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = pkg.lib()
if first == "torch":
    x = torch.zeros(1 << 20, device="cuda")
    x += 1
    torch.cuda.synchronize()
else:
    L.bp_use_blocking_sync(0)
    torch.cuda.set_device(0)
L.bp_tune_host_wait(knob)
threads = 16
st = pg.ProverStateBuilder().set(device=0, n_workers=threads, arena_bytes=5 << 30).build()
drv = BlockDriver(st, n_threads=threads)
blocks = [synthetic_block_irs(b, txns, (16, 9, 12, 14, 9, 12, 17), (128, 128, 192, 2432, 512, 320, 16)) for b in range(4)]
drv.prove_block_distributed(blocks[0], 0, 1, None)
c0, t0 = os.times(), time.perf_counter()
for b in blocks[1:]:
    drv.prove_block_distributed(b, 0, 1, None)
dt = time.perf_counter() - t0
c1 = os.times()
cpu = (c1.user - c0.user) + (c1.system - c0.system)
print("first=%s knob=%d wait_mode=%d: %.2f txn-proofs/s; %.2f cores busy over %d prover threads (%.1f %% of a core per thread)"
      % (first, knob, L.bp_host_wait_mode(0), txns * 3 / dt, cpu / dt, threads, 100.0 * cpu / dt / threads), flush=True)
drv.close()
st.close()
