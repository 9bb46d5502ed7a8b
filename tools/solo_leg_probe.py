"""Repeats bench.py's single-stream roofline leg several times in one process (is the figure stable?)."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as pkg
from proof_protocol_decoder_amd import proof_gen as pg
from proof_protocol_decoder_amd.block_driver import BlockDriver, synthetic_block_irs
S1_LOG_N = (16, 9, 12, 14, 9, 12, 17); S1_WIDTH = (128, 128, 192, 2432, 512, 320, 16)
L = pkg.lib()
if len(sys.argv) > 1 and sys.argv[1] == "spin":
    pass
else:
    L.bp_use_blocking_sync(0)
torch.cuda.set_device(0)
L.bp_profile_read.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
solo = pg.ProverStateBuilder().set(device=0, n_workers=1, arena_bytes=5 << 30).build()
drv = BlockDriver(solo, n_threads=1)
irs = synthetic_block_irs(1000, 2, S1_LOG_N, S1_WIDTH)
drv.prove_shard(irs[:1])
for rep in range(6):
    L.bp_profile_reset(); L.bp_profile_enable(1)
    t0 = time.time(); drv.prove_shard(irs); torch.cuda.synchronize(); dt = time.time() - t0
    L.bp_profile_enable(0)
    out = []
    for fam in (0, 1, 2):
        n, ms, by = C.c_uint64(), C.c_double(), C.c_double()
        L.bp_profile_read(fam, C.byref(n), C.byref(ms), C.byref(by))
        out.append("fam%d n=%d avg %.1f us rate %.1f" % (fam, n.value, ms.value * 1e3 / max(1, n.value), by.value / max(1e-9, ms.value) / 1e6))
    print("rep %d wall %.3f s | %s" % (rep, dt, " | ".join(out)), flush=True)
