"""Regression probe for a hang found in round 3: a table proof through bp_stark_prove_air parks a worker (stream, event,
arena) on the device; if the device is switched to blocking host waits AFTER that (the first bp_state_build used to do
it), the next hipFree -- a device-wide wait, e.g. in bp_state_free -- never returns.  The library now fixes the wait
mode before it creates its first stream and never changes it (capi.cpp, Worker::init).  Steps (argv[1], any of):
  s / S  a small / large table proof first (parks a worker)     c / n  a child process using the GPU / RCCL
  r      bp_release_cached_memory before closing the state       G / F  knobs: per-round Poseidon / unfused Merkle tail
  t      the process uses the device through torch FIRST (kernels on the null stream): the library must then leave the
         wait mode alone (bp_host_wait_mode == 2) -- switching it under used queues hung the first hipFree
A watchdog thread dumps the Python stacks and ends the process after $WD seconds (a native hang cannot be interrupted).
tests/test_gpu_proofgen.py runs `s` in a fresh process."""
import faulthandler, os, subprocess, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
faulthandler.dump_traceback_later(int(os.environ.get("WD", "50")), exit=True)
import torch
import proof_protocol_decoder_amd as pkg
from proof_protocol_decoder_amd import proof_gen as pg
from pg_common import LOG_N, SMALL, WIDTH
steps = sys.argv[1]
L = pkg.lib()
def state():
    b = pg.ProverStateBuilder()
    for t, name in enumerate(pg.TABLES):
        getattr(b, "set_%s_circuit_size" % name)(range(SMALL["table_log_lo"][t], SMALL["table_log_hi"][t]))
    b.set(**{k: v for k, v in SMALL.items() if not k.startswith("table_")}, n_workers=2, arena_bytes=256 << 20)
    return b.build()
if "G" in steps: L.bp_tune_poseidon_grouped(0)
if "F" in steps: L.bp_tune_merkle_fused(0)
t0 = time.time()
def say(x): print("%.1f %s" % (time.time() - t0, x), flush=True)
if "t" in steps:
    x = torch.randint(0, 2**62, (16, 1 << 12), dtype=torch.int64, device="cuda")
    for _ in range(20):
        pkg.ops.merkle_commit(x, 11, 1, 4)
    torch.cuda.synchronize(); say("torch + L0 work on the null stream")
if "s" in steps:
    pkg.ops.stark_prove_synthetic(pkg.ops.stark_cfg(13, 135, n_const=82, deg_pow=3, rate_bits=3, num_queries=28), 1, 2); say("stark proof (parked worker)")
if "S" in steps:
    pkg.ops.stark_prove_synthetic(pkg.ops.stark_cfg(17, 16), 1, 2); say("big stark proof")
st = state(); say("state built")
say("state warnings: " + (st.warnings.replace("\n", " | ") or "none"))
p = pg.generate_txn_proof(st, pg.TxnProofGenIR(7, 0, 100, 121, (1, 2, 3, 4), 5, LOG_N, WIDTH)); say("txn proof")
if "c" in steps:
    r = subprocess.run([sys.executable, "-c", "import torch; x=torch.zeros(10,device='cuda'); torch.cuda.synchronize(); print('child ok')"], capture_output=True, text=True); say("child: " + r.stdout.strip())
if "n" in steps:
    code = "import os,torch,torch.distributed as d; os.environ.update(MASTER_ADDR='127.0.0.1',MASTER_PORT='29741',RANK='0',WORLD_SIZE='1'); torch.cuda.set_device(0); d.init_process_group('nccl', device_id=torch.device('cuda',0)); t=torch.ones(4,device='cuda'); d.all_reduce(t); torch.cuda.synchronize(); d.destroy_process_group(); print('nccl child ok')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")); say("child: " + r.stdout.strip()[-40:] + r.stderr.strip()[-200:])
if "r" in steps:
    L.bp_release_cached_memory(); say("released parked worker")
st.close(); say("state closed")
torch.cuda.synchronize()
torch.cuda.empty_cache()
say("host wait mode %d" % L.bp_host_wait_mode(0))
