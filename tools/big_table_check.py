"""BASELINE configs[3]: one Keccak-wide table (2^20 rows x 2432 columns, rate 2) proved on one GPU
and checked by the oracle's verifier.  ~100 GB of device memory; run on the GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import proof_protocol_decoder_amd as bpg
from oracle import pyoracle as O
log_n, C = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 2432
t0 = time.time()
proof = bpg.ops.stark_prove_synthetic(bpg.ops.stark_cfg(log_n, C), 0x5EED000000000004)
dt = time.time() - t0
t0 = time.time()
proof = bpg.ops.stark_prove_synthetic(bpg.ops.stark_cfg(log_n, C), 0x5EED000000000004)   # tables warm, arena parked by the first call
dt2 = time.time() - t0
cfg = O.make_cfg(log_n, C)
ch = O.PyChallenger(); ch.observe(proof[16:80])
ctl = np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
rc = O.stark_verify(cfg, proof, ctl, ch, None)
n = 1 << log_n
print("log_n=%d C=%d: prove %.2f s first call / %.2f s second call (both incl. witness generation; the first also allocates the ~100 GB arena), proof %.1f MB, "
      "oracle verifier rc=%d, trace %.1f GB" % (log_n, C, dt, dt2, proof.nbytes / 1e6, rc, n * C * 8 / 1e9), flush=True)
assert rc == 0
