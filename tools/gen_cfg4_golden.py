"""Digest of the ORACLE's proof of one Keccak-wide table (2^log_n rows x 2432 columns, rate 2; BASELINE configs[3]
at log_n = 20).  Too large for the build container (64 GB): run it on the GPU box's host cores (270 GB of RAM),
which only executes the CPU oracle -- nothing here touches the GPU:

    gpurun --timeout 1200 -- 'python tools/gen_cfg4_golden.py 20 > gpurun_out/cfg4_golden_20.json'

then merge the printed object into tests/golden/hotpath_golden.json under "tables" (key logn20_C2432).
Same seeds as tests/test_gpu_stark.py::test_keccak_wide_table_2e20_x_2432."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as orc  # noqa: E402

log_n = int(sys.argv[1])
# second argument "keccak_f": the same configuration as a REAL Keccak-f[1600] trace (AIR 1, 2431 columns, witness
# drawn from the seed); merge under "tables" as logn<k>_keccak_f
AIR = 1 if len(sys.argv) > 2 and sys.argv[2] == "keccak_f" else 0
C = 2431 if AIR else 2432
try:  # OpenMP would otherwise start one thread per host core, not per core of this process's share
    import ctypes
    _n = len(os.sched_getaffinity(0))
    _q, _p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if _q != "max":
        _n = min(_n, max(1, int(int(_q) / int(_p))))
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(min(_n, 64))
except (OSError, ValueError):
    pass
orc.build()


def _heartbeat():  # the GPU box takes 7 silent minutes for a hang
    import threading
    t00 = time.time()

    def beat():
        while True:
            time.sleep(45)
            print("... oracle at work, %.0f s" % (time.time() - t00), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


_heartbeat()
t0 = time.time()
cfg = orc.make_cfg(log_n, C, air_id=AIR)
tr = orc.keccak_trace(log_n, seed=0x5EED000000000004) if AIR else orc.synth_trace(0x5EED000000000004, cfg, None)
t1 = time.time()
tc = orc.Committed.from_values(tr, 1, 4)
t2 = time.time()
ch = orc.PyChallenger()
ch.observe(tc.cap())
ctl = np.array([ch.challenge() for _ in range(4)], dtype=np.uint64)
proof = orc.stark_prove(cfg, tr, ctl, ch, None, tc)
t3 = time.time()
w = np.ascontiguousarray(proof, dtype="<u8")
print(json.dumps({"air_id": AIR, "shape": [log_n, C, 0, 1, 1], "seed": "0x5EED000000000004", "sha256": hashlib.sha256(w.tobytes()).hexdigest(),
                  "n_words": int(w.size), "head": [int(x) for x in w[:6]], "tail": [int(x) for x in w[-2:]],
                  "oracle_seconds": {"trace": round(t1 - t0, 1), "commit": round(t2 - t1, 1), "prove": round(t3 - t2, 1)},
                  "peak_rss_gib": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 2**20, 1)}))
