"""Digest of the ORACLE's block proof of BASELINE configs[1]: the 16-txn synthetic S1 block (block number 2000) at
bp_config_default parameters -- all 16 txn proofs, the aggregation tree in the shape of block_driver.tree_reduce
(adjacent pairs per level), the block proof.  ~6 minutes on the GPU box's 16 host cores (CPU oracle only, nothing
touches the GPU); merge the printed object into tests/golden/hotpath_golden.json under "block16_full".

    gpurun --timeout 1200 -- 'python tools/gen_block16_golden.py > gpurun_out/block16_full.json'
"""
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

from gen_hotpath_golden import DEFAULT_PG, block_irs  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

t00 = time.time()


def beat():
    while True:
        time.sleep(45)
        print("... oracle at work, %.0f s" % (time.time() - t00), file=sys.stderr, flush=True)


threading.Thread(target=beat, daemon=True).start()


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


# OpenMP would otherwise start one thread per host core (256 on the GPU box) on a 16-core share: the first run
# of this script spent its whole 20-minute limit that way
import ctypes  # noqa: E402
CORES = usable_cores()
os.environ["OMP_NUM_THREADS"] = str(CORES)
try:
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(CORES)
except OSError:
    pass
N_TXN = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
orc.build()
sha = lambda w: hashlib.sha256(np.ascontiguousarray(w, dtype="<u8").tobytes()).hexdigest()
st = orc.PgState(**DEFAULT_PG)
irs = block_irs(16)[:N_TXN]
level = []
for i, ir in enumerate(irs):
    t0 = time.time()
    level.append((st.txn(ir), False))
    print("txn %d: %.1f s (%d threads)" % (i, time.time() - t0, CORES), file=sys.stderr, flush=True)
out = {"txn_sha256": [sha(p) for p, _ in level], "agg_sha256": []}
while len(level) > 1:
    nxt = []
    for k in range(0, len(level) - 1, 2):
        a = st.agg(level[k][0], level[k][1], level[k + 1][0], level[k + 1][1])
        out["agg_sha256"].append(sha(a))
        nxt.append((a, True))
    if len(level) % 2:
        nxt.append(level[-1])
    level = nxt
blk = st.block(None, level[0][0])
assert st.verify(blk) == 0
out["block_sha256"] = sha(blk)
out["block_words"] = int(blk.size)
out["oracle_seconds"] = round(time.time() - t00, 1)
out["n_txn"] = N_TXN
print(json.dumps(out))
