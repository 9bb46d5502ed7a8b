"""Adds "block256" to tests/golden/hotpath_golden.json: BASELINE configs[2]'s 256-txn synthetic S1 block (block
number 2256) at bp_config_default parameters -- the ORACLE's proofs of transactions 0, 127 and 255 (sha256 of the
proof words) and the IR of txn 255 (pins the public-value chain the oracle was given).  CPU only, a few minutes
in the build container:

    python tools/gen_block256_golden.py

The GPU test (tests/test_gpu_proofgen.py::test_block256_at_default_config) proves the whole block on the device
and compares these three digests; the other 253 transactions are covered by verifier acceptance of the block proof
and the chaining of its public values.  Nothing here reads /root/reference.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_hotpath_golden as g  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (generator of fixtures: test infrastructure)

BLOCK, N_TXN, PICK = 2256, 256, (0, 127, 255)


def main():
    g.use_the_cores_we_may_run_on()
    orc.build()
    g.BLOCK = BLOCK
    irs = g.block_irs(N_TXN)
    st = orc.PgState(**g.DEFAULT_PG)
    out = {"generator": "tools/gen_block256_golden.py", "block_number": BLOCK, "n_txn": N_TXN,
           "ir255": [int(x) for x in irs[255]]}
    for i in PICK:
        t0 = time.time()
        out["txn%d" % i] = g.digest(st.txn(irs[i]))
        print("txn", i, "%.1f s" % (time.time() - t0), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "hotpath_golden.json")
    gold = json.load(open(path))
    gold["block256"] = out
    with open(path, "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
