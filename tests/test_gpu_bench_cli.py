"""`python3 bench.py --gpus 2 ...` exactly as the driver's scaling run starts it -- no launcher, no WORLD_SIZE:
bench.py must start its own ranks (fresh children, before anything touches the GPU), run the block across them and
print ONE JSON line from rank 0.  Two ranks share this box's one GPU (BPG_SHARE_GPU=1: both use device 0 and the
gather runs over gloo, since RCCL wants one device per rank); everything else is the code path of the 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BPG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--txns", "64", "--steps", "1",
                        "--warmup", "0", "--no-profile", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(lines[0])
    # rank 0 verified the block proof before printing (bench.py: VerifierState.verify)
    assert out["n_gpus"] == 2 and out["config"]["txns_per_block"] == 64 and out["value"] > 0
    assert out["metric"].endswith("64-txn synthetic block") and out["scaling"] == "strong"


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--txns", "4"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)
