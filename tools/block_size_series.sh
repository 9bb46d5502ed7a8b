#!/bin/bash
# The block rate against the block size (what a rank of an 8-GPU run proves is 32 of 256 txns): python bench.py --txns N.
# Run on the GPU box: bash tools/block_size_series.sh [out file under gpurun_out/]
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r5_block_size_series.txt}
export GPU_MAX_HW_QUEUES=32
cd "$R"
: > "$OUT"
for n in 256 64 32 16; do
  steps=3; [ "$n" -lt 64 ] && steps=6
  python bench.py --txns $n --steps $steps --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%4d txns: %.2f txn-proofs/s (%.1f ms per block, %d steps)' % ($n, d['value'], d['ms_per_step'], d['steps']))" >> "$OUT"
done
cat "$OUT"
