#!/bin/bash
# Prover streams per GPU on the 256-txn block (round 4: the recursion chains are lock-step batches now)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p "$O"; cd "$R"
out=$O/r4_sweep_threads.txt; : > "$out"
for t in ${@:-14 16 18 20 22 24}; do
  python bench.py --threads $t --steps 3 --warmup 1 --no-profile --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threads=%d: %.2f txn-proofs/s %s' % (d['config']['prover_streams_per_gpu'], d['value'], d['config']['ms_of_each_step_rank0']))" >> "$out"
done
cat "$out"
