#!/bin/bash
# Block-size series + loaded kernel trace (round 4).  Run from anywhere on the GPU box: bash tools/r4_series.sh
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
test -f "$R/bench.py" || { echo "no bench.py under $R" >&2; exit 1; }
O=$R/gpurun_out; mkdir -p "$O"
cd "$R"
out=$O/r4_block_size_series.txt; : > "$out"
for n in 2 16 32 64; do
  python bench.py --txns $n --steps 4 --warmup 2 --no-profile --no-cpu-baseline > $O/r4_series_$n.json 2> $O/r4_series_$n.err
  python - "$O/r4_series_$n.json" $n >> "$out" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("txns=%s streams=%d: %.2f txn-proofs/s, %.1f ms per block %s" % (sys.argv[2], d["config"]["prover_streams_per_gpu"], d["value"], d["ms_per_step"], d["config"]["ms_of_each_step_rank0"]))
PY
done
cat "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_64txn -- python $R/bench.py --txns 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/r4_64txn.log 2>&1
f=$(find $O/r4_64txn -name "*kernel_stats.csv" | head -1); cp "$f" $O/r4_64txn_kernel_stats.csv
find $O/r4_64txn -name "*_kernel_trace.csv" -delete
python - "$O/r4_64txn_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = sum(int(r["Calls"]) for r in rows)
print("kernel launches in the traced run (2 x 64 txns + state build):", calls, "->", calls / 128.0, "per txn (upper bound)")
PY
