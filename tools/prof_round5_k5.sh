#!/bin/bash
# Round 5: K5 (quotient_air_kernel<AIR>) alone per AIR -- timing table, then SQ / FETCH / WRITE counter passes (one
# rocprofv3 --pmc pass each, python itself after `--`).  Run on the GPU box: bash tools/prof_round5_k5.sh
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p "$O"
export GPU_MAX_HW_QUEUES=32
cd /tmp && export TMPDIR=/tmp
python "$R/tools/k5_air_probe.py" > "$O/r5_k5_air_probe.txt" 2> "$O/r5_k5_air_probe.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$O/r5_k5_sq" -- python "$R/tools/k5_air_probe.py" --counters > "$O/r5_k5_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/r5_k5_fetch" -- python "$R/tools/k5_air_probe.py" --counters > "$O/r5_k5_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/r5_k5_write" -- python "$R/tools/k5_air_probe.py" --counters > "$O/r5_k5_write.log" 2>&1
cd "$R" && K5_PREFIX=r5 python tools/prof_round4_summaries.py k5
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -size +20M -delete
