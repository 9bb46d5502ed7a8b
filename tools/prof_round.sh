#!/bin/bash
# Profile set of a round (profiles/<round>_*).  Run on the GPU box:  PROF_ROUND=r5 bash tools/prof_round.sh [k5|all]
# Every rocprofv3 command has `python` itself after `--`; counters are collected in their own passes (never with a
# trace domain other than --kernel-trace).
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
test -f "$R/bench.py" || { echo "no bench.py under $R" >&2; exit 1; }
O=$R/gpurun_out; mkdir -p "$O"
WHAT=${1:-all}
P=${PROF_ROUND:-r5}; export PROF_ROUND=$P
export GPU_MAX_HW_QUEUES=32   # before any rocprofv3 line: the profiler initialises the runtime before python can set it
HEAD_ID=$(cat "$R/.head_for_profiles" 2>/dev/null || echo "?")
cd /tmp && export TMPDIR=/tmp
# K5 alone per AIR: SQ pass, FETCH pass, WRITE pass over one launch per AIR; then the timing table
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$O/${P}_k5_sq" -- python "$R/tools/k5_air_probe.py" --counters > "$O/${P}_k5_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/${P}_k5_fetch" -- python "$R/tools/k5_air_probe.py" --counters > "$O/${P}_k5_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/${P}_k5_write" -- python "$R/tools/k5_air_probe.py" --counters > "$O/${P}_k5_write.log" 2>&1
python "$R/tools/k5_air_probe.py" > "$O/${P}_k5_air_probe.txt" 2> "$O/${P}_k5_air_probe.err"
if [ "$WHAT" = "all" ]; then
  # 1. the roofline leg exactly as bench.py measures it (its child process): kernel trace + stats
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${P}_leg" -- python "$R/bench.py" --leg-only --leg-skip-extras > "$O/${P}_leg.log" 2>&1
  # 2. HBM traffic of the leg's LDE family: FETCH_SIZE and WRITE_SIZE in separate passes
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/${P}_pmc_f" -- python "$R/bench.py" --leg-only --leg-skip-extras > "$O/${P}_pmc_f.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/${P}_pmc_w" -- python "$R/bench.py" --leg-only --leg-skip-extras > "$O/${P}_pmc_w.log" 2>&1
  (cd "$R" && python tools/pmc_family_traffic.py gpurun_out/${P}_pmc_f gpurun_out/${P}_pmc_w gpurun_out/${P}_pmc_f.log > gpurun_out/${P}_pmc_lde_family.txt 2>&1;
   echo "PMC passes taken at HEAD $HEAD_ID (python bench.py --leg-only --leg-skip-extras under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" >> gpurun_out/${P}_pmc_lde_family.txt; cat gpurun_out/${P}_pmc_lde_family.txt)
  # 3. the LOADED run (64 txns on 16 streams): kernel trace + stats, then two SQ passes
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${P}_64txn" -- python "$R/bench.py" --txns 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > "$O/${P}_64txn.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/${P}_sq_loaded1" -- python "$R/bench.py" --txns 64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile > "$O/${P}_sq_loaded1.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$O/${P}_sq_loaded2" -- python "$R/bench.py" --txns 64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile > "$O/${P}_sq_loaded2.log" 2>&1
  # 4. the Poseidon kernel alone on a full chip: SQ counters of leaf hashing 2^21 rows x 8 permutations
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$O/${P}_hash_sq1" -- python "$R/tools/pmc_probe_hash.py" > "$O/${P}_hash_sq1.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/${P}_hash_sq2" -- python "$R/tools/pmc_probe_hash.py" > "$O/${P}_hash_sq2.log" 2>&1
  for d in ${P}_leg ${P}_64txn; do f=$(find "$O/$d" -name "*kernel_stats.csv" | head -1); cp "$f" "$O/${d}_kernel_stats.csv"; done
fi
cd "$R" && python tools/prof_round4_summaries.py
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -size +20M -delete
