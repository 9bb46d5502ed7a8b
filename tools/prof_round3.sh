# Round 3 profile set (profiles/r3_*).  Run from the repo root on the GPU box:  bash tools/prof_round3.sh
# Every rocprofv3 command has `python` itself after `--`; counters are collected in their own passes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
HEAD_ID=$(cat $R/.head_for_profiles 2>/dev/null)
cd /tmp && export TMPDIR=/tmp
# 1. the roofline leg exactly as bench.py measures it (its child process): kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_leg -- python $R/bench.py --leg-only --leg-skip-extras > $O/r3_leg.log 2>&1 &&
# 2. HBM traffic of the leg's LDE family: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r3_pmc_f -- python $R/bench.py --leg-only --leg-skip-extras > $O/r3_pmc_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r3_pmc_w -- python $R/bench.py --leg-only --leg-skip-extras > $O/r3_pmc_w.log 2>&1 &&
cd $R && python tools/pmc_family_traffic.py gpurun_out/r3_pmc_f gpurun_out/r3_pmc_w gpurun_out/r3_pmc_f.log > gpurun_out/r3_pmc_lde_family.txt 2>&1
echo "PMC passes taken at HEAD $HEAD_ID (python bench.py --leg-only --leg-skip-extras under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" >> gpurun_out/r3_pmc_lde_family.txt; cat gpurun_out/r3_pmc_lde_family.txt
# 3. the LOADED run (64 txns on 24 streams): kernel trace + stats, then two SQ passes
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_64txn -- python $R/bench.py --txns 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/r3_64txn.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/r3_sq_loaded1 -- python $R/bench.py --txns 64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/r3_sq_loaded1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/r3_sq_loaded2 -- python $R/bench.py --txns 64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/r3_sq_loaded2.log 2>&1
# 4. the Poseidon kernel alone on a full chip: SQ counters of leaf hashing 2^21 rows x 8 permutations
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/r3_hash_sq1 -- python $R/tools/pmc_probe_hash.py > $O/r3_hash_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/r3_hash_sq2 -- python $R/tools/pmc_probe_hash.py > $O/r3_hash_sq2.log 2>&1
cd $R
for d in r3_leg r3_64txn; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp "$f" $O/${d}_kernel_stats.csv; done
python tools/prof_round3_summaries.py
# 5. the default bench line of the round (not under the profiler)
python bench.py > gpurun_out/r3_bench.log 2>&1; grep '^{' gpurun_out/r3_bench.log | tail -1 > gpurun_out/r3_bench.json
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
