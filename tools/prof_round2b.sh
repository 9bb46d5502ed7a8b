# Round 2, second half (matrix-core Poseidon): the profile set of profiles/r2b_*.  Run from the repo root on the GPU box.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2b_4txn -- python $R/bench.py --txns 4 --threads 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/r2b_4txn.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r2b_pmc_f -- python $R/bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2b_pmc_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r2b_pmc_w -- python $R/bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2b_pmc_w.log 2>&1 &&
cd $R && python tools/pmc_family_traffic.py gpurun_out/r2b_pmc_f gpurun_out/r2b_pmc_w gpurun_out/r2b_pmc_f.log > gpurun_out/r2b_pmc_lde_family.txt 2>&1; echo "PMC passes taken at HEAD $(cat $R/.head_for_profiles 2>/dev/null)" >> gpurun_out/r2b_pmc_lde_family.txt; cat gpurun_out/r2b_pmc_lde_family.txt
rocprofv3 -L > $O/r2b_counters_available.txt 2>&1; grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $O/r2b_counters_available.txt | sort -u | tr "\n" " "; echo
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/r2b_sq -- python $R/bench.py --txns 4 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --quad-threshold-log2 13 > $O/r2b_sq.log 2>&1; tail -2 $O/r2b_sq.log
MF=$(grep -o "SQ_INSTS_VALU_MFMA_I8\|SQ_INSTS_MFMA\|SQ_VALU_MFMA_BUSY_CYCLES" $O/r2b_counters_available.txt | sort -u | tr "\n" " ")
if [ -n "$MF" ]; then cd /tmp && rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $O/r2b_sq -- python $R/bench.py --txns 4 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --quad-threshold-log2 13 > $O/r2b_sq_mfma.log 2>&1; tail -2 $O/r2b_sq_mfma.log; fi
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2b_64txn -- python $R/bench.py --txns 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/r2b_64txn.log 2>&1
cd $R
for d in r2b_4txn r2b_64txn; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp "$f" $O/${d}_kernel_stats.csv; done
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
# per-kernel sums of the SQ counters (one row per kernel name)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(O, "r2b_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
with open(os.path.join(O, "r2b_sq_counters_by_kernel.txt"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES (and a second pass with the MFMA counters) -- python bench.py --txns 4 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --quad-threshold-log2 13\n")
    tot = sum(v.get("SQ_INSTS_VALU", 0) for v in acc.values())
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:25]:
        out.write("%-62s launches %6d  VALU %.3e (%.1f %%)  waves %.3e  %s\n" % (k, cnt[k], v.get("SQ_INSTS_VALU", 0), 100 * v.get("SQ_INSTS_VALU", 0) / max(tot, 1), v.get("SQ_WAVES", 0), " ".join("%s %.3e" % (n, x) for n, x in sorted(v.items()) if "MFMA" in n)))
    out.write("total VALU wave-instructions %.4e\n" % tot)
print(open(os.path.join(O, "r2b_sq_counters_by_kernel.txt")).read())
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
python tools/kernel_bench.py > gpurun_out/r2b_kernel_bench.txt 2>&1
python tools/concurrent_hash_probe.py > gpurun_out/r2b_concurrent_hash_probe.txt 2>&1
for n in 16 32 64 128; do python bench.py --txns $n --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('txns',d['config']['txns_per_block'],'streams',d['config']['prover_streams_per_gpu'],'value',d['value'],'ms_per_block',d['ms_per_step'])"; done > gpurun_out/r2b_block_size_series.txt 2>&1; cat gpurun_out/r2b_block_size_series.txt
