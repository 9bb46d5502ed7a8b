set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2p_4txn -- python $R/bench.py --txns 4 --threads 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/r2p_4txn.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r2p_pmc_f -- python $R/bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2p_pmc_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r2p_pmc_w -- python $R/bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2p_pmc_w.log 2>&1 &&
cd $R && python tools/pmc_family_traffic.py gpurun_out/r2p_pmc_f gpurun_out/r2p_pmc_w gpurun_out/r2p_pmc_f.log > gpurun_out/r2_pmc_lde_family.txt 2>&1; echo "PMC passes taken at HEAD ac91bd2" >> gpurun_out/r2_pmc_lde_family.txt; cat gpurun_out/r2_pmc_lde_family.txt
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/r2p_sq -- python $R/tools/pmc_probe.py > $O/r2p_sq.log 2>&1; tail -2 $O/r2p_sq.log
find $O -name "*_kernel_trace.csv" -delete
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2p_64txn -- python $R/bench.py --txns 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/r2p_64txn.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
cd $R && python tools/ntt_sweep.py > gpurun_out/r2_ntt_sweep.md 2>&1
python tools/kernel_bench.py > gpurun_out/r2_kernel_bench.txt 2>&1
python tools/concurrent_hash_probe.py > gpurun_out/r2_concurrent_hash_probe.txt 2>&1
python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
tail -c 1500 gpurun_out/r2_bench.json
