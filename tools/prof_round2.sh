set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2p_4txn -- python $R/bench.py --txns 4 --threads 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/r2p_4txn.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r2p_pmc_f -- python $R/bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2p_pmc_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r2p_pmc_w -- python $R/bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2p_pmc_w.log 2>&1 &&
cd $R && python tools/pmc_family_traffic.py gpurun_out/r2p_pmc_f gpurun_out/r2p_pmc_w gpurun_out/r2p_pmc_f.log > gpurun_out/r2_pmc_lde_family.txt 2>&1; cat gpurun_out/r2_pmc_lde_family.txt
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/r2p_sq -- python $R/tools/pmc_probe.py > $O/r2p_sq.log 2>&1; tail -2 $O/r2p_sq.log
ls $O/r2p_4txn/*/ $O/r2p_pmc_f/*/ | head -20
