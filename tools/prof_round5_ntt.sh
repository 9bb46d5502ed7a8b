#!/bin/bash
# Round 5: what the NTT / LDE block kernels wait for.  Run on the GPU box: bash tools/prof_round5_ntt.sh
# One rocprofv3 --pmc pass per counter group (python itself after `--`, --kernel-trace only), over tools/ntt_stall_probe.py.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p "$O"
export GPU_MAX_HW_QUEUES=32
cd /tmp && export TMPDIR=/tmp
python "$R/tools/ntt_stall_probe.py" --time --both > "$O/r5_ntt_times.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/r5_ntt_pmc$i" -- python "$R/tools/ntt_stall_probe.py" > "$O/r5_ntt_pmc$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$O/r5_ntt_times.txt"
done
cd "$R" && python tools/prof_round5_ntt_summary.py > "$O/r5_ntt_stalls.txt" 2>&1 || true
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -size +20M -delete
