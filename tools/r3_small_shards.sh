#!/bin/bash
# Small shards (what one rank of an 8-GPU run proves of the 256-txn block) and a lone txn pair, Merkle levels of at
# most 4096 nodes fused (1: up to seven per launch, matrix-core one-set form), fused by load (-1) or one launch per
# level (0).  VERDICT r2 next #6.
set -o pipefail
mkdir -p gpurun_out
for txns in 2 16 32; do
  for fused in 0 1 -1; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile --txns $txns --steps 4 --warmup 1 --merkle-fused $fused \
        > gpurun_out/r3_shard_${txns}_f$fused.json 2> gpurun_out/r3_shard.err || { echo "FAILED $txns $fused"; tail -5 gpurun_out/r3_shard.err; exit 1; }
    python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r3_shard_${txns}_f$fused.json') if l.startswith('{')][-1])
print('txns=$txns merkle_fused=$fused: %.2f txn-proofs/s, %.1f ms per block, streams %d' % (d['value'], d['ms_per_step'], d['config']['prover_streams_per_gpu']), d['config']['ms_of_each_step_rank0'])
PY
  done
done
