#!/bin/bash
# Order effects of the single-stream leg: the same block run (a) alone, (b) after the leg in the
# same process (round 2's order), (c) with the leg in a child AFTER it (round 3's order), clocks / power sampled.
set -o pipefail
mkdir -p gpurun_out
for tag in A_noprofile B_legfirst C_default A2_noprofile; do
  case $tag in
    A_noprofile|A2_noprofile) extra="--no-profile";;
    B_legfirst) extra="--leg-first";;
    C_default) extra="";;
  esac
  timeout -k 10 500 python tools/smi_sampler.py gpurun_out/r3_smi_$tag.csv -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 --phase-marks $extra \
      > gpurun_out/r3_order_$tag.json 2> gpurun_out/r3_order_$tag.err || { echo "FAILED $tag"; tail -5 gpurun_out/r3_order_$tag.err; exit 1; }
  python - <<PY
import json
lines=[l for l in open('gpurun_out/r3_order_$tag.json') if l.startswith('{')]
d=json.loads(lines[-1]); print('$tag', d['value'], d['config']['ms_of_each_step_rank0'])
PY
  grep -A12 "phase summary" gpurun_out/r3_order_$tag.json || true
done
