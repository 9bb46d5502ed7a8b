#!/bin/bash
# Order / instrumentation effects on the headline, clocks and power of OUR card sampled:
#   A  --no-profile                      (no leg, no HIP events in the timed region)
#   D  default, --no-in-situ-profile     (leg in a child AFTER the timed region, no events in it)
#   C  default                           (leg after, events around every LDE launch of the timed region)
#   B  --leg-first                       (round 2's order: leg in this process BEFORE the block)
#   A2 --no-profile again                (drift of the box over the experiment)
set -o pipefail
mkdir -p gpurun_out
for tag in A_noprofile D_no_in_situ C_default B_legfirst A2_noprofile; do
  case $tag in
    A_noprofile|A2_noprofile) extra="--no-profile";;
    D_no_in_situ) extra="--no-in-situ-profile";;
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
