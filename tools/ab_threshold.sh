# block rate against the mx set threshold (4 sets per wave from 2^k items up) and the number of prover streams
for args in "--quad-threshold-log2 8" "--quad-threshold-log2 11" "--quad-threshold-log2 13" "--quad-threshold-log2 15" "--threads 16" "--threads 32 --extra-workers 0"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile $args > gpurun_out/abt.json 2> gpurun_out/abt.err || { echo "FAILED $args"; tail -3 gpurun_out/abt.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/abt.json').read().strip().splitlines()[-1]);print('$args ->',d['value'])"
done
