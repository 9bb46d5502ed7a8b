# small shards (what a rank of an N-GPU run proves): block rate against the number of prover streams
for args in "--txns 32 --threads 16" "--txns 32 --threads 24" "--txns 32 --threads 32" "--txns 64 --threads 24" "--txns 64 --threads 32" "--txns 16 --threads 8" "--txns 16 --threads 16"; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile $args > gpurun_out/abs.json 2> gpurun_out/abs.err || { echo "FAILED $args"; tail -3 gpurun_out/abs.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/abs.json').read().strip().splitlines()[-1]);print('$args ->',d['value'],'ms/block',d['ms_per_step'])"
done
