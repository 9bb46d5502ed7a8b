for args in "--merkle-fused 0" "--merkle-fused 1" "--merkle-fused 0" "--merkle-fused 1"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile $args > gpurun_out/abf.json 2> gpurun_out/abf.err || { echo "FAILED $args"; tail -3 gpurun_out/abf.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/abf.json').read().strip().splitlines()[-1]);print('$args ->',d['value'])"
done
