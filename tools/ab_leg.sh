# does the single-stream leg in front of the block run cost block rate, and for how many blocks?
for args in "--steps 4 --warmup 1" "--steps 4 --warmup 3" "--steps 4 --warmup 1 --no-profile"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline $args > gpurun_out/abg.json 2> gpurun_out/abg.err || { echo "FAILED $args"; tail -3 gpurun_out/abg.err; continue; }
  python -c "
import json;d=json.loads(open('gpurun_out/abg.json').read().strip().splitlines()[-1]);print('$args ->',d['value'])"
done
