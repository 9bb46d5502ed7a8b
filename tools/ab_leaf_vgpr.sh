set -e
L=proof_protocol_decoder_amd/lib
for v in loose tight loose tight; do
  cp $L/libbpg_$v.so $L/libbpg.so
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json;d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]);print('$v',d['value'])"
done
