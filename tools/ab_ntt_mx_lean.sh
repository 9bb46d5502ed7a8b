set -e
L=proof_protocol_decoder_amd/lib
for v in w3 w4; do
  cp $L/libbpg_$v.so $L/libbpg.so
  echo "== $v"
  timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ntt or lde" 2>&1 | tail -1
  timeout -k 10 200 python tools/ntt_mx_probe.py 2>/dev/null | cut -c1-140
  for m in 1 0; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --ntt-mx $m > gpurun_out/abl.json 2> gpurun_out/abl.err
    python -c "
import json;d=json.loads(open('gpurun_out/abl.json').read().strip().splitlines()[-1]);print('$v ntt_mx=$m block rate',d['value'])"
  done
done
cp $L/libbpg_w3.so $L/libbpg.so
