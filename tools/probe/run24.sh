for v in "" _dec3; do
  echo "== lib$v"
  RX_LIB_NAME=libradix_hip$v.so timeout 300 python tools/hetero_decode.py 2>&1 | grep -v amdgpu | grep "lens\|single\|balanced, max 32\|K3 formula, max  8"
  RX_LIB_NAME=libradix_hip$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extend --no-radix-hit --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('decode tok/s',round(d['value']),'frac',round(d['roofline']['frac'],4),'hetero',{k:round(v['us_per_layer'],1) for k,v in d['heterogeneous_decode'].items() if isinstance(v,dict)})
print('mla', {k:(round(v['us'],1),round(v['kernel_us'],1)) for k,v in d['mla_decode'].items() if isinstance(v,dict)})"
done
