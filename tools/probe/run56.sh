timeout 900 python tools/probe/rounds_rule.py 2>&1 | grep -v amdgpu
timeout 1200 python -m pytest tests/test_gpu_split_items.py tests/test_gpu_backend.py tests/test_gpu_parity.py tests/test_cascade_groups.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-extend --no-radix-hit 2>/dev/null | tail -1 > gpurun_out/bench56.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench56.json')); r=d['roofline']
print(round(d['value']), d['ms_per_step'], r['frac'])
print({k:(round(v['us_per_layer'],1),v['splits_of_the_long_request'],round(v['frac_of_hbm_peak'],3)) for k,v in d['heterogeneous_decode'].items() if isinstance(v,dict)})
PY
