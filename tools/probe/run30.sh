timeout 900 python -m pytest tests/test_gpu_split_items.py tests/test_gpu_backend.py tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/probe/policy_sweep.py 2>&1 | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline --no-extend --bs 32 --ctx 1024 --layers 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['us_per_layer'],1),v['splits_of_the_long_request'],round(v['frac_of_hbm_peak'],3)) for k,v in d['heterogeneous_decode'].items() if isinstance(v,dict)})"
