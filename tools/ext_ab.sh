#!/bin/bash
# A/B of extend-kernel builds on the GPU box: tools/ext_ab.sh <outdir> <label>=<env assignments...> ...
# each variant runs `bench.py --extend-only` three times; prints label and TFLOP/s
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  for i in 1 2 3; do
    env $envs python bench.py --extend-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['tflops'],1), round(d['ms_per_chunk'],4))"
  done
done | tee "$out/ab.txt"
