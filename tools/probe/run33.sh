for cfg in "GROUPS=4 PER=64" "GROUPS=8 PER=32" "GROUPS=16 PER=16" "GROUPS=3 PER=64 LONERS=64" "GROUPS=2 PER=8 SHARED=8192 UNIQ=256" "GROUPS=4 PER=64 SHARED=1024 UNIQ=1024"; do
  env $cfg timeout 300 python tools/cascade_groups_bench.py 2>&1 | grep -v amdgpu
done
