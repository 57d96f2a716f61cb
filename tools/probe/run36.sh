export FP8=1 PS=64
for v in "S=8" "S=8 MC=1" "S=4" "S=4 MC=1" "S=16 MC=1" "S=8 STAMPS=1"; do echo "== $v"; env $v timeout 120 python tools/mla_bench.py 2>&1 | grep -v amdgpu | tail -3; done
