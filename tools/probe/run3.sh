set -x
for LIB in libradix_hip_tl.so libradix_hip_tl2.so; do
  export RX_LIB_NAME=$LIB
  echo "=== $LIB"
  SHAPES=128x4096 HQ=8 HKV=1 SPLITS=1,2,4 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids
  ITEMS=0 SHAPES=128x4096 HQ=8 HKV=1 SPLITS=2 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids
  SHAPES=256x4096 HQ=4 HKV=1 SPLITS=1,2 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids
  SHAPES=64x2176 HQ=32 HKV=8 SPLITS=1 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids
  SHAPES=256x4096 HQ=32 HKV=8 SPLITS=1 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids
done
unset RX_LIB_NAME
python bench.py --model llama3-70b --tp-sim 8 --bs 128 --ctx 4096 --layers 80 --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline'],indent=1))"
python bench.py --extend-only 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ext32', d['kernel'], d['kernel_only'], d.get('sustained_clock'))"
RX_OPT_EXTEND_D256_AT128=1 python bench.py --extend-only 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('d256_at128', d['kernel'], d['kernel_only'], d.get('sustained_clock'))"
