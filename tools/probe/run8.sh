for R in 1 0; do
export RX_OPT_DECODE_RESIDENT=$R
echo "=== resident=$R"
RX_LIB_NAME=libradix_hip.so RX_CFLAGS= RX_VARIANT_SOURCES= SHAPES=128x4096 HQ=8 HKV=1 SPLITS=2,4,8,16 NOSTAMPS=1 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
RX_LIB_NAME=libradix_hip.so RX_CFLAGS= RX_VARIANT_SOURCES= SHAPES=256x4096 HQ=4 HKV=1 SPLITS=1,4,8,16 NOSTAMPS=1 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
RX_LIB_NAME=libradix_hip.so RX_CFLAGS= RX_VARIANT_SOURCES= SHAPES=64x2176 HQ=32 HKV=8 SPLITS=1,2,4,8 NOSTAMPS=1 timeout 300 python tools/decode_timeline.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
