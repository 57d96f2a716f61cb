export RX_EXT_PW=2
timeout 300 python tools/pw_debug.py 2>&1 | grep -v amdgpu | grep -c "nan 0"
timeout 300 python tools/pw_debug.py 2>&1 | grep -v amdgpu | awk '{ if ($(NF-2)+0 > 0.02) print }' | head
for v in "" _pw_abl4 _pw_nofence; do
  echo "== lib$v"; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
echo "== stamps"; RX_LIB_NAME=libradix_hip_pwstamp.so timeout 120 python tools/pw_stamps.py 2>&1 | tail -14 | head -7
