export RX_EXT_PW=2
for rep in 1 2; do
for v in "" _pw_r6 _pw_r8; do
  echo -n "== lib$v "; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
done
RX_LIB_NAME=libradix_hip_pw_r8.so timeout 300 python tools/pw_debug.py 2>&1 | grep -v amdgpu | grep -c "nan 0"
