export RX_EXT_PW=2
for v in "" _pw_dmaearly _pw_nofence _pw_nosm _pw_ka3 _pw_nosm_nodma; do
  echo "== lib$v"; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
echo "== stamps"; RX_LIB_NAME=libradix_hip_pwstamp.so timeout 120 python tools/pw_stamps.py 2>&1 | tail -8
echo "== stamps nodma"; RX_LIB_NAME=libradix_hip_pw_stamp_nodma.so timeout 120 python tools/pw_stamps.py 2>&1 | tail -8 | head -3
