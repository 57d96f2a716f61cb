export RX_EXT_PW=2
for v in "" _pw_nosm_abl0 _pw_nosm_abl4 _pw_nosm_abl6 _pw_nosm_abl14 _pw_abl2 _pw_abl4 _pw_abl8; do
  echo "== lib$v"; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
echo "== nosm stamps"; RX_LIB_NAME=libradix_hip_pw_nosm_stamp.so timeout 120 python tools/pw_stamps.py 2>&1 | tail -9 | head -5
