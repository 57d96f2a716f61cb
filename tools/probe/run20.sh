for pwm in 0 2; do for z in "" 1; do
  echo -n "pw=$pwm zero=$z: "; RX_EXT_PW=$pwm RX_EXTEND_ZERO=$z timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done; done
