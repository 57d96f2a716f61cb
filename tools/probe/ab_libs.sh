for i in 1 2 3; do
  for L in libradix_hip.so libradix_old.so; do
    RX_LIB_NAME=$L python bench.py --extend-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['kernel_only']['tflops'],1), round(d['tflops'],1))"
  done
done
