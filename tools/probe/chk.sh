export DIMS=96x96
for i in 1 2; do
echo -n "nd kernel "; RX_EXT_D256_AT96=0 python3 tools/extend_dims.py 2>/dev/null | tail -1
echo -n "d256 form "; python3 tools/extend_dims.py 2>/dev/null | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cascade.py -m gpu -x -q -k "other_head_dims or 96 or cascade" 2>&1 | tail -2
