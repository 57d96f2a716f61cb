export DIMS=128x128
for i in 1 2; do
echo -n "extend32 "; python3 tools/extend_dims.py 2>/dev/null | tail -1
echo -n "d256 form "; RX_EXT_D256_AT128=1 python3 tools/extend_dims.py 2>/dev/null | tail -1
done
RX_EXT_D256_AT128=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extend" 2>&1 | tail -2
