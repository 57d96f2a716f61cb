export DIMS=64x64
timeout 1200 python -m pytest tests -m gpu -x -q -k "extend or baseline or backend or window or d64 or config" 2>&1 | tail -2
for i in 1 2; do
echo -n "new "; python3 tools/extend_dims.py 2>/dev/null | tail -1
echo -n "old "; RX_LIB_NAME=libradix_hip_e64old.so python3 tools/extend_dims.py 2>/dev/null | tail -1
done
