export DIMS=256x256,192x128,192x192
timeout 900 python -m pytest tests -m gpu -x -q -k "256 or 192 or dims or d256 or cascade or baseline" 2>&1 | tail -2
for i in 1 2; do
echo "new"; python3 tools/extend_dims.py 2>/dev/null | tail -3
echo "old (previous commit)"; RX_LIB_NAME=libradix_hip_d256old.so python3 tools/extend_dims.py 2>/dev/null | tail -3
done
