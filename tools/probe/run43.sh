timeout 900 python -m pytest tests -m gpu -x -q -k "mla" 2>&1 | tail -2
for i in 1 2; do
echo new; python3 tools/mla_extend_bench.py 2>/dev/null | tail -2
echo old; RX_LIB_NAME=libradix_hip_xmlaold.so python3 tools/mla_extend_bench.py 2>/dev/null | tail -2
done
