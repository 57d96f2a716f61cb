export DIMS=128x128
for i in 1 2; do
echo -n "plain "; python3 tools/extend_dims.py 2>/dev/null | tail -1
echo -n "q_pack 4 "; RX_EXTEND_QPACK=4 python3 tools/extend_dims.py 2>/dev/null | tail -1
done
