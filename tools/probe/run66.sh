export DIMS=128x128
for tp in 8 4; do for i in 1 2; do
echo -n "TP=$tp auto "; TP=$tp python3 tools/extend_dims.py 2>/dev/null | tail -1
echo -n "TP=$tp off  "; TP=$tp RX_EXT32_AUTOPACK=0 python3 tools/extend_dims.py 2>/dev/null | tail -1
done; done
