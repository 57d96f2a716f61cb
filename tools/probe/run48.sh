export DIMS=128x128
for i in 1 2; do for l in libradix_hip.so libradix_hip_padv50.so libradix_hip_padv100.so libradix_hip_pads50.so libradix_hip_pads100.so; do echo -n "$l "; RX_LIB_NAME=$l python3 tools/extend_dims.py 2>/dev/null | tail -1; done; done
