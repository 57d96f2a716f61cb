export DIMS=256x256
for i in 1 2; do for l in libradix_hip.so libradix_hip_padv200.so libradix_hip_padv400.so libradix_hip_pads200.so libradix_hip_pads400.so; do echo -n "$l "; RX_LIB_NAME=$l python3 tools/extend_dims.py 2>/dev/null | tail -1; done; done
