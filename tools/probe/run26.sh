export TS=512,768,1024,1536,2048,3072,4096 CASES=012
echo "=== default"; timeout 300 python tools/probe/hetero_sweep.py 2>&1 | grep -v amdgpu
echo "=== MINW 3"; RX_LIB_NAME=libradix_hip_dec3.so timeout 300 python tools/probe/hetero_sweep.py 2>&1 | grep -v amdgpu
