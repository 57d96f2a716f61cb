for shape in 3584,512,32 0,2048,8 2048,2048,8 512,128,64 8192,1024,8 0,8192,2; do
  for mode in 0 2; do
    for v in "" _pw_a3; do
      if [ "$mode" = "0" ] && [ "$v" != "" ]; then continue; fi
      echo -n "shape $shape pw=$mode lib$v: "; RX_EXT_PW=$mode RX_EXTEND_SHAPE=$shape RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
    done
  done
done
