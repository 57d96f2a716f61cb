for i in 1 2; do
  for L in libradix_hip.so libradix_r1.so; do
    for T in 0 1; do
      echo -n "$L t64=$T sep: "; env PS=64 RX_LIB_NAME=$L RX_OPT_DECODE_MLA8_T64=$T FP8=1 python tools/mla_bench.py 2>&1 | tail -1
      echo -n "$L t64=$T mc: "; env MC=1 RX_OPT_MERGE_IN_KERNEL_MAX_MB_MLA=64 PS=64 RX_LIB_NAME=$L RX_OPT_DECODE_MLA8_T64=$T FP8=1 python tools/mla_bench.py 2>&1 | tail -1
    done
  done
done
