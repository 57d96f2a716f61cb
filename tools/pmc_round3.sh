#!/bin/bash
# Round-3 SQ-counter passes (rocprofv3 --pmc only, one counter family per pass; program directly after --):
#   D = 128 extend, eight-wave kernel and (RX_EXT_PW=2) the one-wave-per-SIMD kernel; D = 256 / 64 extend; MLA decode.
# bash tools/pmc_round3.sh   (through gpurun; then python tools/pmc_round3_summary.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc3
rm -rf $O; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  rocprofv3 --pmc $P -d $O/ext32_p$i -o p --output-format csv -- python3 $R/bench.py --extend-only > $O/ext32_$i.log 2>&1
  RX_EXT_PW=2 rocprofv3 --pmc $P -d $O/pw_p$i -o p --output-format csv -- python3 $R/bench.py --extend-only > $O/pw_$i.log 2>&1
  DIMS=256x256,64x64,192x128 rocprofv3 --pmc $P -d $O/dims_p$i -o p --output-format csv -- python3 $R/tools/extend_dims.py > $O/dims_$i.log 2>&1
  PS=64 FP8=1 rocprofv3 --pmc $P -d $O/mla8_p$i -o p --output-format csv -- python3 $R/tools/mla_bench.py > $O/mla8_$i.log 2>&1
  PS=64 rocprofv3 --pmc $P -d $O/mla16_p$i -o p --output-format csv -- python3 $R/tools/mla_bench.py > $O/mla16_$i.log 2>&1
done
ls $O
