#!/bin/bash
# SQ-counter passes (rocprofv3 --pmc only, one counter family per pass; program directly after --):
#   D = 128 extend (the one D = 128 kernel); D = 256 / 64 / 192x128 extend; MLA decode at page_size 64.
# bash tools/pmc_round.sh   (through gpurun; then python tools/pmc_round_summary.py gpurun_out/pmc profiles <tag>)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  rocprofv3 --pmc $P -d $O/ext32_p$i -o p --output-format csv -- python3 $R/bench.py --extend-only > $O/ext32_$i.log 2>&1
  DIMS=256x256,64x64,192x128 rocprofv3 --pmc $P -d $O/dims_p$i -o p --output-format csv -- python3 $R/tools/extend_dims.py > $O/dims_$i.log 2>&1
  PS=64 FP8=1 rocprofv3 --pmc $P -d $O/mla8_p$i -o p --output-format csv -- python3 $R/tools/mla_bench.py > $O/mla8_$i.log 2>&1
  PS=64 rocprofv3 --pmc $P -d $O/mla16_p$i -o p --output-format csv -- python3 $R/tools/mla_bench.py > $O/mla16_$i.log 2>&1
done
# the D = 128 extend kernel's issue / wait classes: two more passes, this kernel only
P3="SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS"
P4="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_CVT"
i=2
for P in "$P3" "$P4"; do i=$((i+1))
  rocprofv3 --pmc $P -d $O/ext32_p$i -o p --output-format csv -- python3 $R/bench.py --extend-only > $O/ext32_$i.log 2>&1
done
ls $O
