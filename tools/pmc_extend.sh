cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
i=0
rm -rf $R/gpurun_out/pmcx; for P in "$P1" "$P2" "$P3"; do i=$((i+1)); rocprofv3 --pmc $P -d $R/gpurun_out/pmcx/p$i -o p$i --output-format csv -- python3 $R/bench.py --extend-only > $R/gpurun_out/pmcx_$i.log 2>&1; done
python3 $R/tools/pmc_kernel.py $R/gpurun_out/pmcx extend_mfma32 | tee $R/gpurun_out/pmcx_summary.json
