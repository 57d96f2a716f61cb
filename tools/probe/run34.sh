cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DIMS=256x256,64x64
python3 $R/tools/extend_dims.py 2>/dev/null | tail -2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/dims_pmc1 -- python3 $R/tools/extend_dims.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/dims_pmc2 -- python3 $R/tools/extend_dims.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/dims_pmc3 -- python3 $R/tools/extend_dims.py > /dev/null 2>&1
cd $R
for k in extend_d256 extend_nd extend32 extend_kernel; do echo "== $k"; python3 tools/pmc_kernel.py gpurun_out/dims_pmc1 $k | tr -d '\n' | cut -c1-600; echo; python3 tools/pmc_kernel.py gpurun_out/dims_pmc2 $k | tr -d '\n' | cut -c1-700; echo; python3 tools/pmc_kernel.py gpurun_out/dims_pmc3 $k | tr -d '\n' | cut -c1-700; echo; done
