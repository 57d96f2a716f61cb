cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PS=64
FP8=1 RX_LIB_NAME=libradix_hip_m8r2.so python3 $R/tools/mla_bench.py 2>&1 | tail -1
FP8=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/mla8_pmc1 -- python3 $R/tools/mla_bench.py > /dev/null 2>&1
FP8=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/mla8_pmc2 -- python3 $R/tools/mla_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/mla16_pmc1 -- python3 $R/tools/mla_bench.py > /dev/null 2>&1
cd $R
python3 tools/pmc_kernel.py gpurun_out/mla8_pmc1 decode_mla8 | tr -d '\n' | sed 's/"_launches.*//'; echo
python3 tools/pmc_kernel.py gpurun_out/mla8_pmc2 decode_mla8 | tr -d '\n' | sed 's/"_launches.*//'; echo
python3 tools/pmc_kernel.py gpurun_out/mla16_pmc1 decode_mla_kernel | tr -d '\n' | sed 's/"_launches.*//'; echo
