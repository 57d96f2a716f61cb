R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "mla or extend or dims or cascade or baseline or window or nd" 2>&1 | tail -3
export PS=64
FP8=1 python3 tools/mla_bench.py 2>&1 | tail -1
python3 tools/mla_bench.py 2>&1 | tail -1
python3 tools/extend_dims.py 2>/dev/null | tail -6
python3 tools/mla_extend_bench.py 2>/dev/null | tail -3
cd /tmp; export TMPDIR=/tmp
FP8=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/mla8b_pmc1 -- python3 $R/tools/mla_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/mla16b_pmc1 -- python3 $R/tools/mla_bench.py > /dev/null 2>&1
cd $R
python3 tools/pmc_kernel.py gpurun_out/mla8b_pmc1 decode_mla8 | tr -d '\n' | sed 's/"_launches.*//'; echo
python3 tools/pmc_kernel.py gpurun_out/mla16b_pmc1 decode_mla_kernel | tr -d '\n' | sed 's/"_launches.*//'; echo
