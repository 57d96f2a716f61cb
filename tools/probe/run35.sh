cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DIMS=256x256,192x128,192x192
python3 $R/tools/extend_dims.py 2>/dev/null | tail -3
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/dims2_pmc1 -- python3 $R/tools/extend_dims.py > /dev/null 2>&1
cd $R
python3 tools/pmc_kernel.py gpurun_out/dims2_pmc1 "extend_d256_kernel<rx::BF16, 256" | tr -d '\n' | cut -c1-300; echo
timeout 900 python -m pytest tests -m gpu -x -q -k "d256 or 256 or 192 or dims or cascade" 2>&1 | tail -3
