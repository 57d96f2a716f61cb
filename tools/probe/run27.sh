cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TS=2048 CASES=0
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/het0 -o het0 -- python3 $R/tools/probe/hetero_sweep.py 2>&1 | grep -v amdgpu | tail -3
export TS=1024 CASES=3
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/het3 -o het3 -- python3 $R/tools/probe/hetero_sweep.py 2>&1 | grep -v amdgpu | tail -3
cd $R; for d in het0 het3; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); echo $f; head -4 $f | cut -c1-200; done
