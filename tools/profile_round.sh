#!/bin/bash
# Reproduces the committed profiles/ set on an MI355X box:  bash tools/profile_round.sh <tag>
# (run through gpurun; then `python profiles/summarize.py gpurun_out/prof <tag>` condenses the CSVs)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
# bench.py prints the full record on a '[bench-full] ' line and the compact line (what the driver keeps) last: both are kept
split_lines() {  # <stdout file> <stem>: <stem>.json = the compact line, <stem>_full.json = the full record
  python3 - "$1" "$2" <<'PY'
import sys
lines = open(sys.argv[1]).read().splitlines()
full = [ln[len("[bench-full] "):] for ln in lines if ln.startswith("[bench-full] ")]
js = [ln for ln in lines if ln.startswith("{")]
if js:
    open(sys.argv[2] + ".json", "w").write(js[-1] + "\n")
if full:
    open(sys.argv[2] + "_full.json", "w").write(full[-1] + "\n")
PY
}
python3 $R/bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err
split_lines $OUT/bench_default.out $OUT/${TAG}_bench_default
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-radix-hit --no-peaked --steps 2 --warmup 1 > $OUT/bench_kt.out 2> $OUT/kt.err
split_lines $OUT/bench_kt.out $OUT/${TAG}_bench_under_kernel_trace
# counter passes: eager launches (--no-graph), one counter family per pass, nothing but --pmc on the command line
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pf --output-format csv -- python3 $R/bench.py --no-extra --no-graph --no-cpu-baseline --no-extend --no-radix-hit --steps 2 --warmup 1 > /dev/null 2> $OUT/pf.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pw --output-format csv -- python3 $R/bench.py --no-extra --no-graph --no-cpu-baseline --no-extend --no-radix-hit --steps 2 --warmup 1 > /dev/null 2> $OUT/pw.err
# MLA decode kernels: HBM read bytes per launch at the bench's page_size 64 (one --pmc pass each, nothing else on the command line)
PS=64 rocprofv3 --pmc FETCH_SIZE -d $OUT/mla16_pmc -o p --output-format csv -- python3 $R/tools/mla_bench.py > /dev/null 2> $OUT/mla16_pmc.err
PS=64 FP8=1 rocprofv3 --pmc FETCH_SIZE -d $OUT/mla8_pmc -o p --output-format csv -- python3 $R/tools/mla_bench.py > /dev/null 2> $OUT/mla8_pmc.err
python3 $R/profiles/summarize.py $OUT $TAG
# SHORT decode launches (round 6): the shard legs' kernel under the kernel trace -- rocprofv3's duration of the decode kernel
# is what bench.py's back-to-back figure (GraphStep.probe_back_to_back) must agree with, not the single launch between events
rocprofv3 --kernel-trace --stats -d $OUT/c3 -o c3 --output-format csv -- python3 $R/bench.py --model llama3-70b --tp-sim 8 --bs 128 --ctx 4096 --layers 80 --no-extend --no-radix-hit --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/c3.out 2> $OUT/c3.err
split_lines $OUT/c3.out $OUT/${TAG}_config3_shard_under_kernel_trace
cp $(find $OUT/c3 -name "*_kernel_stats.csv" | head -1) $R/profiles/${TAG}_config3_shard_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/tp8 -o tp8 --output-format csv -- python3 $R/bench.py --tp-sim 8 --no-extend --no-radix-hit --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/tp8.out 2> $OUT/tp8.err
split_lines $OUT/tp8.out $OUT/${TAG}_tp8_shard_under_kernel_trace
cp $(find $OUT/tp8 -name "*_kernel_stats.csv" | head -1) $R/profiles/${TAG}_tp8_shard_kernel_stats.csv 2>/dev/null
cp $OUT/${TAG}_config3_shard_under_kernel_trace_full.json $OUT/${TAG}_tp8_shard_under_kernel_trace_full.json $R/profiles/ 2>/dev/null
# MLA decode (config 5 shape): kernel traces for 16-bit and fp8 latent rows
rocprofv3 --kernel-trace --stats -d $OUT/mla16 -o m --output-format csv -- python3 $R/tools/mla_bench.py > $OUT/mla16.txt 2> $OUT/mla16.err
FP8=1 rocprofv3 --kernel-trace --stats -d $OUT/mla8 -o m --output-format csv -- python3 $R/tools/mla_bench.py > $OUT/mla8.txt 2> $OUT/mla8.err
cp $(find $OUT/mla16 -name "*_kernel_stats.csv" | head -1) $R/profiles/${TAG}_mla_bf16_kernel_stats.csv 2>/dev/null
cp $(find $OUT/mla8 -name "*_kernel_stats.csv" | head -1) $R/profiles/${TAG}_mla_fp8_kernel_stats.csv 2>/dev/null
# latent MLA extend (rx::extend_mla_kernel): kernel trace of the bench shape
rocprofv3 --kernel-trace --stats -d $OUT/mlaext -o m --output-format csv -- python3 $R/tools/mla_extend_bench.py > $OUT/${TAG}_mla_extend.txt 2> $OUT/mlaext.err
cp $(find $OUT/mlaext -name "*_kernel_stats.csv" | head -1) $R/profiles/${TAG}_mla_extend_kernel_stats.csv 2>/dev/null
# shared-prefix (cascade) decode: per-kernel times of the radix-hit batch (plain vs cascade, 4 layer buffers)
rocprofv3 --kernel-trace --stats -d $OUT/casc -o casc --output-format csv -- python3 $R/tools/cascade_bench.py > $OUT/${TAG}_cascade_bench.txt 2> $OUT/casc.err
cp $OUT/casc/casc_kernel_stats.csv $R/profiles/${TAG}_cascade_kernel_stats.csv 2>/dev/null
# cascade over several shared prefixes (one per radix-tree node): plain vs ops.CascadeGroups
(python3 $R/tools/cascade_groups_bench.py; GROUPS=8 PER=32 python3 $R/tools/cascade_groups_bench.py; GROUPS=3 PER=64 LONERS=64 python3 $R/tools/cascade_groups_bench.py) > $OUT/${TAG}_cascade_groups_bench.txt 2> $OUT/cascg.err
cp $OUT/${TAG}_bench_default.json $OUT/${TAG}_bench_default_full.json $OUT/${TAG}_bench_under_kernel_trace.json $OUT/${TAG}_bench_under_kernel_trace_full.json $R/gpurun_out/ 2>/dev/null
mkdir -p $R/gpurun_out/profiles_new && cp $R/profiles/${TAG}_kernel_stats.csv $R/profiles/${TAG}_pmc_summary.json $R/profiles/${TAG}_cascade_kernel_stats.csv $R/profiles/${TAG}_mla_bf16_kernel_stats.csv $R/profiles/${TAG}_mla_fp8_kernel_stats.csv $R/profiles/${TAG}_mla_pmc_summary.json $OUT/${TAG}_cascade_bench.txt $OUT/mla16.txt $OUT/mla8.txt $R/profiles/${TAG}_mla_extend_kernel_stats.csv $OUT/${TAG}_mla_extend.txt $OUT/${TAG}_cascade_groups_bench.txt $R/gpurun_out/profiles_new/ 2>/dev/null
tail -c 1500 $OUT/${TAG}_bench_default.json
