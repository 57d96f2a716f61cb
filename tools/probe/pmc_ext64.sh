#!/bin/bash
# SQ-counter passes of the one-wave-per-SIMD extend kernel (option ext64) on the bench's extend chunk
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc64
rm -rf $O; mkdir -p $O
export RX_OPT_EXT64=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES"
P3="SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS"
P4="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU_CVT"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  rocprofv3 --pmc $P -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --extend-only --no-peaked > $O/log_$i.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "extend_mfma64" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(k, tot[k] / max(n[k],1), n[k])
PY
