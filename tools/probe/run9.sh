export RX_EXT_PW=2
for v in "" _pw_nosm_nodma _pw_nosm_abl6 _pw_nosm_abl14 _pw_nosm_abl2 _pw_abl8; do
  echo "== lib$v"; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pw_pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --extend-only > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pw_pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --extend-only > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pw_pmc1","gpurun_out/pw_pmc2"):
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in acc.items():
            if "extend" in k: print(k, {a:round(b) for a,b in v.items()})
PY
