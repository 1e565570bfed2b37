# SQ counters of the Gram kernel (2M x 64 fp64, tools/gram_bench.py), one --pmc pass per group (never with a trace domain).
# bash tools/gram_pmc.sh <outdir under gpurun_out>   (PBN_GRAM_LDS / PBN_GRAM_DEBUG are inherited)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-gram_pmc}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/tools/gram_bench.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/lds -- python3 $R/tools/gram_bench.py > $OUT/lds.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for grp in ("sq", "lds"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gram_" in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(f"{grp} {k}: {v / max(n, 1):.4g} per dispatch ({n} dispatches)")
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
