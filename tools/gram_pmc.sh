# Counters of the Gram kernel (2M x 64 fp64, tools/gram_bench.py), one --pmc pass per group (never with a trace domain):
# SQ issue / MFMA counters, LDS counters, FETCH_SIZE.  Prints per-dispatch averages; FETCH_SIZE is doubled as
# MI355X_MICROARCH.md "HBM" prescribes for 16-byte-per-lane streaming reads on gfx950 (raw value kept beside it).
# bash tools/gram_pmc.sh <outdir under gpurun_out>   (PBN_GRAM_LDS / PBN_GRAM_DEBUG / GRAM_MODE are inherited; GRAM_KERNEL = substring of
# the kernel name to aggregate, default gram_)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-gram_pmc}
rm -rf $OUT
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/tools/gram_bench.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_ANY --output-format csv -d $OUT/lds -- python3 $R/tools/gram_bench.py > $OUT/lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/gram_bench.py > $OUT/fetch.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
vals = {}
for grp in ("sq", "lds", "fetch"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    name = None
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if os.environ.get("GRAM_KERNEL", "gram_") in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]:
                name = row["Kernel_Name"].split("(")[0]
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        vals[k] = v / max(n, 1)
        print(f"{grp} {k}: {v / max(n, 1):.5g} per dispatch ({n} dispatches of {name})")
if "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
    print(f"derived: {cyc:.0f} GPU cycles per dispatch; MFMA busy per SIMD {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles = "
          f"{vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.3f} of them")
if "FETCH_SIZE" in vals:   # KB
    print(f"derived: FETCH_SIZE {vals['FETCH_SIZE'] / 1e6:.4f} GB raw, x2 = {2 * vals['FETCH_SIZE'] / 1e6:.4f} GB (algorithmic 1.024 GB)")
PY
# per-kernel per-dispatch averages + the blob hash of stats_kernels.hip they were taken on: copy to profiles/rN/gram_pmc.json (bench.py: secondary.roofline.mfma_busy)
python3 tools/pmc_aggregate.py $OUT/gram_pmc.json $OUT/sq $OUT/lds $OUT/fetch
find $OUT -name "*counter_collection.csv" -size +2M -delete
