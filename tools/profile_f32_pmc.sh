# SQ counters (separate --pmc passes, never with a trace domain) of the fp32 headline sweep (bench.py --dtype f32): where the issue slots go.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/f32_pmc; rm -rf $OUT; mkdir -p $OUT
CMD="python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --hc none --no-e2e --no-c3 --no-extra-legs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
vals = {}
for grp in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "kde_sweep_f16_kernel" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        vals[k] = v / max(n, 1)
        print(f"{k}: {v / max(n, 1):.5g} per dispatch ({n} dispatches)")
if "GRBM_GUI_ACTIVE" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8
    simd = cyc * 1024
    print(f"derived: {cyc:.0f} GPU cycles per dispatch = {cyc / 2.4e6:.2f} ms at 2.4 GHz")
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INST_CYCLES_VMEM"):
        if k in vals:
            print(f"   {k} / SIMD cycles = {vals[k] / simd:.3f}")
    if "SQ_INSTS_VALU" in vals:
        print(f"   VALU instructions per pair value: {vals['SQ_INSTS_VALU'] * 64 / 1e11:.3f} (wave instructions x 64 lanes / 1e11 pairs); MFMA per 1024 pairs: {vals.get('SQ_INSTS_MFMA', 0) * 1024 / 1e11:.2f}")
PY
find $OUT -name "*.csv" -size +1M -delete
