# SQ counters of ONE kernel family inside a bench.py command (separate --pmc passes, never with a trace domain; the program directly after `--`).
#   bash tools/pmc_kernel.sh <kernel substring> <output name> <bench.py arguments...>
# e.g. the tile-moment pass of C3's first iteration:
#   bash tools/pmc_kernel.sh kde_moment_group_kernel moment_pmc --no-e2e --no-cpu-baseline --no-extra-legs --hc c3 --hc-max-iters 1 --steps 1 --warmup 1
# Writes gpurun_out/<output name>.txt: per-dispatch means, the counters as shares of the SIMD cycles, and the git blob hash of kde_kernels.hip they were taken on.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KERNEL=$1; NAME=$2; shift 2
OUT=$R/gpurun_out/$NAME; rm -rf $OUT; mkdir -p $OUT
CMD="python3 $R/bench.py $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_MFMA --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
cd $R
python3 - $OUT "$KERNEL" "$CMD" > $R/gpurun_out/$NAME.txt <<'PY'
import csv, glob, sys, collections, hashlib, os
out, kern, cmd = sys.argv[1:4]
src = os.path.join(os.environ["GRAFT_REPO_ROOT"], "pybnesian_amd", "csrc", "kde_kernels.hip")
data = open(src, "rb").read()
blob = hashlib.sha1(("blob %d" % len(data)).encode() + bytes(1) + data).hexdigest()
print("# " + cmd)
print("# kernel filter: " + kern + "; kde_kernels.hip blob " + blob)
vals, per_kernel = {}, collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for grp in ("p1", "p2", "p3"):
    files = glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(f"# pass {grp}: no counter file (see {grp}.log: a counter of this pass may not exist on gfx950)")
    for f in files:
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                a = per_kernel[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for kname, acc in per_kernel.items():
    print(f"== {kname}")
    vals = {}
    for k, (v, n) in sorted(acc.items()):
        vals[k] = v / max(n, 1)
        print(f"{k}: {vals[k]:.5g} per dispatch ({n} dispatches)")
    if "GRBM_GUI_ACTIVE" in vals:
        cyc = vals["GRBM_GUI_ACTIVE"] / 8
        simd = cyc * 1024
        print(f"derived: {cyc:.0f} GPU cycles per dispatch = {cyc / 2.4e6:.3f} ms at 2.4 GHz; shares of the SIMD cycles (quad-cycle counters x 4 where the guide says so are NOT applied: raw ratios)")
        for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
                  "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_SCA", "SQ_VALU_MFMA_BUSY_CYCLES"):
            if k in vals:
                print(f"   {k} / SIMD cycles = {vals[k] / simd:.4f}")
        if "SQ_WAVE_CYCLES" in vals:
            for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if k in vals:
                    print(f"   {k} / SQ_WAVE_CYCLES = {vals[k] / vals['SQ_WAVE_CYCLES']:.4f}")
        if "SQ_INSTS_VALU" in vals:
            print(f"   VALU wave-instructions per GPU cycle and SIMD: {vals['SQ_INSTS_VALU'] / simd:.4f}; LDS per VALU instruction: {vals.get('SQ_INSTS_LDS', 0) / vals['SQ_INSTS_VALU']:.4f}")
        if "SQ_LDS_BANK_CONFLICT" in vals and vals.get("SQ_LDS_IDX_ACTIVE"):
            print(f"   LDS bank-conflict cycles / LDS active cycles = {vals['SQ_LDS_BANK_CONFLICT'] / vals['SQ_LDS_IDX_ACTIVE']:.4f}")
PY
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
