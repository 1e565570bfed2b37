# Effective clock and issue shares per kernel of a bench.py command: ONE rocprofv3 --pmc pass (no trace domain) whose counter rows carry the
# dispatch's own start / end timestamps - cycles (GRBM_GUI_ACTIVE / 8 XCDs) over duration is the clock the chip held under that kernel.
#   bash tools/clock_probe.sh <output name> <bench.py arguments...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=$1; shift
OUT=$R/gpurun_out/$NAME; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 $R/bench.py $* > $OUT/run.log 2>&1
cd $R
python3 - $OUT "$*" > $R/gpurun_out/$NAME.txt <<'PY'
import csv, glob, sys, collections
out, cmd = sys.argv[1:3]
print("# python3 bench.py " + cmd)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(dict)
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("pbn::", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
rows = []
for k, v in acc.items():
    ns = sum(disp[k].values())
    if "GRBM_GUI_ACTIVE" not in v or ns <= 0:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    rows.append((ns, k, len(disp[k]), cyc, v))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'launches':>8s} {'ms':>10s} {'GHz':>6s} {'VALU/cyc/SIMD':>13s} {'MFMA busy':>9s} {'wait_inst':>9s} {'wait_any':>8s} {'wave_cyc':>8s}")
for ns, k, n, cyc, v in rows[:14]:
    simd = cyc * 1024.0
    print(f"{k[:70]:70s} {n:8d} {ns / 1e6:10.3f} {cyc / ns:6.3f} {v.get('SQ_INSTS_VALU', 0) / simd:13.4f} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd:9.3f} "
          f"{v.get('SQ_WAIT_INST_ANY', 0) / simd:9.3f} {v.get('SQ_WAIT_ANY', 0) / simd:8.3f} {v.get('SQ_WAVE_CYCLES', 0) / simd:8.3f}")
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
cat $R/gpurun_out/$NAME.txt
