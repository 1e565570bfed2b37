# kernel stats of the C4 Gram pass (2M x 64 fp64, tools/gram_bench.py): gram kernel duration under rocprofv3
# (name, calls, total ns, average ns, %, min, max, stddev) plus the per-call durations of the partial-sum kernel - the first
# calls of a process run cold, the median is the steady figure.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-gram_prof}
rm -rf $R/gpurun_out/$OUT
mkdir -p $R/gpurun_out/$OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT -- python3 $R/tools/gram_bench.py > $R/gpurun_out/$OUT.log 2>&1
cd $R
f=$(find gpurun_out/$OUT -name "*kernel_stats.csv" | head -1)
grep -i "gram" $f | cut -c1-200
t=$(find gpurun_out/$OUT -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv, sys, statistics
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1]))
     if "gram_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
print("per call (us):", " ".join(f"{x / 1e3:.1f}" for x in d), "| median", f"{statistics.median(d) / 1e3:.1f}")
PY
tail -1 gpurun_out/$OUT.log
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
