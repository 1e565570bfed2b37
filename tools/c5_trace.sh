# rocprofv3 kernel trace of a hill-climb config (default C5) with the default issue lanes: device busy time (union of the kernel
# intervals) against the span, per-kernel totals, and the sweep launches bucketed by grid size.
#   bash tools/c5_trace.sh [max_iters] [output dir under gpurun_out] [hc config: c5mmhc, cv64, c3]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ITERS=${1:-1000000}
OUT=${2:-c5_trace}
HC=${3:-c5mmhc}
rm -rf $R/gpurun_out/$OUT
mkdir -p $R/gpurun_out/$OUT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$OUT -- python3 $R/bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $HC --hc-max-iters $ITERS --steps 1 --warmup 1 > $R/gpurun_out/$OUT.log 2>&1
cd $R
t=$(find gpurun_out/$OUT -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv, sys, collections, re
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", ""), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the hill-climb part: from the first to the last launch of the pruning tables' kernel (neither the headline step nor MMPC use it)
idx = [i for i, r in enumerate(rows) if "tile_box_kernel" in r[2]]
hc = rows[idx[0]:min(idx[-1] + 8, len(rows))]
span = hc[-1][1] - hc[0][0]
busy, cur_s, cur_e = 0, hc[0][0], hc[0][1]
for s, e, _, _ in hc[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in hc:
    k = re.sub(r"\(.*", "", n)[:110]
    tot[k][0] += e - s
    tot[k][1] += 1
print(f"hill-climb span {span / 1e9:.2f} s, device busy (union) {busy / 1e9:.2f} s, sum of kernel durations {sum(v[0] for v in tot.values()) / 1e9:.2f} s, {len(hc)} launches")
for k, (ns, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{ns / 1e9:7.2f} s {n:7d}  {ns / n / 1e3:8.1f} us  {k}")
b = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in hc:
    if "kde_sweep" in n:
        key = 1 << max(0, g - 1).bit_length()
        b[key][0] += e - s
        b[key][1] += 1
print("sweep launches by grid size (workgroups, rounded up to a power of two):")
for k in sorted(b):
    print(f"  <= {k:7d}: {b[k][1]:6d} launches {b[k][0] / 1e9:6.2f} s  {b[k][0] / b[k][1] / 1e3:8.1f} us each")
PY
tail -1 gpurun_out/$OUT.log | cut -c1-300
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
