# where the grouped sweeps' time goes by dimension of the term (experiments build, PBN_SWEEP_LOG=1: one line per grouped sweep launch)   bash tools/dim_share.sh
cd $GRAFT_REPO_ROOT
export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so PBN_SWEEP_LOG=1
for leg in ${LEGS:-"cv64 1" "c3 1" "c3 6" "c5mmhc 1000000"}; do
  set -- $leg
  python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>gpurun_out/dim_share.err >/dev/null
  python3 - "$leg" <<'PY'
import re, sys, collections
t = collections.defaultdict(lambda: [0.0, 0, 0])
for line in open("gpurun_out/dim_share.err"):
    m = re.match(r"pbn-group-sweep d=(\d+) (\w+) pools=(\d+) units=(\d+) pairs=(\d+) ms=([\d.]+)", line)
    if m:
        k = (int(m.group(1)), m.group(2)); t[k][0] += float(m.group(6)); t[k][1] += int(m.group(4)); t[k][2] += int(m.group(5))
tot = sum(v[0] for v in t.values())
print(f"== {sys.argv[1]}: grouped sweeps {tot / 1e3:.2f} s")
for k in sorted(t):
    ms, units, pairs = t[k]
    print(f"   d={k[0]} {k[1]}: {ms / 1e3:.2f} s ({ms / max(tot, 1e-9):.1%}), {units} units, {pairs / 1e12:.2f}e12 pairs offered -> {pairs / (ms * 1e-3) / 1e12:.1f}e12 offered pairs/s")
PY
done
