# A/B of library builds on ONE box: bash tools/ab_libs.sh <outdir under gpurun_out> <name>=<lib path or "main"> ...
# per build: the default bench command (short) -> headline ms, C3 first iteration, C5, cv_weak, fp32 headline; cv64 first iterations
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-ab}; shift
mkdir -p $OUT
for spec in "$@"; do
  n=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = main ]; then unset PBN_LIB; else export PBN_LIB=$PWD/$lib; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/$n.line.json 2> $OUT/$n.err; cp bench_full.json $OUT/$n.full.json
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --hc cv64 --hc-max-iters 1 > $OUT/$n.cv64.json 2>> $OUT/$n.err
  python - $OUT $n <<'P'
import json, sys
out, n = sys.argv[1:3]
d = json.load(open(f"{out}/{n}.line.json")); L = d["legs"]
cv = json.load(open(f"{out}/{n}.cv64.json"))
sec = cv.get("secondary") or cv.get("legs", {}).get("cv64") or {}
print(f"{n:8s} C2 {d['ms_per_step']:.2f} ms frac {d['roofline']['frac']:.3f} | c3 {L['c3']['estimate_s']:.3f} s (moment {L['c3']['roofline'].get('moment_s')}) | c5 {L['c5']['estimate_s']:.3f} | cv_weak {L['cv_weak']['estimate_s']:.3f} | f32 {L['f32']['ms_per_step']:.2f} ms | cv64 {sec.get('estimate_s')}")
P
done
