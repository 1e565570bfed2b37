# A/B of library builds on ONE box, hill-climb legs only: bash tools/ab_hc.sh <outdir under gpurun_out> <name>=<lib path or "main"> ...
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-ab}; shift
mkdir -p $OUT
for spec in "$@"; do
  n=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = main ]; then unset PBN_LIB; else export PBN_LIB=$PWD/$lib; fi
  for leg in cv64 c3; do
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --hc $leg --hc-max-iters 1 2>> $OUT/$n.err | tail -1 > $OUT/$n.$leg.json; cp bench_full.json $OUT/$n.$leg.full.json
  done
  python - $OUT $n <<'P'
import json, sys
out, n = sys.argv[1:3]
r = []
for leg in ("cv64", "c3"):
    d = json.load(open(f"{out}/{n}.{leg}.full.json"))
    sec = d.get("secondary") or {}
    r.append(f"{leg} {sec.get('estimate_s')}")
print(f"{n:8s} " + " | ".join(r))
P
done
