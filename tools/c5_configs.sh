# C5 hill-climb under environment switches, one line each: bash tools/c5_configs.sh "PBN_PRUNE_MAX_TILES=512" "PBN_SWEEP_BLOCKS_PER_CU=48 PBN_SCORE_LANES=3"
run() { echo "== $*"; env $* python bench.py --no-c3 --no-e2e --no-cpu-baseline --hc c5mmhc --hc-max-iters 1000000 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['secondary']; print(d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'])"; }
run PBN_NONE=1
for c in "$@"; do run $c; done
