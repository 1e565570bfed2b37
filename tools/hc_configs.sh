# One hill-climb config of bench.py under environment switches, one line each:
#   bash tools/hc_configs.sh cv64 "PBN_SWEEP_MIN_TILES=64" "PBN_SWEEP_BLOCKS_PER_CU=12 PBN_SCORE_LANES=3"
HC=$1; shift
IT=${HC_ITERS:-0}
run() { echo "== $*"; env $* python bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $HC --hc-max-iters $IT --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['secondary']; print(d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'])"; }
run PBN_NONE=1
for c in "$@"; do run $c; done
