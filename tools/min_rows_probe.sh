# smaller slices on the pruned (grouped) path: C5 under PBN_PRUNE_MIN_ROWS   bash tools/min_rows_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for m in 32768 16384 8192 4096; do echo "== PBN_PRUNE_MIN_ROWS=$m"; PBN_PRUNE_MIN_ROWS=$m hc c5mmhc 1000000; done
