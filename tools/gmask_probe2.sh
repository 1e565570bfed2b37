# per-group visit masks (PBN_PRUNE_GROUP_MASKS) on the 1e6 x 1e5 handles (fp64 / fp32, d = 1..5) and on config 5's hill-climb
cd $GRAFT_REPO_ROOT
for m in 1 0; do echo "== PBN_PRUNE_GROUP_MASKS=$m"; PBN_PRUNE_GROUP_MASKS=$m python3 tools/prune_handles_timing.py 2>&1 | sed 's/prune=0: fit [0-9.]* ms slogl [0-9.]* ms logl [0-9.]* ms (slogl [-0-9.]*) | //g' | cut -c1-260; done
run() { hc=$1; shift; env "$@" python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $hc --hc-max-iters 1000000 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('%.3f s  cells %d iterations %d' % (d['estimate_s'], d['cells_scored'], d['iterations']))"; }
for m in 1 0; do echo -n "c5 PBN_PRUNE_GROUP_MASKS=$m: "; run c5mmhc PBN_PRUNE_GROUP_MASKS=$m; done
