# round 4: the error budget of the sum-only sweeps - pruning margins whose dropped-mass bound matches the arithmetic's own error
# (fp64 sums: 2^f on the fp32 unit, 1.4e-7 per term -> margin 43 at 1e6 rows = 1.1e-7 of a sum; fp32: 36 = 1.5e-5)   bash tools/margin_probe4.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for m in "52 40" "46 38" "43 36" "40 34"; do
  set -- $m
  echo "== PBN_PRUNE_MARGIN=$1 PBN_PRUNE_MARGIN_F32=$2"
  export PBN_PRUNE_MARGIN=$1 PBN_PRUNE_MARGIN_F32=$2
  hc cv64 1; hc c3 1; hc c5mmhc 1000000
done
