# what a tighter near-field radius would buy: cv64 and the first C3 iteration under PBN_PRUNE_MARGIN (exponent distance below the
# queries' bound beyond which a tile is skipped; default 52)
cd $GRAFT_REPO_ROOT
run() { hc=$1; shift; env "$@" python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $hc --hc-max-iters 1 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('%.3f s  cells %d' % (d['estimate_s'], d['cells_scored']))"; }
for m in 52 44 36 32 28 24; do echo -n "cv64 PBN_PRUNE_MARGIN=$m: "; run cv64 PBN_PRUNE_MARGIN=$m; done
for m in 52 32; do echo -n "c3 PBN_PRUNE_MARGIN=$m: "; run c3 PBN_PRUNE_MARGIN=$m; done
