# split size following the unit size (default) against 512 everywhere   bash tools/split_auto_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for i in 1 2; do
echo "== default (512 / 1024 / 2048 by unit size)"; unset PBN_GROUP_SPLIT_TILES; hc cv64 1; hc c3 1; hc c5mmhc 1000000
echo "== PBN_GROUP_SPLIT_TILES=512"; export PBN_GROUP_SPLIT_TILES=512; hc cv64 1; hc c3 1; hc c5mmhc 1000000
done
