# training tiles per split of the grouped sweeps (PBN_GROUP_SPLIT_TILES, default 512)   bash tools/split_tiles_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for st in 512 1024 2048 4096 16384; do
echo "== PBN_GROUP_SPLIT_TILES=$st"; export PBN_GROUP_SPLIT_TILES=$st; hc cv64 1; hc c3 1; hc c5mmhc 1000000
done
