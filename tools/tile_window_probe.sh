# the grouped prepass's tile-box bound of the queries' sums (PBN_GROUP_TILE_WINDOW tiles on either side; 0 = the 64-row scan alone)   bash tools/tile_window_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for w in 0 16 64 256; do echo "== PBN_GROUP_TILE_WINDOW=$w"; export PBN_GROUP_TILE_WINDOW=$w; hc cv64 1; hc c3 1; hc c5mmhc 1000000; done
