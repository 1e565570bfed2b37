# the prepass's ROW scan (PBN_GROUP_WINDOW rows on either side; default 32) now that the tile boxes carry the sum bound   bash tools/row_window_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for w in 32 16 8 4; do echo "== PBN_GROUP_WINDOW=$w"; export PBN_GROUP_WINDOW=$w; hc cv64 1; hc c3 1; hc c5mmhc 1000000; done
