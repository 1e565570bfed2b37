# arena budget of the grouped chains (PBN_GROUP_ARENA_MB) on the plain-CKDE hill-climbs   bash tools/arena_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for i in 1 2; do
for mb in 4096 16384; do
echo "== PBN_GROUP_ARENA_MB=$mb"; PBN_GROUP_ARENA_MB=$mb hc cv64 1; PBN_GROUP_ARENA_MB=$mb hc c3 1; PBN_GROUP_ARENA_MB=$mb hc c5mmhc 1000000
done
done
