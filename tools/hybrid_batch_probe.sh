# hybrid candidates of one pbn_score_batch call in one chain (PBN_HYBRID_BATCH=0: one chain per candidate)   bash tools/hybrid_batch_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for i in 1 2; do
echo "== default"; hc c5mmhc 1000000
echo "== PBN_HYBRID_BATCH=0"; PBN_HYBRID_BATCH=0 hc c5mmhc 1000000
for mb in 4096 8192 32768 65536; do
echo "== PBN_GROUP_ARENA_MB=$mb"; PBN_GROUP_ARENA_MB=$mb hc c5mmhc 1000000
done
echo "== PBN_GROUP_ARENA_MB=65536 PBN_GROUP_MAX_POOLS=64"; PBN_GROUP_MAX_POOLS=64 PBN_GROUP_ARENA_MB=65536 hc c5mmhc 1000000
done
