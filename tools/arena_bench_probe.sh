# the default bench line's hill-climb legs (one process, the legs one after the other) under different arena budgets / with the hybrid
# candidates one by one   bash tools/arena_bench_probe.sh
cd $GRAFT_REPO_ROOT
run() {
echo "== $*"
env "$@" python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print(' '.join('%s %.3f' % (k[10:], d[k]['estimate_s']) for k in d if k.startswith('secondary_') and 'estimate_s' in d[k]), 'c5 8-rank', d['secondary_c5']['eight_rank_estimate'].get('per_rank_s'), d['secondary_c5']['eight_rank_estimate'].get('slowest_over_mean_share'))"
}
for i in 1 2; do
run PBN_HYBRID_BATCH=0 PBN_GROUP_ARENA_MB=4096
run PBN_GROUP_ARENA_MB=4096
run PBN_GROUP_ARENA_MB=8192
done
