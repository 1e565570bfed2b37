# the default bench line's hill-climb legs under different arena budgets   bash tools/arena_bench_probe.sh
cd $GRAFT_REPO_ROOT
for mb in 4096 8192 16384; do
echo "== PBN_GROUP_ARENA_MB=$mb"
PBN_GROUP_ARENA_MB=$mb python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print(' '.join('%s %.3f' % (k[10:], d[k]['estimate_s']) for k in d if k.startswith('secondary_') and 'estimate_s' in d[k]), 'c5 8-rank', d['secondary_c5']['eight_rank_estimate'].get('per_rank_s'), d['secondary_c5']['eight_rank_estimate'].get('slowest_over_mean_share'))"
done
