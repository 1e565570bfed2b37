# round 5: training tiles per split of the grouped sweeps with the moment pass beside them (C3's first iteration)   bash tools/r5_probe_k.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
hc cv64 1 > /dev/null
for cfg in "PBN_GROUP_SPLIT_TILES=0" "PBN_GROUP_SPLIT_TILES=1024" "PBN_GROUP_SPLIT_TILES=2048" "PBN_GROUP_SPLIT_TILES=4096" "PBN_GROUP_SPLIT_TILES=16384"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc c3 1; PBN_MOMENT_PASS=0 hc c3 1"
done
