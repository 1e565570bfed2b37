# round 5: the tile-moment pass of the grouped fp64 sum-only sweeps (d <= 2) on / off   bash tools/r5_probe_g.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
hc cv64 1 > /dev/null
for cfg in "PBN_MOMENT_PASS=1" "PBN_MOMENT_PASS=0"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c3 6"
done
