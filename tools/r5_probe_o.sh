# round 5: what the fp32 tail of far tiles (FARP, FOLD shapes: d = 1..3) is worth on C3's first 12 iterations   bash tools/r5_probe_o.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
hc cv64 1 > /dev/null
for cfg in "PBN_FAR_SPAN=17" "PBN_FAR_SPAN=0" "PBN_FAR_SPAN=21"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc c3 12"
done
