# round 5 A/B on the experiments build: Morton key cells at 3 / 4 key dimensions (grouped: cells per sigma; stand-alone: cell edge in whitened units)
cd $GRAFT_REPO_ROOT
export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
for cfg in "PBN_GROUP_KEY_SCALE3=16 PBN_GROUP_KEY_SCALE4=8" "PBN_GROUP_KEY_SCALE3=32 PBN_GROUP_KEY_SCALE4=8" "PBN_GROUP_KEY_SCALE3=64 PBN_GROUP_KEY_SCALE4=8" "PBN_GROUP_KEY_SCALE3=32 PBN_GROUP_KEY_SCALE4=16"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc cv64 6; hc c3 4; hc c5mmhc 1000000"
done
for cfg in "PBN_KEY_CELL4=2.0 PBN_KEY_CELL5=2.0" "PBN_KEY_CELL4=0.5 PBN_KEY_CELL5=1.0" "PBN_KEY_CELL4=0.25 PBN_KEY_CELL5=0.5"; do
  echo "== $cfg"
  env $cfg python3 tools/prune_visits.py | tail -2
done
