# round 5 A/B on the experiments build: Hilbert order of the pool keys at two key dimensions (GROUP_HILBERT=1) against the Z-order (0)
cd $GRAFT_REPO_ROOT
export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
for cfg in "PBN_GROUP_HILBERT=1" "PBN_GROUP_HILBERT=0"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c5mmhc 1000000"
done
