# round 5: the first level of the grouped sweeps' walk (boxes of the 64-tile batches) on / off, experiments build   bash tools/r5_probe_j.sh
cd $GRAFT_REPO_ROOT
export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
hc cv64 1 > /dev/null
for cfg in "PBN_GROUP_BATCH_BOXES=1" "PBN_GROUP_BATCH_BOXES=0"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc c5mmhc 1000000; hc c5mmhc 1000000; hc c3 1"
done
