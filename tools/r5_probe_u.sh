# round 5: the sum bound's windows and the far span after the tiles became compact (experiments build)   bash tools/r5_probe_u.sh
cd $GRAFT_REPO_ROOT
export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
hc cv64 1 > /dev/null
for cfg in "PBN_X=0" "PBN_GROUP_TILE_WINDOW=512" "PBN_GROUP_TILE_WINDOW=1024" "PBN_GROUP_TILE_WINDOW=128" "PBN_GROUP_WINDOW=16" "PBN_GROUP_WINDOW=32" "PBN_FAR_SPAN=21" "PBN_FAR_SPAN=19"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 6"
done
