# round 5: the pruned fp32 sweeps compiled for 5 instead of 4 waves per SIMD (-DPBN_F16_PRUNE_WAVES=5), C5's hill-climb   bash tools/r5_probe_r.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
hc cv64 1 > /dev/null
for lib in libpbn_hip.so libpbn_hip_b5.so libpbn_hip.so libpbn_hip_b5.so; do
  echo "== $lib"
  PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc c5mmhc 1000000"
done
