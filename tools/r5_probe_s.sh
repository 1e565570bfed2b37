# round 5: query groups per wave of the pruned fp32 sweeps (-DPBN_F16_QG_PRUNE=2 against 4): C5's hill-climb and the fp32 handles   bash tools/r5_probe_s.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
hc cv64 1 > /dev/null
for lib in ${LIBS:-libpbn_hip.so libpbn_hip_bq2.so libpbn_hip.so libpbn_hip_bq2.so}; do
  echo "== $lib"
  PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc c5mmhc 1000000"
  PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib python3 tools/prune_handles_timing.py 2>&1 | grep "float32" | sed 's/(slogl[^)]*)//g; s/KDE prune=0[^|]*|//g' | cut -c1-150
done
