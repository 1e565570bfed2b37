# round 5: the pruned fp64 sweeps compiled for 2 instead of 3 waves per SIMD (-DPBN_F64_PRUNE_WAVES=2: 256 VGPRs, nothing in scratch)   bash tools/r5_probe_q.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
hc cv64 1 > /dev/null
for lib in ${LIBS:-libpbn_hip.so libpbn_hip_w2.so}; do
  echo "== $lib"
  PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c3 24"
  PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib python3 tools/prune_handles_timing.py 2>&1 | grep "float64" | sed 's/(slogl[^)]*)//g' | cut -c1-250
done
