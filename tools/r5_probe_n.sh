# round 5: Hilbert order at three / four key dimensions (experiments build): grouped sweeps of C3's first 12 iterations, stand-alone handles   bash tools/r5_probe_n.sh
cd $GRAFT_REPO_ROOT
export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
hc cv64 1 > /dev/null
for cfg in "PBN_GROUP_HILBERT=2" "PBN_GROUP_HILBERT=1"; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc c3 12; hc c3 12"
done
for cfg in "PBN_PRUNE_HILBERT_ND=1" "PBN_PRUNE_HILBERT_ND=0"; do
  echo "== $cfg"
  env $cfg python3 tools/prune_handles_timing.py 2>&1 | grep "float64 | d=[345]\|float32 | d=[34]" | cut -c1-400
done
