# pruned fp64 sum-only sweeps with / without the fp32 tail path of far tiles (FARP)   bash tools/far_probe.sh  (build/variants/libpbn_nofar.so = the library before it)
cd $GRAFT_REPO_ROOT
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
all() { hc cv64 1; hc c3 1; python3 tools/prune_visits.py 2>/dev/null | cut -c1-100; }
echo "== far path (default)"; all
echo "== PBN_FAR_SPAN=0"; PBN_FAR_SPAN=0 hc cv64 1; PBN_FAR_SPAN=0 hc c3 1
echo "== PBN_FAR_SPAN=21"; PBN_FAR_SPAN=21 hc cv64 1; PBN_FAR_SPAN=21 hc c3 1
cp build/variants/libpbn_nofar.so pybnesian_amd/libpbn_hip.so
echo "== library before"; all
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
