# fp32 sweeps with / without the fp32 accumulation inside blind batches (PBN_F16_FSUM), and the cost of the engine's check-after (PBN_F32_WIDEN_AT=inf)
#   bash tools/fsum_probe.sh   (build/variants/libpbn_nofsum.so = the library built with -DPBN_F16_FSUM=0)
cd $GRAFT_REPO_ROOT
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
f32() { python3 bench.py --dtype f32 --hc none --no-e2e --no-cpu-baseline --no-c3 --no-extra-legs --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 headline ms/step %.3f  frac %.4f slogl %.10g' % (d['ms_per_step'], d['roofline']['frac'], d['config']['slogl_step0_rank_sum']))"; }
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
echo "== fsum (default)"; f32; hc c5mmhc 1000000; hc c5mmhc 1000000
echo "== fsum, PBN_F32_WIDEN_AT=inf"; PBN_F32_WIDEN_AT=inf hc c5mmhc 1000000
cp build/variants/libpbn_nofsum.so pybnesian_amd/libpbn_hip.so
echo "== nofsum"; f32; hc c5mmhc 1000000; hc c5mmhc 1000000
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
