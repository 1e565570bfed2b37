# round 5: the unpruned fp32 headline sweep (kde_sweep_f16_kernel<2, false, 4, false>) compiled for 2 / 3 / 4 waves per SIMD (-DPBN_F16_WAVES)
cd $GRAFT_REPO_ROOT
f32() { python3 bench.py --dtype f32 --no-cpu-baseline --hc none --no-extra-legs --no-e2e --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 headline %.3f ms per step, sweep %.3f ms, frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
f32 > /dev/null
for w in "" _w3 _w4; do echo "== libpbn_hip$w.so"; PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip$w.so f32; PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip$w.so f32; done
