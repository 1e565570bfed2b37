# fp32 unpruned sweeps compiled for 3 / 4 waves per SIMD (-DPBN_F16_WAVES)   bash tools/waves_probe.sh w3 w4
cd $GRAFT_REPO_ROOT
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
f32() { python3 bench.py --dtype f32 --hc none --no-e2e --no-cpu-baseline --no-c3 --no-extra-legs --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 headline ms/step %.3f  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
echo "== base"; f32; f32
for v in "$@"; do cp build/variants/libpbn_$v.so pybnesian_amd/libpbn_hip.so; echo "== $v"; f32; f32; done
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
