# fp32 headline (C2 in fp32) with variant builds of the library (build/variants/libpbn_*.so)   bash tools/f32_variants.sh fsum0 fsum8w3
cd $GRAFT_REPO_ROOT
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
f32() { python3 bench.py --dtype f32 --hc none --no-e2e --no-cpu-baseline --no-c3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 headline ms/step %.3f  frac %.4f  slogl %.6f' % (d['ms_per_step'], d['roofline']['frac'], d['config']['slogl_step0_rank_sum']))"; }
echo "== base"; f32
for v in "$@"; do cp build/variants/libpbn_$v.so pybnesian_amd/libpbn_hip.so; echo "== $v"; f32; done
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
echo "== base again"; f32
