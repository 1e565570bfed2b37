# fp32 headline + C5 + cv64 with variant builds of the library (build/variants/libpbn_*.so)   bash tools/variant_probe.sh noslp ...
cd $GRAFT_REPO_ROOT
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
f32() { python3 bench.py --dtype f32 --hc none --no-e2e --no-cpu-baseline --no-c3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 headline ms/step %.3f  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
f64() { python3 bench.py --hc none --no-e2e --no-cpu-baseline --no-c3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 headline ms/step %.3f  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
all() { f32; f64; hc c5mmhc 1000000; hc cv64 1; }
echo "== base"; all
for v in "$@"; do cp build/variants/libpbn_$v.so pybnesian_amd/libpbn_hip.so; echo "== $v"; all; done
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
