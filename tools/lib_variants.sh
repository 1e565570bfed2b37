# C5 hill-climb with variant builds of the library (build/variants/libpbn_*.so, made by hand with -D switches): copies each over
# the in-tree library ON THE GPU BOX's scratch copy, runs, restores.  bash tools/lib_variants.sh w3 w4
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
run() { python bench.py --no-c3 --no-e2e --no-cpu-baseline --hc c5mmhc --hc-max-iters 1000000 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['secondary']; print(d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'])"; }
f32() { [ -n "$PV_F32" ] && python bench.py --dtype f32 --hc none --no-e2e --no-cpu-baseline --no-c3 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 headline ms/step', d['ms_per_step'])"; }
sl() { [ -n "$PV_SLICES" ] && python tools/slice_visits.py 2>/dev/null | grep -E " (720000|240000) x" | cut -c1-130; }
echo "== base"; run; sl; f32
for v in "$@"; do
  cp build/variants/libpbn_$v.so pybnesian_amd/libpbn_hip.so
  echo "== $v"; run; sl; f32
done
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
echo "== base again"; run
