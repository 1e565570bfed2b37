# fp64 pruned handles (tools/prune_visits.py) and the bounded C3 leg with variant builds of the library, see lib_variants.sh
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
run() { python tools/prune_visits.py 2>/dev/null | cut -c1-60; python bench.py --no-e2e --no-cpu-baseline --hc c3 --hc-max-iters 1 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['secondary']; print('c3 bounded', d['estimate_s'], d['cells_scored'])"; }
echo "== base"; run
for v in "$@"; do
  cp build/variants/libpbn_$v.so pybnesian_amd/libpbn_hip.so
  echo "== $v"; run
done
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
