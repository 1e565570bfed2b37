# RING pass (far field of the grouped fp64 sweeps on the bf16 cores) under PBN_RING_NEAR: parity against the per-unit chain and timings
cd $GRAFT_REPO_ROOT
run() { hc=$1; it=$2; shift; shift; env "$@" python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $hc --hc-max-iters $it --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('%.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for r in 32 24 0; do
  echo "== PBN_RING_NEAR=$r"
  PBN_RING_NEAR=$r python3 tools/group_check.py 2>&1 | grep "n=\|oracle\|ok\|Error\|assert" | head -9
  echo -n "cv64: "; run cv64 1 PBN_RING_NEAR=$r
  echo -n "c3 (1 iteration): "; run c3 1 PBN_RING_NEAR=$r
done
