# cost of the fp32 engine's check-after (max-norm slots, PBN_F32_CHECK=0 switches it off)   bash tools/check_after_probe.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for i in 1 2; do
echo "== default"; hc c5mmhc 1000000
echo "== PBN_F32_CHECK=0"; PBN_F32_CHECK=0 hc c5mmhc 1000000
done
