# round-3 timing set: grouped / per-unit checks, cv64, first C3 iteration, config 5, the 1e6 x 1e5 handles
cd $GRAFT_REPO_ROOT
python3 tools/group_check.py 2>&1 | tail -11
run() { hc=$1; it=$2; shift; shift; env "$@" python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $hc --hc-max-iters $it --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('%.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
echo -n "cv64: "; run cv64 1 A=1
echo -n "c3 (1 iteration): "; run c3 1 A=1
echo -n "c5mmhc: "; run c5mmhc 1000000 A=1
python3 tools/prune_handles_timing.py 2>&1 | sed 's/prune=0: fit [0-9.]* ms slogl [0-9.]* ms logl [0-9.]* ms (slogl [-0-9.]*) | //g' | cut -c1-260
