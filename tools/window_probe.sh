cd $GRAFT_REPO_ROOT
run() { hc=$1; it=$2; shift; shift; env "$@" python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $hc --hc-max-iters $it --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('%.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
for w in 32 64 128 256; do echo -n "cv64 PBN_GROUP_WINDOW=$w: "; run cv64 1 PBN_GROUP_WINDOW=$w; done
for w in 32 128; do echo -n "c3 PBN_GROUP_WINDOW=$w: "; run c3 1 PBN_GROUP_WINDOW=$w; done
for w in 32 128; do echo -n "c5 PBN_GROUP_WINDOW=$w: "; run c5mmhc 1000000 PBN_GROUP_WINDOW=$w; done
