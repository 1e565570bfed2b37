# round 5: (a) the finer pool Morton keys + the near-zero guard on the search legs, (b) the KMI timings   bash tools/r5_probe_a.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
hc cv64 1; hc c3 1; hc c5mmhc 1000000
python3 tools/kmi_scale.py
