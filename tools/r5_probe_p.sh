# NOTE: needs the library of commit c7bb7dc (PBN_MARGIN_CUT does not exist in the shipped code).
# round 5: the a-posteriori pruning radius (PBN_MARGIN_CUT bits inside the a-priori margin, dropped mass proved per query): time and redone terms   bash tools/r5_probe_p.sh
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweeps %s redone %s' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('launches'), d.get('redone_terms')))"; }
hc cv64 1 > /dev/null
for cfg in ${CUTS:-"PBN_MARGIN_CUT=0" "PBN_MARGIN_CUT=6" "PBN_MARGIN_CUT=10" "PBN_MARGIN_CUT=14"}; do
  echo "== $cfg"
  env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c3 12"
done
