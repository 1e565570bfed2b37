# From how many training rows on does the moment pass pay (PBN_MOMENT_MIN_ROWS)?  cv64 (64 nodes, 10 folds, first iteration) at several table
# sizes, pass forced / off.  (round 5: tools/r5_probes.sh m; round 6: re-run after the pass got 11 % faster)
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc cv64 --hc-rows $1 --hc-max-iters 1 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=list(json.loads(sys.stdin.read())['legs'].values())[0]; print('cv64 rows $1: %.3f s  cells %d' % (d['estimate_s'], d['cells_scored']))"; }
hc 100000 > /dev/null
for rows in ${ROWS:-100000 200000 300000 400000}; do
  for cfg in "PBN_MOMENT_MIN_ROWS=0" "PBN_MOMENT_PASS=0"; do
    echo -n "$cfg  "; env $cfg bash -c "$(declare -f hc); hc $rows"
  done
done
