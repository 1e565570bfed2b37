# round 5, verdict item 2 (far-field evaluation): what a FREE far field could buy.  The sum-only grouped fp64 sweeps with the pruning
# margin pulled in to m and the fp32 far path off evaluate exactly the pairs a near pass would keep if everything below 2^-m of the sum
# bound went to an expansion that cost nothing - an upper bound on the gain of any far-field scheme with its hand-over at 2^-m.
#   bash tools/near_only_probe.sh  -> gpurun_out/near_only_probe.txt
cd $GRAFT_REPO_ROOT
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
echo "== default (margin 43, far path at 26)"; hc cv64 1; hc c3 1
for m in 43 36 30 26 22 18 14; do
  echo "== PBN_PRUNE_MARGIN_SUM=$m PBN_FAR_SPAN=0 (near pass alone, hand-over at 2^-$m)"
  PBN_PRUNE_MARGIN_SUM=$m PBN_FAR_SPAN=0 hc cv64 1
  PBN_PRUNE_MARGIN_SUM=$m PBN_FAR_SPAN=0 hc c3 1
done
