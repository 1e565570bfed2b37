# cv64 (64-node CV-likelihood hill-climb, initial delta cache + 1 iteration) under the knobs of the grouped evaluation
#   bash tools/cv64_knobs.sh  -> gpurun_out/cv64_knobs.txt
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --hc cv64 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('%.3f s  cells %d' % (d['estimate_s'], d['cells_scored']))"; }
for st in 128 256 512 1024 2048; do echo -n "PBN_GROUP_SPLIT_TILES=$st: "; run PBN_GROUP_SPLIT_TILES=$st; done
echo -n "PBN_GROUP_SUM_BOUND=0: "; run PBN_GROUP_SUM_BOUND=0
echo -n "PBN_GROUP_MAX_POOLS=32: "; run PBN_GROUP_MAX_POOLS=32
echo -n "PBN_GROUP_MAX_POOLS=8: "; run PBN_GROUP_MAX_POOLS=8
echo -n "PBN_SCORE_GROUPED=0: "; run PBN_SCORE_GROUPED=0
