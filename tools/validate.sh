# the randomised checks against the oracle / the unpruned sweeps on the final code of a round (round 6: f16x2 fragments of fp32 tables, the
# restructured moment pass, the dealing of lone candidates); results under gpurun_out/val -> profiles/rN/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/val
python3 tools/fuzz_pruned.py > gpurun_out/val/fuzz_pruned.txt 2>&1; tail -2 gpurun_out/val/fuzz_pruned.txt
python3 tools/fuzz_grouped.py > gpurun_out/val/fuzz_grouped.txt 2>&1; tail -2 gpurun_out/val/fuzz_grouped.txt
python3 tools/fuzz_sharding.py > gpurun_out/val/fuzz_sharding.txt 2>&1; tail -2 gpurun_out/val/fuzz_sharding.txt
python3 tools/fuzz_hc.py > gpurun_out/val/fuzz_hc.txt 2>&1; tail -2 gpurun_out/val/fuzz_hc.txt
PBN_PRUNE_MIN_ROWS=256 python3 tools/fuzz_hc.py > gpurun_out/val/fuzz_hc_pruned.txt 2>&1; tail -2 gpurun_out/val/fuzz_hc_pruned.txt
python3 tools/fuzz_mmhc.py > gpurun_out/val/fuzz_mmhc.txt 2>&1; tail -2 gpurun_out/val/fuzz_mmhc.txt
python3 tools/group_check.py > gpurun_out/val/group_check.txt 2>&1; tail -3 gpurun_out/val/group_check.txt
# ... and with the tile-moment pass forced on every fp64 unit of one or two variables (the shipped rule takes it from 400 000 training rows)
PBN_MOMENT_MIN_ROWS=0 python3 tools/fuzz_grouped.py 40 7 > gpurun_out/val/fuzz_grouped_moments.txt 2>&1; tail -2 gpurun_out/val/fuzz_grouped_moments.txt
PBN_MOMENT_MIN_ROWS=0 PBN_PRUNE_MIN_ROWS=256 python3 tools/fuzz_hc.py > gpurun_out/val/fuzz_hc_moments.txt 2>&1; tail -2 gpurun_out/val/fuzz_hc_moments.txt
