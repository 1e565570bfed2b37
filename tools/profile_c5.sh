# rocprofv3 kernel stats of config C5's MMPC-restricted hill-climb (first N iterations): which sweep variants the time goes to
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$(dirname ${2:-c5_prof})
ITERS=${1:-25}
OUT=${2:-c5_prof}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT -- python3 $R/tools/c5_breakdown.py $ITERS > $R/gpurun_out/$OUT.log 2>&1
cd $R
f=$(find gpurun_out/$OUT -name "*kernel_stats.csv" | head -1)
head -25 "$f"
tail -8 gpurun_out/$OUT.log
