# kernel stats of the C4 Gram pass (2M x 64 fp64, tools/gram_bench.py): gram kernel duration under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-gram_prof}
rm -rf $R/gpurun_out/$OUT
mkdir -p $R/gpurun_out/$OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT -- python3 $R/tools/gram_bench.py > $R/gpurun_out/$OUT.log 2>&1
cd $R
f=$(find gpurun_out/$OUT -name "*kernel_stats.csv" | head -1)
grep -i "gram" $f | cut -c1-200
tail -1 gpurun_out/$OUT.log
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
