# kernel times of one KMutualInformation evaluation + a few permuted samples at N = 1e6 (tools/kmi_scale.py), rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kmi_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/kmi_scale.py > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-200
tail -3 $OUT/run.log
find $OUT -name "*kernel_trace.csv" -delete
