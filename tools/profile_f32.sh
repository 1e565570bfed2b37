cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f32 -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --hc none > $R/gpurun_out/prof_f32.log 2>&1
cd $R
f=$(find gpurun_out/prof_f32 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/bench_f32_kernel_stats.csv
head -4 gpurun_out/bench_f32_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_f32
