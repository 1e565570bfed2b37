# round 5: kernel shares of the C3 / cv64 first iterations with the moment pass (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for leg in "cv64 1" "c3 1"; do
  set -- $leg
  OUT=$R/gpurun_out/r5_stats_$1
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 > $OUT/run.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $leg"; head -8 $f | cut -c1-170
  find $OUT -name "*kernel_trace.csv" -delete
done
