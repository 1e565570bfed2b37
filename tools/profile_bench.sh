# rocprofv3 evidence for bench.py's headline kernel: kernel stats + three separate PMC passes (never combined with a trace
# domain: gpurun refuses that).  bash tools/profile_bench.sh [outdir under gpurun_out]   then
#   python tools/pmc_aggregate.py profiles/rN/pmc_per_dispatch.json gpurun_out/<out>/pmc_fetch gpurun_out/<out>/pmc_write gpurun_out/<out>/pmc_sq
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --hc none --no-c3 --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -delete
python tools/pmc_aggregate.py $OUT/pmc_per_dispatch.json $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/bench_kernel_stats.csv; head -6 $f | cut -c1-200
