cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 5 --warmup 2 2>gpurun_out/bench_full.err | tail -1 > gpurun_out/bench_n1.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --hc none > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --hc none > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --hc none > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --hc none > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq -name "*.csv" | head -20
