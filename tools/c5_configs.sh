run() { echo "== $*"; env "$@" python bench.py --no-c3 --no-e2e --no-cpu-baseline --hc c5mmhc --hc-max-iters 1000000 --steps 1 --warmup 1 2>gpurun_out/c5_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['secondary']; print(d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'])"; }
run PBN_SWEEP_LOG=1
grep pbn-sweep gpurun_out/c5_err.log | sort | uniq -c | sort -k1,1nr > gpurun_out/c5_sweeps.txt; wc -l gpurun_out/c5_sweeps.txt; rm gpurun_out/c5_err.log
run PBN_PRUNE_MIN_ROWS=200000
run PBN_PRUNE_MIN_ROWS=400000
