# round-3 evidence set -> gpurun_out/r3/ (copied into profiles/r3/ afterwards)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
bash tools/profile_bench.sh r3/prof > $O/profile_bench.log 2>&1
bash tools/c5_trace.sh 1 r3/cv64_trace cv64 > $O/cv64_trace.txt 2>&1
bash tools/c5_trace.sh 1000000 r3/c5_trace c5mmhc > $O/c5_trace.txt 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --hc c3 --hc-max-iters 1000000 --no-c3 > $O/bench_c3_full.json 2> $O/bench_c3_full.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --hc cv64 --no-c3 > $O/bench_cv64.json 2> $O/bench_cv64.err
(time python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_default.time
PBN_BENCH_DEVICE=0 python3 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
SCALE_DETAIL=1 python3 tools/scale_emulate.py 1,2,4,8 2>&1 | grep -v amdgpu.ids > $O/scale_emulate.txt
bash tools/gram_evidence.sh > $O/gram_paths.txt 2>&1
find $O -name "*.csv" -size +2M -delete
ls -la $O
