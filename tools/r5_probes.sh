# The one-shot A/B probes of round 5 (profiles/r5/*_probe.txt, moment_pass.txt, waves_probe.txt ...), one function per probe; most need the
# experiments build (make -C pybnesian_amd/csrc EXPERIMENTS=1 OUT=../libpbn_hip_exp.so BUILD=../../build/csrc_exp; PBN_LIB=...).
#   bash tools/r5_probes.sh <letter>      (a ... u; `bash tools/r5_probes.sh list` prints what each one measured)

probe_a() {
  # round 5: (a) the finer pool Morton keys + the near-zero guard on the search legs, (b) the KMI timings   bash tools/r5_probes.sh a
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1; hc c3 1; hc c5mmhc 1000000
  python3 tools/kmi_scale.py
}

probe_b() {
  # round 5 A/B on the experiments build (make EXPERIMENTS=1 OUT=../libpbn_hip_exp.so BUILD=../../build/csrc_exp): fine pool Morton keys and
  # the near-zero guard, one at a time   bash tools/r5_probes.sh b
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  for cfg in "PBN_GROUP_FINE_KEYS=1 PBN_NEAR_ZERO_LOGL=0.66" "PBN_GROUP_FINE_KEYS=0 PBN_NEAR_ZERO_LOGL=0.66" "PBN_GROUP_FINE_KEYS=1 PBN_NEAR_ZERO_LOGL=0" "PBN_GROUP_FINE_KEYS=0 PBN_NEAR_ZERO_LOGL=0"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc cv64 1; hc c3 1; hc c5mmhc 1000000"
  done
}

probe_c() {
  # round 5 A/B on the experiments build: tile-box sum bounds per query (GROUP_QUERY_BOUNDS=1) against per 16-query group (0)   bash tools/r5_probes.sh c
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  for cfg in "PBN_GROUP_QUERY_BOUNDS=1" "PBN_GROUP_QUERY_BOUNDS=0"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc cv64 1; hc c3 1; hc c5mmhc 1000000"
  done
}

probe_d() {
  # round 5 A/B on the experiments build: Morton key cells at 3 / 4 key dimensions (grouped: cells per sigma; stand-alone: cell edge in whitened units)
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  for cfg in "PBN_GROUP_KEY_SCALE3=16 PBN_GROUP_KEY_SCALE4=8" "PBN_GROUP_KEY_SCALE3=32 PBN_GROUP_KEY_SCALE4=8" "PBN_GROUP_KEY_SCALE3=64 PBN_GROUP_KEY_SCALE4=8" "PBN_GROUP_KEY_SCALE3=32 PBN_GROUP_KEY_SCALE4=16"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 6; hc c3 4; hc c5mmhc 1000000"
  done
  for cfg in "PBN_KEY_CELL4=2.0 PBN_KEY_CELL5=2.0" "PBN_KEY_CELL4=0.5 PBN_KEY_CELL5=1.0" "PBN_KEY_CELL4=0.25 PBN_KEY_CELL5=0.5"; do
    echo "== $cfg"
    env $cfg python3 tools/prune_visits.py | tail -2
  done
}

probe_e() {
  # round 5 A/B on the experiments build: Hilbert order of the pool keys at two key dimensions (GROUP_HILBERT=1) against the Z-order (0)
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  for cfg in "PBN_GROUP_HILBERT=1" "PBN_GROUP_HILBERT=0"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c5mmhc 1000000"
  done
}

probe_f() {
  # round 5: the unpruned fp32 headline sweep (kde_sweep_f16_kernel<2, false, 4, false>) compiled for 2 / 3 / 4 waves per SIMD (-DPBN_F16_WAVES)
  cd $GRAFT_REPO_ROOT
  f32() { python3 bench.py --dtype f32 --no-cpu-baseline --hc none --no-extra-legs --no-e2e --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 headline %.3f ms per step, sweep %.3f ms, frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
  f32 > /dev/null
  for w in "" _w3 _w4; do echo "== libpbn_hip$w.so"; PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip$w.so f32; PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip$w.so f32; done
}

probe_g() {
  # round 5: the tile-moment pass of the grouped fp64 sum-only sweeps (d <= 2) on / off   bash tools/r5_probes.sh g
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_MOMENT_PASS=1" "PBN_MOMENT_PASS=0"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c3 6"
  done
}

probe_h() {
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
}

probe_i() {
  # round 5: variants of the moment kernel (separately built libraries) on the cv64 / C3 first iterations   bash tools/r5_probes.sh i
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for lib in ${LIBS:-libpbn_hip.so libpbn_hip_w3.so}; do
    echo "== $lib"
    PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc cv64 1; hc c3 1"
  done
}

probe_j() {
  # round 5: the first level of the grouped sweeps' walk (boxes of the 64-tile batches) on / off, experiments build   bash tools/r5_probes.sh j
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_GROUP_BATCH_BOXES=1" "PBN_GROUP_BATCH_BOXES=0"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc c5mmhc 1000000; hc c5mmhc 1000000; hc c3 1"
  done
}

probe_k() {
  # round 5: training tiles per split of the grouped sweeps with the moment pass beside them (C3's first iteration)   bash tools/r5_probes.sh k
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_GROUP_SPLIT_TILES=0" "PBN_GROUP_SPLIT_TILES=1024" "PBN_GROUP_SPLIT_TILES=2048" "PBN_GROUP_SPLIT_TILES=4096" "PBN_GROUP_SPLIT_TILES=16384"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc c3 1; PBN_MOMENT_PASS=0 hc c3 1"
  done
}

probe_l() {
  # round 5: training tiles per split of the grouped fp64 sweeps after the two-level walk (cv64's first iteration, C3 six iterations)   bash tools/r5_probes.sh l
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_GROUP_SPLIT_TILES=0" "PBN_GROUP_SPLIT_TILES=256" "PBN_GROUP_SPLIT_TILES=1024" "PBN_GROUP_SPLIT_TILES=2048" "PBN_GROUP_SPLIT_TILES=4096"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc cv64 1; hc c3 6"
  done
}

probe_m() {
  # round 5: from how many training rows on does the moment pass pay?  cv64 (64 nodes, 10 folds, first iteration) at several table sizes, pass forced / off
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc cv64 --hc-rows $1 --hc-max-iters 1 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('cv64 rows $1: %.3f s  cells %d' % (d['estimate_s'], d['cells_scored']))"; }
  hc 100000 > /dev/null
  for rows in 100000 150000 200000 300000; do
    for cfg in "PBN_MOMENT_MIN_ROWS=0" "PBN_MOMENT_PASS=0"; do
      echo -n "$cfg  "; env $cfg bash -c "$(declare -f hc); hc $rows"
    done
  done
}

probe_n() {
  # round 5: Hilbert order at three / four key dimensions (experiments build): grouped sweeps of C3's first 12 iterations, stand-alone handles   bash tools/r5_probes.sh n
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_GROUP_HILBERT=2" "PBN_GROUP_HILBERT=1"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc c3 12; hc c3 12"
  done
  for cfg in "PBN_PRUNE_HILBERT_ND=1" "PBN_PRUNE_HILBERT_ND=0"; do
    echo "== $cfg"
    env $cfg python3 tools/prune_handles_timing.py 2>&1 | grep "float64 | d=[345]\|float32 | d=[34]" | cut -c1-400
  done
}

probe_o() {
  # round 5: what the fp32 tail of far tiles (FARP, FOLD shapes: d = 1..3) is worth on C3's first 12 iterations   bash tools/r5_probes.sh o
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweep share %.3f' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('share_of_estimate_s', -1)))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_FAR_SPAN=17" "PBN_FAR_SPAN=0" "PBN_FAR_SPAN=21"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc c3 12"
  done
}

probe_p() {
  # NOTE: needs the library of commit c7bb7dc (PBN_MARGIN_CUT does not exist in the shipped code).
  # round 5: the a-posteriori pruning radius (PBN_MARGIN_CUT bits inside the a-priori margin, dropped mass proved per query): time and redone terms   bash tools/r5_probes.sh p
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d  sweeps %s redone %s' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found'], d.get('roofline', {}).get('launches'), d.get('redone_terms')))"; }
  hc cv64 1 > /dev/null
  for cfg in ${CUTS:-"PBN_MARGIN_CUT=0" "PBN_MARGIN_CUT=6" "PBN_MARGIN_CUT=10" "PBN_MARGIN_CUT=14"}; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c3 12"
  done
}

probe_q() {
  # round 5: the pruned fp64 sweeps compiled for 2 instead of 3 waves per SIMD (-DPBN_F64_PRUNE_WAVES=2: 256 VGPRs, nothing in scratch)   bash tools/r5_probes.sh q
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
  hc cv64 1 > /dev/null
  for lib in ${LIBS:-libpbn_hip.so libpbn_hip_w2.so}; do
    echo "== $lib"
    PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc cv64 1; hc c3 1; hc c3 24"
    PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib python3 tools/prune_handles_timing.py 2>&1 | grep "float64" | sed 's/(slogl[^)]*)//g' | cut -c1-250
  done
}

probe_r() {
  # round 5: the pruned fp32 sweeps compiled for 5 instead of 4 waves per SIMD (-DPBN_F16_PRUNE_WAVES=5), C5's hill-climb   bash tools/r5_probes.sh r
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
  hc cv64 1 > /dev/null
  for lib in libpbn_hip.so libpbn_hip_b5.so libpbn_hip.so libpbn_hip_b5.so; do
    echo "== $lib"
    PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc c5mmhc 1000000"
  done
}

probe_s() {
  # round 5: query groups per wave of the pruned fp32 sweeps (-DPBN_F16_QG_PRUNE=2 against 4): C5's hill-climb and the fp32 handles   bash tools/r5_probes.sh s
  cd $GRAFT_REPO_ROOT
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
  hc cv64 1 > /dev/null
  for lib in ${LIBS:-libpbn_hip.so libpbn_hip_bq2.so libpbn_hip.so libpbn_hip_bq2.so}; do
    echo "== $lib"
    PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib bash -c "$(declare -f hc); hc c5mmhc 1000000"
    PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/$lib python3 tools/prune_handles_timing.py 2>&1 | grep "float32" | sed 's/(slogl[^)]*)//g; s/KDE prune=0[^|]*|//g' | cut -c1-150
  done
}

probe_u() {
  # round 5: the sum bound's windows and the far span after the tiles became compact (experiments build)   bash tools/r5_probes.sh u
  cd $GRAFT_REPO_ROOT
  export PBN_LIB=$GRAFT_REPO_ROOT/pybnesian_amd/libpbn_hip_exp.so
  hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
  hc cv64 1 > /dev/null
  for cfg in "PBN_X=0" "PBN_GROUP_TILE_WINDOW=512" "PBN_GROUP_TILE_WINDOW=1024" "PBN_GROUP_TILE_WINDOW=128" "PBN_GROUP_WINDOW=16" "PBN_GROUP_WINDOW=32" "PBN_FAR_SPAN=21" "PBN_FAR_SPAN=19"; do
    echo "== $cfg"
    env $cfg bash -c "$(declare -f hc); hc cv64 1; hc c3 6"
  done
}

case "$1" in
  list) cat <<'EOT'
a: round 5: (a) the finer pool Morton keys + the near-zero guard on the search legs, (b) the KMI timings   bash tools/r5_probes.sh a
b: round 5 A/B on the experiments build (make EXPERIMENTS=1 OUT=../libpbn_hip_exp.so BUILD=../../build/csrc_exp): fine pool Morton keys and
c: round 5 A/B on the experiments build: tile-box sum bounds per query (GROUP_QUERY_BOUNDS=1) against per 16-query group (0)   bash tools/r5_probes.sh c
d: round 5 A/B on the experiments build: Morton key cells at 3 / 4 key dimensions (grouped: cells per sigma; stand-alone: cell edge in whitened units)
e: round 5 A/B on the experiments build: Hilbert order of the pool keys at two key dimensions (GROUP_HILBERT=1) against the Z-order (0)
f: round 5: the unpruned fp32 headline sweep (kde_sweep_f16_kernel<2, false, 4, false>) compiled for 2 / 3 / 4 waves per SIMD (-DPBN_F16_WAVES)
g: round 5: the tile-moment pass of the grouped fp64 sum-only sweeps (d <= 2) on / off   bash tools/r5_probes.sh g
h: round 5: kernel shares of the C3 / cv64 first iterations with the moment pass (rocprofv3 kernel stats)
i: round 5: variants of the moment kernel (separately built libraries) on the cv64 / C3 first iterations   bash tools/r5_probes.sh i
j: round 5: the first level of the grouped sweeps' walk (boxes of the 64-tile batches) on / off, experiments build   bash tools/r5_probes.sh j
k: round 5: training tiles per split of the grouped sweeps with the moment pass beside them (C3's first iteration)   bash tools/r5_probes.sh k
l: round 5: training tiles per split of the grouped fp64 sweeps after the two-level walk (cv64's first iteration, C3 six iterations)   bash tools/r5_probes.sh l
m: round 5: from how many training rows on does the moment pass pay?  cv64 (64 nodes, 10 folds, first iteration) at several table sizes, pass forced / off
n: round 5: Hilbert order at three / four key dimensions (experiments build): grouped sweeps of C3's first 12 iterations, stand-alone handles   bash tools/r5_probes.sh n
o: round 5: what the fp32 tail of far tiles (FARP, FOLD shapes: d = 1..3) is worth on C3's first 12 iterations   bash tools/r5_probes.sh o
p: NOTE: needs the library of commit c7bb7dc (PBN_MARGIN_CUT does not exist in the shipped code).
q: round 5: the pruned fp64 sweeps compiled for 2 instead of 3 waves per SIMD (-DPBN_F64_PRUNE_WAVES=2: 256 VGPRs, nothing in scratch)   bash tools/r5_probes.sh q
r: round 5: the pruned fp32 sweeps compiled for 5 instead of 4 waves per SIMD (-DPBN_F16_PRUNE_WAVES=5), C5's hill-climb   bash tools/r5_probes.sh r
s: round 5: query groups per wave of the pruned fp32 sweeps (-DPBN_F16_QG_PRUNE=2 against 4): C5's hill-climb and the fp32 handles   bash tools/r5_probes.sh s
u: round 5: the sum bound's windows and the far span after the tiles became compact (experiments build)   bash tools/r5_probes.sh u
EOT
  ;;
  a) probe_a "${@:2}" ;;
  b) probe_b "${@:2}" ;;
  c) probe_c "${@:2}" ;;
  d) probe_d "${@:2}" ;;
  e) probe_e "${@:2}" ;;
  f) probe_f "${@:2}" ;;
  g) probe_g "${@:2}" ;;
  h) probe_h "${@:2}" ;;
  i) probe_i "${@:2}" ;;
  j) probe_j "${@:2}" ;;
  k) probe_k "${@:2}" ;;
  l) probe_l "${@:2}" ;;
  m) probe_m "${@:2}" ;;
  n) probe_n "${@:2}" ;;
  o) probe_o "${@:2}" ;;
  p) probe_p "${@:2}" ;;
  q) probe_q "${@:2}" ;;
  r) probe_r "${@:2}" ;;
  s) probe_s "${@:2}" ;;
  u) probe_u "${@:2}" ;;
  *) echo "usage: bash tools/r5_probes.sh <letter | list>"; exit 2 ;;
esac
