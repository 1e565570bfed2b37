# fp64 headline (with the oracle parity block) + cv64 + bounded C3 with the base library and the -DPBN_EXP2_F32=1 variant   bash tools/expf32_probe.sh
cd $GRAFT_REPO_ROOT
cp pybnesian_amd/libpbn_hip.so /tmp/libpbn_base.so
f64() { python3 bench.py --hc none --no-e2e --no-extra-legs --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 headline ms/step %.3f  frac %.4f  parity max_rel_logl %.3g rel_slogl %.3g ok %s' % (d['ms_per_step'], d['roofline']['frac'], d['parity']['max_rel_logl'], d['parity']['rel_slogl'], d['parity']['ok']))"; }
hc() { python3 bench.py --no-c3 --no-e2e --no-cpu-baseline --no-extra-legs --hc $1 --hc-max-iters $2 --steps 1 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['secondary']; print('$1 %.3f s  cells %d iterations %d arcs %d' % (d['estimate_s'], d['cells_scored'], d['iterations'], d['arcs_found']))"; }
all() { f64; hc cv64 1; hc c3 1; }
echo "== base"; all
for v in "$@"; do cp build/variants/libpbn_$v.so pybnesian_amd/libpbn_hip.so; echo "== $v"; all; done
cp /tmp/libpbn_base.so pybnesian_amd/libpbn_hip.so
