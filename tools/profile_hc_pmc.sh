# PMC pass (SQ counters only: never combined with a trace domain) over a hill-climb config, aggregated per kernel:
#   bash tools/profile_hc_pmc.sh [cv64|c3|c5mmhc] [max_iters] [out name under gpurun_out]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
HC=${1:-cv64}; IT=${2:-1}; OUT=$R/gpurun_out/${3:-hc_pmc}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --no-c3 --no-e2e --no-cpu-baseline --hc $HC --hc-max-iters $IT --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
cd $R
python3 tools/pmc_aggregate.py $OUT/pmc_per_dispatch.json $OUT/pmc_sq
python3 - $OUT/pmc_per_dispatch.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for name, k in d.items():
    if "sweep" not in name:
        continue
    simd = k["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0      # SIMD cycles per dispatch (8 XCDs x ... see MI355X_MICROARCH.md)
    print(name[:90])
    print("   dispatches %d  gui_active/8 %.3e cycles  VALU %.3e  MFMA %.3e  mfma_busy/simd %.3f  (mfma_busy + 4*valu)/simd %.3f  wave_cycles/simd %.2f waves"
          % (k["dispatches_SQ_INSTS_VALU"], k["GRBM_GUI_ACTIVE"] / 8.0, k["SQ_INSTS_VALU"], k["SQ_INSTS_MFMA"], k["SQ_VALU_MFMA_BUSY_CYCLES"] / simd,
             (k["SQ_VALU_MFMA_BUSY_CYCLES"] + 4.0 * k["SQ_INSTS_VALU"]) / simd, k["SQ_WAVE_CYCLES"] / simd))
PY
find $OUT -name "*.csv" -size +1M -delete
