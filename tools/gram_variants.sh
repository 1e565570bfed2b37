# Gram kernel duration (2M x 64 fp64, tools/gram_bench.py) for the three kernels: PBN_GRAM_LDS = 2 (default: per-wave LDS-DMA
# rings, gram_glds_kernel), 1 (register-staged block-wide LDS image, gram_lds_kernel), 0 (rows in registers, gram_kernel), the
# default kernel's floors (PBN_GRAM_DEBUG 1 = no MFMAs, 2 = no DMA) and the float table (ring kernel and gram_lds_kernel<float>).
# Usage (GPU box): bash tools/gram_variants.sh
for v in 2 1 0; do echo "== PBN_GRAM_LDS=$v"; PBN_GRAM_LDS=$v bash tools/gram_timing.sh gram_v$v | grep "gram_[a-z]*_*kernel\|per call"; done
# the floor variants exist in measurement builds only: rebuild the library with -DPBN_GRAM_MEASURE on the box, restore it afterwards
touch pybnesian_amd/csrc/stats_kernels.hip; make -C pybnesian_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DPBN_GRAM_MEASURE" > /dev/null
for d in 1 2; do echo "== PBN_GRAM_LDS=2 PBN_GRAM_DEBUG=$d"; PBN_GRAM_DEBUG=$d bash tools/gram_timing.sh gram_v2d$d | grep "gram_glds\|per call"; done
touch pybnesian_amd/csrc/stats_kernels.hip; make -C pybnesian_amd/csrc > /dev/null
echo "== float table, PBN_GRAM_LDS=2 (gram_glds_f32_kernel)"; GRAM_DTYPE=f32 bash tools/gram_timing.sh gram_f32 | grep "gram_glds\|per call"
echo "== float table, PBN_GRAM_LDS=1 (gram_lds_kernel<float>)"; GRAM_DTYPE=f32 PBN_GRAM_LDS=1 bash tools/gram_timing.sh gram_f32b | grep "gram_lds\|per call"
