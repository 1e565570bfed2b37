// Does v_mfma_f64_16x16x4_f64 overlap with VALU work that is NOT double precision?  Round 1 measured that it does not overlap
// v_fma_f64 (tools/microbench.hip); the sum-only tail of the fp64 sweep is mostly fp32 / integer / transcendental work, so the
// answer decides whether that tail can hide behind the two MFMAs of a tile.  Per iteration: 2 MFMA + NV instructions of one kind.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl tools/mfma64_overlap_probe.hip && /tmp/ovl   (result: profiles/r6/mfma64_overlap.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 16384;

// KIND 0 v_fma_f64, 1 v_fma_f32, 2 v_exp_f32, 3 v_lshl_add_u32, 4 v_cvt_f32_f64, 5 v_fract_f64, 6 v_pk_fma_f32, 7 v_ldexp_f32,
//      8 v_cvt_i32_f64, 9 v_cvt_f64_f32, 10 v_ldexp_f64, 11 v_add_f64
template <int KIND>
__device__ __forceinline__ void op(double& d, float& f, int& n, double c) {
    float cf = 1.0000001f;
    if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(c));
    else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(cf));
    else if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f));
    else if (KIND == 3) asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(n));
    else if (KIND == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d));
    else if (KIND == 5) asm volatile("v_fract_f64 %0, %0" : "+v"(d));
    else if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(c));
    else if (KIND == 7) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f) : "v"(n));
    else if (KIND == 8) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(d));
    else if (KIND == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f));
    else if (KIND == 10) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d) : "v"(n));
    else if (KIND == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(c));
}

template <int KIND, int NV, int NM>
__global__ __launch_bounds__(256) void k_mix(double* out, double c) {
    d4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = 1.0 + threadIdx.x * 1e-9, b = c;
    double v[8]; float f[8]; int n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = 1.0 + i; f[i] = 0.5f + i; n[i] = i; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (NM) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[h], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) op<KIND>(v[(j + 4 * h) % 8], f[(j + 4 * h) % 8], n[(j + 4 * h) % 8], c);
        }
    }
    double s = acc[0][0] + acc[1][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + f[i] + n[i];
    if (s == 12345.678) out[0] = s;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3.0;
}

template <int KIND>
void row(const char* name, dim3 grid, double* out, int wps) {
    const double clk = 2.4e9;
    auto cyc = [&](double ms) { return ms * 1e-3 * clk / ((double)ITER * wps); };
    double m0 = cyc(time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, 0, 1>), grid, dim3(256), 0, 0, out, 1.0000001); }));
    double v16 = cyc(time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, 16, 0>), grid, dim3(256), 0, 0, out, 1.0000001); }));
    double v32 = cyc(time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, 32, 0>), grid, dim3(256), 0, 0, out, 1.0000001); }));
    double x16 = cyc(time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, 16, 1>), grid, dim3(256), 0, 0, out, 1.0000001); }));
    double x32 = cyc(time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, 32, 1>), grid, dim3(256), 0, 0, out, 1.0000001); }));
    printf("  %-16s 2 MFMA %6.1f | 16 ops %6.1f  32 ops %6.1f | 2 MFMA + 16 ops %6.1f  + 32 ops %6.1f   (sum %6.1f / %6.1f)\n", name, m0, v16, v32,
           x16, x32, m0 + v16, m0 + v32);
}

int main() {
    double* out; if (hipMalloc(&out, 64) != hipSuccess) return 1;
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    printf("device %s, %d CUs; cycles per iteration per SIMD wave slot at an assumed 2.4 GHz\n", p.name, p.multiProcessorCount);
    for (int wps = 2; wps <= 4; wps *= 2) {
        dim3 grid(p.multiProcessorCount * wps);
        printf("waves per SIMD: %d\n", wps);
        row<0>("v_fma_f64", grid, out, wps);
        row<11>("v_add_f64", grid, out, wps);
        row<5>("v_fract_f64", grid, out, wps);
        row<8>("v_cvt_i32_f64", grid, out, wps);
        row<4>("v_cvt_f32_f64", grid, out, wps);
        row<9>("v_cvt_f64_f32", grid, out, wps);
        row<10>("v_ldexp_f64", grid, out, wps);
        row<1>("v_fma_f32", grid, out, wps);
        row<6>("v_pk_fma_f32", grid, out, wps);
        row<2>("v_exp_f32", grid, out, wps);
        row<7>("v_ldexp_f32", grid, out, wps);
        row<3>("v_lshl_add_u32", grid, out, wps);
    }
    return 0;
}
