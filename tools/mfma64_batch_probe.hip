// What does a v_mfma_f64 <-> VALU transition cost?  tools/mfma64_overlap_probe.hip showed that 2 MFMA + 32 VALU instructions take MORE than the
// two alone.  Here the same work - NM MFMAs and 16 NM VALU instructions (v_exp_f32 / v_cvt_f64_f32 / v_lshl_add_u32 / v_fmac_f64 in the
// sweep's proportions) - is issued in groups: G MFMAs back to back (independent accumulators), then their 16 G VALU instructions.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mbp tools/mfma64_batch_probe.hip && /tmp/mbp      (result: profiles/r6/mfma64_batch_probe.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 8192;

template <int G>   // MFMAs per group; 8 MFMAs per iteration in 8 / G groups
__global__ __launch_bounds__(256) void k_batch(double* out, double c) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = c;
    double v[8]; float f[8]; int n[8];
    for (int i = 0; i < 8; ++i) { v[i] = 1.0 + i; f[i] = 0.5f + i; n[i] = i; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int g0 = 0; g0 < 8; g0 += G) {
#pragma unroll
            for (int g = g0; g < g0 + G; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4 * G; ++j) {   // 16 VALU instructions per MFMA: 4 x (exp, cvt, lshl_add, fmac)
                asm volatile("v_exp_f32 %0, %0" : "+v"(f[j % 8]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[(j + 1) % 8]) : "v"(f[j % 8]));
                asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(n[j % 8]));
                asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(v[(j + 2) % 8]) : "v"(c));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i] + f[i] + n[i];
    if (s == 12345.678) out[0] = s;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3.0;
}

int main() {
    double* out; if (hipMalloc(&out, 64) != hipSuccess) return 1;
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    printf("device %s, %d CUs; cycles per MFMA + its 16 VALU instructions (per SIMD wave slot, assumed 2.4 GHz)\n", p.name, p.multiProcessorCount);
    for (int wps = 2; wps <= 4; wps *= 2) {
        dim3 grid(p.multiProcessorCount * wps);
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * wps); };
        printf("waves per SIMD %d:  groups of 1 MFMA %6.1f   2 %6.1f   4 %6.1f   8 %6.1f\n", wps,
               cyc(time_ms([&] { hipLaunchKernelGGL(k_batch<1>, grid, dim3(256), 0, 0, out, 1.0000001); })),
               cyc(time_ms([&] { hipLaunchKernelGGL(k_batch<2>, grid, dim3(256), 0, 0, out, 1.0000001); })),
               cyc(time_ms([&] { hipLaunchKernelGGL(k_batch<4>, grid, dim3(256), 0, 0, out, 1.0000001); })),
               cyc(time_ms([&] { hipLaunchKernelGGL(k_batch<8>, grid, dim3(256), 0, 0, out, 1.0000001); })));
    }
    return 0;
}
