// v_mfma_f64_16x16x4_f64 issue rate under the Gram kernel's conditions: 10 accumulator tiles fed by 4 operand fragments
// (pairs I <= J), 4 waves per SIMD, operands with constant / random mantissas.  hipcc --offload-arch=gfx950 -O3 -o
// /tmp/mfma64_probe tools/mfma64_probe.hip && /tmp/mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000;

template <int MODE>   // 0: 10 tiles from 4 fragments; 1: 4 tiles, one fragment pair (the round-1 microbench); 2: 10 tiles, one pair
__global__ __launch_bounds__(256, 4) void k(const double* __restrict__ in, double* out, long long* cyc) {
    double x[4];
    for (int i = 0; i < 4; ++i) x[i] = in[(blockIdx.x * 256 + threadIdx.x) * 4 + i];
    d4 acc[10];
    for (int p = 0; p < 10; ++p) acc[p] = d4{0, 0, 0, 0};
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
            int p = 0;
#pragma unroll
            for (int I = 0; I < 4; ++I)
#pragma unroll
                for (int J = I; J < 4; ++J, ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[I], x[J], acc[p], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], x[1], acc[p], 0, 0, 0);
        } else {
#pragma unroll
            for (int p = 0; p < 10; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], x[1], acc[p], 0, 0, 0);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int p = 0; p < 10; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}

// the Gram kernel's per-chunk body, built up step by step: STEP 0 = 20 MFMAs from registers, 1 = + 8 ds_read_b64 operands,
// 2 = + shift / select / column sums, 3 = + a barrier per chunk, 4 = + rotating priority
template <int STEP>
__global__ __launch_bounds__(256, 4) void g(const double* __restrict__ in, double* out, long long* cyc) {
    __shared__ double lds[2 * 2064];
    for (int e = threadIdx.x; e < 2 * 2064; e += 256) lds[e] = in[(blockIdx.x * 256 + e) % (256 * 1024)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
    double sh[4], cs[4] = {0, 0, 0, 0};
    bool valid[4];
    for (int i = 0; i < 4; ++i) { sh[i] = in[threadIdx.x * 4 + i]; valid[i] = in[threadIdx.x + i] < 1e300; }
    d4 acc[10];
    for (int p = 0; p < 10; ++p) acc[p] = d4{0, 0, 0, 0};
    unsigned slot; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(slot));
    const int off0 = wave * 64 + (kq >> 1) * 32 + c * 2 + (kq & 1);
    const long long t0 = clock64(), w0 = wall_clock64();
    int buf = 0;
    for (int it = 0; it < ITER / 2; ++it) {
        if (STEP >= 4) {
            switch ((it + slot) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        const double* img = lds + buf * 2064 + off0;
        asm volatile("" ::: "memory");   // the LDS image does not change here: keep the reads in the loop
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            double x[4];
#pragma unroll
            for (int I = 0; I < 4; ++I) {
                if (STEP == 0) x[I] = sh[I];
                else if (STEP == 1) x[I] = img[(2 * I + s2) * 258];
                else { const double v = img[(2 * I + s2) * 258] - sh[I]; x[I] = valid[I] ? v : 0.0; cs[I] += x[I]; }
            }
            int p = 0;
#pragma unroll
            for (int I = 0; I < 4; ++I)
#pragma unroll
                for (int J = I; J < 4; ++J, ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[I], x[J], acc[p], 0, 0, 0);
        }
        if (STEP >= 3) __syncthreads();
        buf ^= 1;
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = cs[0] + cs[1] + cs[2] + cs[3];
    for (int p = 0; p < 10; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 4;
    const size_t n = (size_t)blocks * 256 * 4;
    std::vector<double> h(n);
    double *in, *out; long long* cyc;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 2); hipMalloc(&cyc, blocks * 16);
    for (int data = 0; data < 3; ++data) {
        srand(1);
        for (size_t i = 0; i < n; ++i) h[i] = data == 0 ? 0.0 : data == 1 ? 1.0 : (rand() / (double)RAND_MAX - 0.5) * 1e-3 * (1 + (i & 7));
        hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {
            const int per_iter = mode == 1 ? 4 : 10;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(blocks * 2);
            hipMemcpy(c.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
            long long cmax = 0, wmax = 0;
            for (int b = 0; b < blocks; ++b) { if (c[b * 2] > cmax) cmax = c[b * 2]; if (c[b * 2 + 1] > wmax) wmax = c[b * 2 + 1]; }
            printf("data %s  mode %d (%2d tiles): %7.2f shader cycles per MFMA per SIMD, %6.1f ns-derived cycles @2.4GHz, clock %.2f GHz\n",
                   data == 0 ? "zeros " : data == 1 ? "ones  " : "random", mode, per_iter, (double)cmax / (ITER * per_iter * 4.0),
                   wmax * 10e-9 * 2.4e9 / (ITER * per_iter * 4.0), cmax / (wmax * 10.0));
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int step = 0; step < 5; ++step) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (step == 0) hipLaunchKernelGGL(g<0>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
            if (step == 1) hipLaunchKernelGGL(g<1>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
            if (step == 2) hipLaunchKernelGGL(g<2>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
            if (step == 3) hipLaunchKernelGGL(g<3>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
            if (step == 4) hipLaunchKernelGGL(g<4>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        std::vector<long long> c(blocks * 2);
        hipMemcpy(c.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
        long long cmax = 0, wmax = 0, cmin = 1ll << 60;
        for (int b = 0; b < blocks; ++b) { if (c[b * 2] > cmax) cmax = c[b * 2]; if (c[b * 2] < cmin) cmin = c[b * 2]; if (c[b * 2 + 1] > wmax) wmax = c[b * 2 + 1]; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("gram body step %d: %7.2f shader cycles per MFMA per SIMD (slowest block; fastest %7.2f), clock %.2f GHz; kernel %.1f us = %.1f ns per MFMA per SIMD\n", step,
               (double)cmax / (ITER * 10 * 4.0), (double)cmin / (ITER * 10 * 4.0), cmax / (wmax * 10.0), ms * 1e3, ms * 1e6 / (ITER * 10 * 4.0));
    }
    return 0;
}
