// v_mfma_f64_16x16x4_f64 issue rate under the Gram kernel's conditions: 10 accumulator tiles fed by 4 operand fragments
// (pairs I <= J), 4 waves per SIMD, operands with constant / random mantissas.  Result (profiles/r2/mfma64_probe.txt): 64.0
// shader cycles per MFMA per SIMD in every case; random mantissas lower the CLOCK (2.36 -> 2.1-2.2 GHz), not the cycle count.  hipcc --offload-arch=gfx950 -O3 -o
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
    return 0;
}
