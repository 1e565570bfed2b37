// Issue-rate microbenchmarks for the instructions the KDE sweep is built from (gfx950).
// Prints cycles per wave-instruction per SIMD at an assumed 2.4 GHz plus the wall rate.
// Usage: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 32768;

struct OpFma { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c)); } };
struct OpAdd { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c)); } };
struct OpMul { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c)); } };
struct OpLdexp { static __device__ void run(double (&v)[8], double c) { int n = 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(n)); } };
struct OpRndne { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i])); } };
struct OpCvt { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { int n; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(v[i])); asm volatile("" :: "v"(n)); } } };
struct OpMax { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c)); } };
struct OpCmp { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(v[i]), "v"(c) : "vcc"); } };
struct OpFma32 { static __device__ void run(double (&v)[8], double c) { float* f = (float*)v; float cc = (float)c;
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cc)); } };
struct OpExp32 { static __device__ void run(double (&v)[8], double c) { float* f = (float*)v;
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i])); } };
struct OpLshlAdd { static __device__ void run(double (&v)[8], double c) { int* f = (int*)v; int s = 20;
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_lshl_add_u32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(s)); } };

struct OpFract { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_fract_f64 %0, %0" : "+v"(v[i])); } };
// the two range reductions of the sweep's 2^x, each followed by the same 7 FMAs and the ldexp (11 vs 10 instructions)
struct OpExp2Rndne { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { double nf, r, p; int n;
    asm volatile("v_rndne_f64 %0, %1" : "=v"(nf) : "v"(v[i]));
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(v[i]), "v"(nf));
    asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(p) : "v"(c), "v"(r));
    for (int k = 0; k < 6; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p) : "v"(r), "v"(c));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(nf));
    asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(v[i]) : "v"(p), "v"(n)); } } };
struct OpExp2Fract { static __device__ void run(double (&v)[8], double c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { double r, p; int n;
    asm volatile("v_fract_f64 %0, %1" : "=v"(r) : "v"(v[i]));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(v[i]));
    asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(p) : "v"(c), "v"(r));
    for (int k = 0; k < 6; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p) : "v"(r), "v"(c));
    asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(v[i]) : "v"(p), "v"(n)); } } };

template <typename Op>
__global__ __launch_bounds__(256) void k_valu(double* out, double c) {
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 1.0 + threadIdx.x * 1e-9 + i;
  for (int it = 0; it < ITER; ++it) Op::run(v, c);
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mfma64(double* out, double c) {
  d4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  double a = 1.0 + threadIdx.x * 1e-9, b = c;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mfma32(double* out, double c) {
  f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  float a = 1.0f + threadIdx.x * 1e-6f, b = (float)c;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (s == 12345.678f) out[0] = s;
}
// MFMA f64 + VALU fma in the same wave: 2 MFMA + NV fma per iteration
template <int NV>
__global__ __launch_bounds__(256) void k_mix(double* out, double c) {
  d4 acc[2] = {{0,0,0,0},{0,0,0,0}};
  double a = 1.0 + threadIdx.x * 1e-9, b = c;
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 1.0 + i;
  for (int it = 0; it < ITER; ++it) {
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[j % 8]) : "v"(c));
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[1], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[(j + 4) % 8]) : "v"(c));
  }
  double s = acc[0][0] + acc[1][1];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678) out[0] = s;
}

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// matrix-pipe / VALU overlap for the fp32 sweep: MODE 0 = 2 x v_mfma_f32_16x16x32_bf16, MODE 1 = 2 x v_mfma_f32_16x16x4_f32,
// each with NV f32 VALU instructions (half v_exp_f32, half v_add_f32) per iteration
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_mix32(double* out, double c) {
  f4 acc[2] = {{0,0,0,0},{0,0,0,0}};
  bf8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b8[i] = (__bf16)(float)c; }
  float a = 1.0f + threadIdx.x * 1e-6f, b = (float)c;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MODE == 0) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[h], 0, 0, 0);
      else if (MODE == 1) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[h], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 4; ++j) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[(2 * j) % 8]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(2 * j + 1) % 8]) : "v"(b));
      }
    }
  }
  float s = acc[0][0] + acc[1][1];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
double time_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3.0;
}

int main() {
  double* out; CHECK(hipMalloc(&out, 64));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double clk = 2.4e9;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
  for (int wps = 2; wps <= 4; wps *= 2) {
    dim3 grid(cus * wps), block(256);
    const double waves_per_simd = wps;  // 4 waves per block, 4 SIMDs per CU
    auto report = [&](const char* name, double ms, double instr_per_wave) {
      double cyc = ms * 1e-3 * clk / (instr_per_wave * waves_per_simd);
      printf("  %-28s %8.3f ms  %6.2f cycles/wave-instr/SIMD (@2.4GHz)\n", name, ms, cyc);
    };
    printf("waves per SIMD: %d\n", wps);
#define V(Op, name) report(name, time_ms([&] { hipLaunchKernelGGL(k_valu<Op>, grid, block, 0, 0, out, 1.0000001); }), (double)ITER * 8)
    V(OpFma, "v_fma_f64"); V(OpAdd, "v_add_f64"); V(OpMul, "v_mul_f64"); V(OpLdexp, "v_ldexp_f64"); V(OpRndne, "v_rndne_f64");
    V(OpCvt, "v_cvt_i32_f64"); V(OpMax, "v_max_f64"); V(OpCmp, "v_cmp_lt_f64"); V(OpFma32, "v_fma_f32"); V(OpExp32, "v_exp_f32");
    V(OpLshlAdd, "v_lshl_add_u32"); V(OpFract, "v_fract_f64");
    V(OpExp2Rndne, "2^x rndne form: cyc per 2^x"); V(OpExp2Fract, "2^x fract form: cyc per 2^x");
    report("v_mfma_f64_16x16x4", time_ms([&] { hipLaunchKernelGGL(k_mfma64, grid, block, 0, 0, out, 1.0000001); }), (double)ITER * 4);
    report("v_mfma_f32_16x16x4", time_ms([&] { hipLaunchKernelGGL(k_mfma32, grid, block, 0, 0, out, 1.0000001); }), (double)ITER * 4);
    // mixes: cycles per iteration (2 MFMA + NV FMA)
    auto mix = [&](const char* name, double ms) {
      printf("  %-28s %8.3f ms  %7.1f cycles/iteration/SIMD-wave-slot\n", name, ms, ms * 1e-3 * clk / ((double)ITER * waves_per_simd));
    };
    mix("2 mfma64 + 0 fma64", time_ms([&] { hipLaunchKernelGGL(k_mix<0>, grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma64 + 16 fma64", time_ms([&] { hipLaunchKernelGGL(k_mix<16>, grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma64 + 32 fma64", time_ms([&] { hipLaunchKernelGGL(k_mix<32>, grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma64 + 64 fma64", time_ms([&] { hipLaunchKernelGGL(k_mix<64>, grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma_bf16_16x16x32 + 0 valu32", time_ms([&] { hipLaunchKernelGGL((k_mix32<0, 0>), grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma_bf16_16x16x32 + 16 valu32", time_ms([&] { hipLaunchKernelGGL((k_mix32<0, 16>), grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma_bf16_16x16x32 + 32 valu32", time_ms([&] { hipLaunchKernelGGL((k_mix32<0, 32>), grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma_f32_16x16x4 + 0 valu32", time_ms([&] { hipLaunchKernelGGL((k_mix32<1, 0>), grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma_f32_16x16x4 + 16 valu32", time_ms([&] { hipLaunchKernelGGL((k_mix32<1, 16>), grid, block, 0, 0, out, 1.0000001); }));
    mix("2 mfma_f32_16x16x4 + 32 valu32", time_ms([&] { hipLaunchKernelGGL((k_mix32<1, 32>), grid, block, 0, 0, out, 1.0000001); }));
    mix("0 mfma + 32 valu32 (ref)", time_ms([&] { hipLaunchKernelGGL((k_mix32<2, 32>), grid, block, 0, 0, out, 1.0000001); }));
  }
  return 0;
}
