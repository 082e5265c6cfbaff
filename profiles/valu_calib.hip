// Vector-ALU issue rate of gfx950, measured: how many wave64 fp32 instructions per second the chip sustains when every
// SIMD holds many waves that do nothing else.  Calibrates the "valu" roofline of bench.py (the blend kernels are bound
// by it), the way pmc_calib.hip calibrates the HBM counters.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_build/valu_calib profiles/valu_calib.hip && profiles/_build/valu_calib
// Kernels: 16 independent accumulators per lane, ITER rounds of one instruction per accumulator.
//   fma      v_fma_f32            (plain fp32)
//   pk_fma   v_pk_fma_f32         (two fp32 per lane and instruction)
//   mix      12 v_fma_f32 + 2 v_exp_f32 + 2 v_rcp_f32 per round (roughly the blend loop's transcendental share)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using f2 = __attribute__((ext_vector_type(2))) float;
constexpr int ITER = 4096;

__global__ __launch_bounds__(256) void k_fma(float* out, float x, float y) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));   // (the compiler packs plain C)
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk_fma(float* out, float x, float y) {
  f2 a[16];
  const f2 xx = {x, x}, yy = {y, y};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (f2){(float)(threadIdx.x + i), (float)i};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(xx), "v"(yy));
  }
  f2 s = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void k_mix(float* out, float x, float y) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.001f * (float)(threadIdx.x + i);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
    asm volatile("v_exp_f32 %0, %0" : "+v"(a[12]));
    asm volatile("v_exp_f32 %0, %0" : "+v"(a[13]));
    asm volatile("v_rcp_f32 %0, %0" : "+v"(a[14]));
    asm volatile("v_rcp_f32 %0, %0" : "+v"(a[15]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
double run(K kernel, int blocks, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0 * 1e-3;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out;
  (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8 * 4);
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"results\": [", p.gcnArchName, cus, p.clockRate / 1000);
  bool first = true;
  for (int wps : {1, 2, 4, 8}) {                         // waves per SIMD
    const int blocks = cus * wps;                        // 256 threads = 4 waves = one per SIMD of a CU
    const double waves = (double)blocks * 4;
    const double insts = waves * ITER * 16;              // wave-instructions of the measured kind per launch
    const double t_f = run(k_fma, blocks, out), t_p = run(k_pk_fma, blocks, out), t_m = run(k_mix, blocks, out);
    printf("%s{\"waves_per_simd\": %d, \"fma_Ginst_s\": %.1f, \"pk_fma_Ginst_s\": %.1f, \"mix_Ginst_s\": %.1f}", first ? "" : ", ",
           wps, insts / t_f / 1e9, insts / t_p / 1e9, insts / t_m / 1e9);
    first = false;
  }
  printf("]}\n");
  (void)hipFree(out);
  return 0;
}
