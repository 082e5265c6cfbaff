// fp32 matrix-pipe rate of gfx950, measured: how many v_mfma_f32_32x32x2_f32 per second the chip sustains when every
// SIMD does nothing else (independent accumulators, operands in registers, no memory).  The decoder kernels' roofline
// (bench.py decoder leg) is quoted against the data-sheet 157.3 TFLOP/s = 256 CUs x 4 SIMDs x
// 64 flop/cycle x 2.4 GHz; this program says what the clock actually is under that load.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_build/mfma_calib profiles/mfma_calib.hip && profiles/_build/mfma_calib
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2048;

template <int ACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, float x, float y) {
  f32x16 a[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) a[i][q] = 0.f;
  const float va = x + threadIdx.x * 1e-6f, vb = y;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, a[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += a[i][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
double run(K kernel, int blocks, float* out, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e-3;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out;
  (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"datasheet_TFLOPs\": %.1f, \"results\": [", p.gcnArchName, cus,
         p.clockRate / 1000, cus * 4 * 64.0 * p.clockRate * 1e3 / 1e12);
  bool first = true;
  for (int wps : {1, 2}) {
    for (int acc : {1, 2, 4}) {
      for (int reps : {1, 20}) {           // one launch (~0.4 ms) and a 20-launch burst (~8 ms: sustained clocks)
        const int blocks = cus * wps;
        const double mfmas = (double)blocks * 4 * ITER * acc;
        const double t = acc == 1 ? run(k_mfma<1>, blocks, out, reps)
                       : acc == 2 ? run(k_mfma<2>, blocks, out, reps) : run(k_mfma<4>, blocks, out, reps);
        printf("%s{\"waves_per_simd\": %d, \"accumulators\": %d, \"launches\": %d, \"ms\": %.4f, \"TFLOPs\": %.1f}", first ? "" : ", ",
               wps, acc, reps, t * 1e3 , mfmas * 4096.0 / t / 1e12);
        first = false;
      }
    }
  }
  printf("]}\n");
  (void)hipFree(out);
  return 0;
}
