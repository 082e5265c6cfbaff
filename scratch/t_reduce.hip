#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ inline float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float xor16_sum(float v) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ inline float xor32_sum(float v) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(float* out) {
  int lane = threadIdx.x;
  float v = (float)(1 << (lane % 16)) ;  // distinct per lane-in-row
  float w = (float)lane;
  out[lane] = dpp_mov<0xb1>(w);
  out[64 + lane] = dpp_mov<0x4e>(w);
  out[128 + lane] = dpp_mov<0x124>(w);
  out[192 + lane] = dpp_mov<0x128>(w);
  {
    const unsigned x = __builtin_bit_cast(unsigned, w);
    unsigned y = x; asm volatile("" : "+v"(y));
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[256 + lane] = __builtin_bit_cast(float, r[0]);
    out[320 + lane] = __builtin_bit_cast(float, r[1]);
    auto s = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[384 + lane] = __builtin_bit_cast(float, s[0]);
    out[448 + lane] = __builtin_bit_cast(float, s[1]);
  }
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"qp1032","qp2301","ror4","ror8","p16s.0","p16s.1","p32s.0","p32s.1"};
  for (int r = 0; r < 8; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %g", h[r*64+i]); printf("\n"); }
  return 0;
}
