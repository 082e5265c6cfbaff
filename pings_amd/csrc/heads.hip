// Output heads of the per-neighbour decoders over a kNN query (utils/mesher.py:132-153, utils/tracker.py:322-331):
// the reference applies the activation to every neighbour's decoder output ([B, k, C]) and then takes the IDW-weighted
// sum over the k neighbours — three to five full-size torch passes per head and batch.  One streaming pass here:
//   colour    out[b, c] = sum_j w[b, j] sigmoid(raw[b, j, c])                 (Decoder.regress_color, decoder.py:133-134)
//   semantic  s[b, c]   = sum_j w[b, j] log_softmax(raw[b, j, :])[c],  label[b] = argmax_c s[b, c] (first maximum, as
//             torch.argmax)                                                    (Decoder.sem_label_prob, decoder.py:119-122)
// w == nullptr is the `weighted_first` form: k = 1, weight one.  HBM-bound: 4 k C bytes in, 4 C (or 8) out per query.
#include "common.hpp"

namespace {

constexpr int kMaxK = 16;

__global__ __launch_bounds__(256) void head_color_kernel(long long B, int k, int C, const float* __restrict__ raw,
                                                         const float* __restrict__ w, float* __restrict__ out) {
  // one thread per (query, channel): consecutive threads read consecutive channels of a neighbour row
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const long long b = e / C;
  const int c = (int)(e - b * C);
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    const float x = raw[((size_t)b * k + j) * C + c];
    const float s = 1.0f / (1.0f + expf(-x));
    acc += (w ? w[(size_t)b * k + j] : 1.0f) * s;
  }
  out[e] = acc;
}

__global__ __launch_bounds__(256) void head_sem_kernel(long long B, int k, int C, const float* __restrict__ raw,
                                                       const float* __restrict__ w, float* __restrict__ prob,
                                                       long long* __restrict__ label) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float lse[kMaxK], wj[kMaxK];
#pragma unroll
  for (int j = 0; j < kMaxK; ++j) {
    lse[j] = 0.f;
    wj[j] = 0.f;
    if (j < k) {
      const float* r = raw + ((size_t)b * k + j) * C;
      float m = r[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, r[c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(r[c] - m);
      lse[j] = m + logf(s);
      wj[j] = w ? w[(size_t)b * k + j] : 1.0f;
    }
  }
  float best = 0.f;
  int arg = 0;
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxK; ++j)
      if (j < k) acc += wj[j] * (raw[((size_t)b * k + j) * C + c] - lse[j]);
    if (prob) prob[(size_t)b * C + c] = acc;
    if (c == 0 || acc > best) { best = acc; arg = c; }
  }
  label[b] = arg;
}

}  // namespace

PINGS_API int pings_head_reduce(const float* raw, const float* weight, int64_t B, int32_t k, int32_t C, int32_t mode,
                                float* out_value, int64_t* out_label, void* stream) {
  PINGS_ARG_CHECK(B >= 0 && k >= 1 && k <= kMaxK && C >= 1, "bad shape");
  if (B == 0) return PINGS_OK;
  PINGS_ARG_CHECK(raw != nullptr, "null input");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("head_reduce", st);
  if (mode == PINGS_HEAD_COLOR) {
    PINGS_ARG_CHECK(out_value != nullptr, "colour head needs out_value");
    const long long n = (long long)B * C;
    head_color_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(B, k, C, raw, weight, out_value);
  } else if (mode == PINGS_HEAD_SEMANTIC) {
    PINGS_ARG_CHECK(out_label != nullptr, "semantic head needs out_label");
    head_sem_kernel<<<(unsigned)((B + 255) / 256), 256, 0, st>>>(B, k, C, raw, weight, out_value,
                                                                 reinterpret_cast<long long*>(out_label));
  } else {
    PINGS_ARG_CHECK(false, "unknown head mode");
  }
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
