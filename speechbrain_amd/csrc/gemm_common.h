// Shared by the contraction translation units (gemm.hip, gemm_lp.hip): the epilogue's activation and the fp32 launch record.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SBK_ACT_SWISH: return v / (1.0f + expf(-v));
    case SBK_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case SBK_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case SBK_ACT_LEAKY_RELU: return v > 0.0f ? v : 0.01f * v;
    default: return v;
  }
}

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* R;
  float* C;
  int lda, ldw, ldr, ldc, M, N, K, act;
  float alpha;
  const int32_t* seq_len;  // optional: rows are [batch][rows_per_seq]; rows >= seq_len[batch] produce v = 0
  int rows_per_seq;
};

}  // namespace
