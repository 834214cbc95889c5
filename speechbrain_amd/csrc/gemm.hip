// fp32 MFMA GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)   (sbk_gemm_nt_f32)
//
// Roofline: MFMA fp32 (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense on MI355X).
// One workgroup owns a BM x BN tile of C; its waves each own a WM x WN
// sub-tile made of 32x32 MFMA accumulators.  A and W panels are staged through
// LDS as [rows][BK+1] (odd pitch => the 32 lanes of an MFMA operand read hit 32
// distinct banks), loaded from HBM as 16-byte vectors along K (both operands
// are K-contiguous, so every global load is a full 128-byte line per 8 lanes)
// into registers one K tile ahead, so HBM latency hides under the MFMAs.
// The epilogue (bias, activation, scaled residual) runs on the accumulators in
// registers and writes 128-byte rows (32 lanes x 4 B) per store instruction.
#include "common.h"
#include "internal.h"

#include <map>
#include <mutex>
#include <type_traits>

namespace {

using sbk::f32x16;

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

// Epilogue of a 32x32 register tile held by ONE wave (lane: column r, rows (q&3) + 8*(q>>2) + 4*half): straight-line
// code -- the residual rows / sequence lengths are requested together, the activation is chosen by ONE uniform
// switch outside the per-row work, and the sixteen row stores are issued back to back (a branchy per-row loop makes
// the compiler drain the memory counter before every store: sixteen serialized round trips on a 5 us kernel).
__device__ __forceinline__ void tile_epilogue_32x32(const GemmArgs& g, float (&v)[16], int mt, int nt, int r, int half) {
  const int col = nt * 32 + r;
  const bool col_ok = col < g.N;
  const float bv = (g.bias && col_ok) ? g.bias[col] : 0.0f;
  float res[16];
  int len[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
    const bool live = col_ok && row < g.M;
    res[q] = (live && g.R) ? g.R[(size_t)row * g.ldr + col] : 0.0f;
    len[q] = (live && g.seq_len) ? g.seq_len[row / g.rows_per_seq] : 0x7fffffff;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] += bv;
  switch (g.act) {  // uniform
    case SBK_ACT_SWISH:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = v[q] / (1.0f + expf(-v[q]));
      break;
    case SBK_ACT_GELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.70710678118654752440f));
      break;
    case SBK_ACT_RELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = v[q] > 0.0f ? v[q] : 0.0f;
      break;
    case SBK_ACT_LEAKY_RELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = v[q] > 0.0f ? v[q] : 0.01f * v[q];
      break;
    default: break;
  }
  if (g.seq_len) {  // uniform
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      v[q] = (row % g.rows_per_seq) >= len[q] ? 0.0f : v[q] * g.alpha;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] *= g.alpha;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] += res[q];
  sbk::sched_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
    if (col_ok && row < g.M) g.C[(size_t)row * g.ldc + col] = v[q];
  }
}

// ---------------------------------------------------------------------------
// bf16-operand fast path (sbk_gemm_nt_bf16, SURVEY 8b "fast entry points"): C = epilogue(bf16(A) . Wb^T) with fp32
// accumulation on v_mfma_f32_32x32x16_bf16 (16x the f32 matrix rate).  A stays fp32 in HBM -- every kernel around
// the contraction (LayerNorm, attention, GLU/conv, residual stream) is the fp32 one -- and is rounded to bf16 (RNE) on
// its way into LDS; Wb is the weight matrix converted once by the caller.  Same 128x128 tiling, XCD-aware order and
// epilogue as the f32 kernel; LDS rows are 32 bf16 + 8 pad (80 B: the 16-lane groups of a ds_read_b128 hit 16
// distinct 4-bank groups), each operand fragment is one ds_read_b128 of 8 consecutive k.  With the MFMA work cut
// 16x the kernel is bound by the fp32 A / C traffic (4 B per element each), not by the matrix pipe.
struct GemmBf16Args {
  const float* A;
  const void* W;  // [N,K] reduced-precision bits: bf16 / fp16 (2 bytes) or fp8 e4m3 (1 byte)
  const float* bias;
  const float* R;
  float* C;
  int lda, ldw, ldr, ldc, M, N, K, act;
  float alpha;
  const int32_t* seq_len;
  int rows_per_seq;
  // fp8 only: A is multiplied by 448 / a_absmax[0] (device scalar) before it is rounded to e4m3, the accumulators by
  // a_absmax[0] / 448 * w_scale afterwards (w_scale = the weight's own absmax / 448, applied when it was quantised)
  const float* a_absmax;
  float w_scale;
};

// DT: 0 = bf16, 1 = fp16 (operands 8 x 16 bit per lane), 2 = fp8 e4m3 (8 x 8 bit per lane); all on the
// 32x32x16 matrix-core shape with fp32 accumulation.
template <int BM, int BN, int DT>
__global__ void __launch_bounds__(256, 2) gemm_nt_lp_kernel(GemmBf16Args g) {
  using Elem = typename std::conditional<DT == 2, unsigned char, unsigned short>::type;
  constexpr int ES = (int)sizeof(Elem);
  constexpr int BK = 32, PITCH = BK + 16 / ES;  // elements per LDS row: 16 bytes of padding
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int APER = BM * BK / 4 / 256;          // float4 slots of the A panel per thread
  constexpr int WV = 16 / ES;                      // W elements per 16-byte load
  constexpr int WSLOTS = BN * BK / WV;             // 16-byte slots of the W panel
  constexpr int WPER = (WSLOTS + 255) / 256;
  static_assert(APER >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) Elem As[BM][PITCH];
  __shared__ __attribute__((aligned(16))) Elem Ws[BN][PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int id = by * gx + bx;
    if (nwg % 8 == 0) {
      const int swz = (id % 8) * (nwg / 8) + id / 8;
      bx = swz % gx;
      by = swz / gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int lrow = lane & 31, kh = lane >> 5;
  float a_mul = 1.0f, out_mul = 1.0f;
  if constexpr (DT == 2) {
    const float amax = fmaxf(g.a_absmax ? g.a_absmax[0] : 448.0f, 1e-30f);
    a_mul = 448.0f / amax;
    out_mul = amax / 448.0f * g.w_scale;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float4 ra[APER];
  uint4 rw[WPER];
  const Elem* Wp = reinterpret_cast<const Elem*>(g.W);
  const bool interior = m0 + BM <= g.M && n0 + BN <= g.N && (g.K % BK) == 0;  // uniform: unpredicated panel loads
  auto fetch = [&](int k0) SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < APER; ++i) {
      const int s = tid + i * 256, rr = s / (BK / 4), c = (s % (BK / 4)) * 4;
      const int gr = m0 + rr, gk = k0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (interior || (gr < g.M && gk < g.K)) v = *reinterpret_cast<const float4*>(g.A + (size_t)gr * g.lda + gk);  // K % 8 == 0
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
      const int s = tid + i * 256, rr = s / (BK / WV), c = (s % (BK / WV)) * WV;
      const int gr = n0 + rr, gk = k0 + c;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (s < WSLOTS && (interior || (gr < g.N && gk < g.K))) v = *reinterpret_cast<const uint4*>(Wp + (size_t)gr * g.ldw + gk);
      rw[i] = v;
    }
  };
  auto commit = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < APER; ++i) {
      const int s = tid + i * 256, rr = s / (BK / 4), c = (s % (BK / 4)) * 4;
      if constexpr (DT == 0) {
        uint2 p;
        p.x = (unsigned)sbk::f32_to_bf16(ra[i].x) | ((unsigned)sbk::f32_to_bf16(ra[i].y) << 16);
        p.y = (unsigned)sbk::f32_to_bf16(ra[i].z) | ((unsigned)sbk::f32_to_bf16(ra[i].w) << 16);
        *reinterpret_cast<uint2*>(&As[rr][c]) = p;
      } else if constexpr (DT == 1) {
        uint2 p;
        p.x = (unsigned)sbk::f32_to_f16(ra[i].x) | ((unsigned)sbk::f32_to_f16(ra[i].y) << 16);
        p.y = (unsigned)sbk::f32_to_f16(ra[i].z) | ((unsigned)sbk::f32_to_f16(ra[i].w) << 16);
        *reinterpret_cast<uint2*>(&As[rr][c]) = p;
      } else {
        const unsigned p = (unsigned)sbk::f32x2_to_fp8(ra[i].x * a_mul, ra[i].y * a_mul) |
                           ((unsigned)sbk::f32x2_to_fp8(ra[i].z * a_mul, ra[i].w * a_mul) << 16);
        *reinterpret_cast<unsigned*>(&As[rr][c]) = p;
      }
    }
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
      const int s = tid + i * 256, rr = s / (BK / WV), c = (s % (BK / WV)) * WV;
      if (s < WSLOTS) *reinterpret_cast<uint4*>(&Ws[rr][c]) = rw[i];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    commit();
    __syncthreads();
    if (k0 + BK < g.K) fetch(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      if constexpr (DT == 2) {
        sbk::fp8x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const sbk::fp8x8*>(&As[wm0 + i * 32 + lrow][ks + kh * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const sbk::fp8x8*>(&Ws[wn0 + j * 32 + lrow][ks + kh * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_fp8(a[i], b[j], acc[i][j]);
      } else if constexpr (DT == 1) {
        sbk::f16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const sbk::f16x8*>(&As[wm0 + i * 32 + lrow][ks + kh * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const sbk::f16x8*>(&Ws[wn0 + j * 32 + lrow][ks + kh * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_f16(a[i], b[j], acc[i][j]);
      } else {
        sbk::bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const sbk::bf16x8*>(&As[wm0 + i * 32 + lrow][ks + kh * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const sbk::bf16x8*>(&Ws[wn0 + j * 32 + lrow][ks + kh * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(a[i], b[j], acc[i][j]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + j * 32 + lrow;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= g.M) continue;
        float v = apply_act(acc[i][j][r] * out_mul + bv, g.act) * g.alpha;
        if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
        if (g.R) v += g.R[(size_t)row * g.ldr + col];
        g.C[(size_t)row * g.ldc + col] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = sbk::f32_to_bf16(x[i]);
}
__global__ void __launch_bounds__(256) f32_to_f16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = sbk::f32_to_f16(x[i]);
}
// W [N][K] fp32 -> [N][K/32][3][32] bf16: the three exact pieces of every element (hi = bf16(x), mid = bf16(x - hi),
// lo = x - hi - mid, round to nearest even), one 192-byte record per row and 32-deep K tile -- the W operand of gemm_nt_sk_kernel<.., X3>
__global__ void __launch_bounds__(256) split_bf16x3_kernel(const float* __restrict__ W, int ldw, unsigned short* __restrict__ out,
                                                           long n, int K) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / K;
    const int k = (int)(i - row * K);
    const float x = W[row * ldw + k];
    const unsigned short hi = sbk::f32_to_bf16(x);
    const float r = x - __uint_as_float((unsigned)hi << 16);
    const unsigned short mid = sbk::f32_to_bf16(r);
    const float q = r - __uint_as_float((unsigned)mid << 16);  // <= 7 significant bits: a bf16 exactly
    unsigned short* o = out + (row * (K / 32) + k / 32) * 96 + (k & 31);
    o[0] = hi;
    o[32] = mid;
    o[64] = sbk::f32_to_bf16(q);
  }
}
// y = e4m3(x * mul), two values per thread (n even)
__global__ void __launch_bounds__(256) f32_to_fp8_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n2,
                                                         float mul) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
    y[i] = sbk::f32x2_to_fp8(x[2 * i] * mul, x[2 * i + 1] * mul);
}
// out[0] = max |x| (non-negative floats order like their bit patterns: atomicMax on the int image; out zeroed by the caller)
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int* __restrict__ out, long n) {
  float m = 0.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  m = sbk::wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, (int)__float_as_uint(m));
}

// Register-staged panel: global -> registers (issued early, in flight under the MFMAs of the previous
// K tile) -> LDS [rows][BK+1].
template <int ROWS, int BK, int NT>
struct PanelStage {
  static constexpr int V = BK / 4;                       // float4 slots per row
  static constexpr int PER = (ROWS * V + NT - 1) / NT;   // float4 slots per thread
  float4 r[PER];

  template <bool VEC>
  __device__ __forceinline__ void fetch(const float* __restrict__ src, int ld, int row0, int nrows, int k0, int K,
                                        int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      const int rr = s / V, c = (s % V) * 4;
      const int gr = row0 + rr, gk = k0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s < ROWS * V && gr < nrows) {
        const float* p = src + (size_t)gr * ld + gk;
        if (VEC && gk + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
      r[i] = v;
    }
  }
  // Interior tiles (whole panel inside the matrix, K % BK == 0, 16-byte aligned rows): no predicates at all.  The
  // predicated form above costs a branch + mask sequence per load (~150 instructions per K tile and wave).
  __device__ __forceinline__ void fetch_interior(const float* __restrict__ src, int ld, int row0, int k0, int tid) {
    static_assert((ROWS * V) % NT == 0, "fetch_interior: the panel must divide evenly over the threads");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      r[i] = *reinterpret_cast<const float4*>(src + (size_t)(row0 + s / V) * ld + k0 + (s % V) * 4);
    }
  }
  // [rows][BK+4] layout: one 16-byte LDS store per slot (rows stay 16-byte aligned: (BK+4)*4 is a multiple of 16)
  __device__ __forceinline__ void commit_vec(float (*dst)[BK + 4], int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      if (s < ROWS * V) *reinterpret_cast<float4*>(&dst[s / V][(s % V) * 4]) = r[i];
    }
  }
  __device__ __forceinline__ void commit(float (*dst)[BK + 1], int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      if ((ROWS * V) % NT == 0 || s < ROWS * V) {  // evenly divided panels: every slot exists (no mask, no branch)
        const int rr = s / V, c = (s % V) * 4;
        dst[rr][c] = r[i].x;
        dst[rr][c + 1] = r[i].y;
        dst[rr][c + 2] = r[i].z;
        dst[rr][c + 3] = r[i].w;
      }
    }
  }
};

template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__device__ __forceinline__ void gemm_nt_tile(const GemmArgs& g, int* tile_x = nullptr, int* tile_y = nullptr) {
  constexpr int WAVES_N = BN / WN;
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  __shared__ float As[BM][BK + 1];
  __shared__ float Ws[BN][BK + 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  // XCD-aware tile order (workgroup id % 8 = XCD): each XCD walks a contiguous range of tiles, so the
  // A row panel shared by neighbouring tiles is fetched into one L2 instead of eight.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int id = by * gx + bx;
    if (nwg % 8 == 0) {
      const int swz = (id % 8) * (nwg / 8) + id / 8;
      bx = swz % gx;
      by = swz / gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  if (tile_x) {  // (the caller continues on this tile)
    *tile_x = bx;
    *tile_y = by;
  }
  const int lrow = lane & 31, lk = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  PanelStage<BM, BK, NT> pa;
  PanelStage<BN, BK, NT> pw;
  // (uniform per workgroup) interior tile: unpredicated panel loads
  const bool interior = VEC && m0 + BM <= g.M && n0 + BN <= g.N && (g.K % BK) == 0 && (BM * (BK / 4)) % NT == 0 &&
                        (BN * (BK / 4)) % NT == 0;
  auto fetch = [&](int k0) {
    if constexpr ((BM * (BK / 4)) % NT == 0 && (BN * (BK / 4)) % NT == 0) {
      if (interior) {
        pa.fetch_interior(g.A, g.lda, m0, k0, tid);
        pw.fetch_interior(g.W, g.ldw, n0, k0, tid);
        return;
      }
    }
    pa.template fetch<VEC>(g.A, g.lda, m0, g.M, k0, g.K, tid);
    pw.template fetch<VEC>(g.W, g.ldw, n0, g.N, k0, g.K, tid);
  };
  fetch(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    pa.commit(As, tid);
    pw.commit(Ws, tid);
    __syncthreads();
    if (k0 + BK < g.K) fetch(k0 + BK);  // next K tile: loads fly while this tile is multiplied
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[wm0 + i * 32 + lrow][kk + lk];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Ws[wn0 + j * 32 + lrow][kk + lk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x2(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

  // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5).  (The batched straight-line epilogue of
  // the register-operand kernels was tried here: it costs registers -- 3 -> 2 waves per SIMD -- and 17 % throughput.)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + j * 32 + lrow;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        float v = apply_act(acc[i][j][r] + bv, g.act) * g.alpha;
        if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
        if (g.R) v += g.R[(size_t)row * g.ldr + col];
        g.C[(size_t)row * g.ldc + col] = v;
      }
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool VEC>
// (two waves per SIMD = a 256-register budget wherever the accumulators fit it: with the 512-register budget of one wave
// per SIMD the compiler parks the accumulators in AGPRs and copies every one of them in and out around the K loop body)
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, (WM * WN <= 64 * 64) ? 2 : 1) gemm_nt_kernel(GemmArgs g) {
  gemm_nt_tile<BM, BN, BK, WM, WN, VEC>(g);
}

// The same tile over one K slice per blockIdx.z: raw partial products to ws[z][M][N] (splitk_reduce_kernel applies the
// epilogue).  For few-row, long-K shapes (the decoder's second feed-forward projection at ~1 K rows): 64x64 LDS tiles
// move half the operand bytes of the 32x32 register-operand tiles through L2, and the K split puts 2-3 workgroups on
// every CU so that their MFMA and load phases overlap.
template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, (WM * WN <= 64 * 64) ? 2 : 1) gemm_nt_splitk_kernel(GemmArgs g, float* __restrict__ ws,
                                                                                     int kper) {
  const int z = blockIdx.z;
  g.A += (size_t)z * kper;
  g.W += (size_t)z * kper;
  g.K = (g.K - z * kper) < kper ? (g.K - z * kper) : kper;
  g.C = ws + (size_t)z * g.M * g.N;
  g.ldc = g.N;
  g.bias = nullptr;
  g.R = nullptr;
  g.act = SBK_ACT_NONE;
  g.alpha = 1.0f;
  g.seq_len = nullptr;
  gemm_nt_tile<BM, BN, BK, WM, WN, VEC>(g);
}

// ... and the reduction by whichever of a tile's K slices finishes last (round 3): every workgroup publishes its partial
// tile (agent-scope release), takes a ticket on the tile's counter, and the last ticket sums the SK partial tiles in
// slice order -- the order splitk_reduce_kernel uses, so the result is bit-identical to the two-launch path -- applies
// the epilogue and re-arms the counter.  One launch less per long-K projection of a decoding step; nobody waits.
template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, (WM * WN <= 64 * 64) ? 2 : 1) gemm_nt_splitk_fused_kernel(GemmArgs g, float* __restrict__ ws,
                                                                                           int kper, int* __restrict__ cnt) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  __shared__ int ticket;
  const int z = blockIdx.z, SK = gridDim.z;
  GemmArgs p = g;
  p.A += (size_t)z * kper;
  p.W += (size_t)z * kper;
  p.K = (g.K - z * kper) < kper ? (g.K - z * kper) : kper;
  p.C = ws + (size_t)z * g.M * g.N;
  p.ldc = g.N;
  p.bias = nullptr;
  p.R = nullptr;
  p.act = SBK_ACT_NONE;
  p.alpha = 1.0f;
  p.seq_len = nullptr;
  int bx = 0, by = 0;
  gemm_nt_tile<BM, BN, BK, WM, WN, VEC>(p, &bx, &by);
  sbk::vm_drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    sbk::release_agent();
    ticket = sbk::atomic_add_agent(cnt + by * gridDim.x + bx, 1);
  }
  __syncthreads();
  if (sbk::uniform(ticket) != SK - 1) return;
  if (threadIdx.x == 0) {
    sbk::acquire_agent();
    sbk::atomic_store_agent(cnt + by * gridDim.x + bx, 0);  // re-armed for the next launch on this stream
  }
  __syncthreads();
  const size_t total = (size_t)g.M * g.N;
  const int m0 = by * BM, n0 = bx * BN;
  for (int e = threadIdx.x; e < BM * BN; e += NT) {
    const int row = m0 + e / BN, col = n0 + e % BN;
    if (row >= g.M || col >= g.N) continue;
    const size_t i = (size_t)row * g.N + col;
    float acc = ws[i];
    for (int ks = 1; ks < SK; ++ks) acc += ws[(size_t)ks * total + i];
    float v = apply_act(acc + (g.bias ? g.bias[col] : 0.0f), g.act) * g.alpha;
    if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
    if (g.R) v += g.R[(size_t)row * g.ldr + col];
    g.C[(size_t)row * g.ldc + col] = v;
  }
}

// Same tiling with 16-byte LDS traffic.  The two k-slices of v_mfma_f32_32x32x2 need not be neighbours in
// memory: any pairing of k values is a valid contraction as long as A and W use the same one.  Lane half
// `lk` therefore owns the contiguous run [lk*BK/2, (lk+1)*BK/2) of a K tile and MFMA number j multiplies
// k = j (lanes 0-31) with k = BK/2 + j (lanes 32-63): every operand fetch is a ds_read_b128 of four
// consecutive k (4x fewer LDS instructions than scalar reads at pitch BK+1) and every panel store a
// ds_write_b128.  Pitch BK+4 floats: the 16-lane groups of a b128 read (MI355X_MICROARCH.md, LDS table) land
// on 16 distinct 4-bank groups because (BK+4)/4 is odd; the 8-lane groups of a b128 write cover one row.
template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, (WM * WN <= 64 * 64) ? 2 : 1) gemm_nt_v4_kernel(GemmArgs g) {
  constexpr int WAVES_N = BN / WN;
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(((BK + 4) / 4) % 2 == 1 && BK % 8 == 0, "pitch (BK+4)/4 must be odd");
  __shared__ __attribute__((aligned(16))) float As[BM][BK + 4];
  __shared__ __attribute__((aligned(16))) float Ws[BN][BK + 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  int bx = blockIdx.x, by = blockIdx.y;
  {  // XCD-aware tile order, as in gemm_nt_kernel
    const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int id = by * gx + bx;
    if (nwg % 8 == 0) {
      const int swz = (id % 8) * (nwg / 8) + id / 8;
      bx = swz % gx;
      by = swz / gx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int lrow = lane & 31, lk = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  PanelStage<BM, BK, NT> pa;
  PanelStage<BN, BK, NT> pw;
  pa.template fetch<VEC>(g.A, g.lda, m0, g.M, 0, g.K, tid);
  pw.template fetch<VEC>(g.W, g.ldw, n0, g.N, 0, g.K, tid);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    pa.commit_vec(As, tid);
    pw.commit_vec(Ws, tid);
    __syncthreads();
    if (k0 + BK < g.K) {  // next K tile: loads fly while this tile is multiplied
      pa.template fetch<VEC>(g.A, g.lda, m0, g.M, k0 + BK, g.K, tid);
      pw.template fetch<VEC>(g.W, g.ldw, n0, g.N, k0 + BK, g.K, tid);
    }
#pragma unroll
    for (int kv = 0; kv < BK / 2; kv += 4) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(&As[wm0 + i * 32 + lrow][lk * (BK / 2) + kv]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(&Ws[wn0 + j * 32 + lrow][lk * (BK / 2) + kv]);
      // k-major order: consecutive MFMAs go to DIFFERENT accumulators (a same-accumulator pair with anything
      // scheduled in between stalls, MI355X_MICROARCH.md per-instruction table)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
            const float bw = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
            acc[i][j] = sbk::mfma_32x32x2(av, bw, acc[i][j]);
          }
    }
    __syncthreads();
  }

  // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + j * 32 + lrow;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        float v = apply_act(acc[i][j][r] + bv, g.act) * g.alpha;
        if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
        if (g.R) v += g.R[(size_t)row * g.ldr + col];
        g.C[(size_t)row * g.ldc + col] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Persistent GEMM on LDS-DMA panels with a stream-K tail (the encoder's contractions: M = frames of a batch, 4-24 K
// rows, i.e. 100-3000 tiles of 128x128 -- tile counts that fill 256 CUs badly when every workgroup takes whole tiles).
//
// The launch has a FIXED number of workgroups G = 8 W (W per XCD, two per CU).  XCD x (= workgroup id % 8, observed)
// owns the contiguous tile range [T x/8, T (x+1)/8) in row-major tile order; its workgroup j takes
//   * whole tiles  t0 + r W + j,  r = 0 .. T_x / W - 1   (at any moment the XCD's W workgroups multiply W CONSECUTIVE
//     tiles: neighbours share the A row panel and every W column panel through that XCD's L2 -- a contiguous range per
//     workgroup instead was measured at a 29 % L2 hit rate and 3.8x the fabric traffic, profiles/r03_*), then
//   * its share of the XCD's T_x % W LEFTOVER tiles, stream-K style: the leftover (tile, 32-deep K tile) units are cut
//     into W equal contiguous ranges, so the tail costs (T_x % W) / W of a tile time instead of a whole one.
// A leftover tile whose K range is cut is finished by whichever of its workgroups arrives last: each writes its partial
// accumulators to its own slab, publishes (agent-scope release, MI355X_MICROARCH.md "Workgroup dispatch ... visibility")
// and takes a ticket on the tile's counter; the last ticket reads ALL the tile's slabs in K order (fixed summation
// order => run-to-run deterministic), applies the epilogue and re-arms the counter.  No workgroup ever waits for
// another one, so nothing depends on residency or dispatch order.  Workgroups of the upper half of an XCD run their
// tail share FIRST (knob 23): the two workgroups of a CU are then half a tile apart and do not sit in their epilogues
// (no MFMA) at the same time.
//
// Panels: global_load_lds_dwordx4 straight into a double-buffered LDS image [stage][A 128 rows | W 128 rows][32 k]
// (no staging registers, no ds_write pass, ONE barrier per K tile).  The LDS image of an LDS-DMA is lane-linear, so
// the bank swizzle is applied to the global SOURCE address: the 16-byte slot s of row r lands in slot s ^ ((r>>1)&7),
// and the MFMA operand fetch (ds_read_b128 of four consecutive k, lane half h of row r reads slot (2g+h) ^ ((r>>1)&7))
// is conflict-free for the 16-lane groups of a b128 read (SQ_LDS_BANK_CONFLICT = 0 measured).  Lanes 0-31 feed
// k = 8g+e, lanes 32-63 k = 8g+4+e of MFMA e of group g (any pairing of the two k slices is a valid contraction as long
// as A and W use the same one).
struct SkArgs {
  GemmArgs g;
  float* slabs;  // [2 * G][128 * 128] partial tiles (slot 0: the workgroup's segment that does not start a tile; 1: the one that does)
  int* cnt;      // [tiles] arrival tickets, zero between launches
  int tiles_n, tiles, KT;
  int stagger;   // upper half of each XCD's workgroups runs the tail share first
  int noload;    // measurement only (knob 22): panels are loaded once per workgroup (wrong results, MFMA/LDS ceiling)
};

// BT: tile edge (128: four waves of 64x64, the encoder shapes; 64: four waves of 32x32, 32 KB of LDS -- the decode-step
// shapes of 640-1 280 rows, where 128-wide tiles leave most CUs without one)
// IL: the LDS-DMA pieces of the next K tile are issued BETWEEN the MFMA groups of the current one (one A piece and one W
// piece per group of 4 k) instead of in one block in front of them: a piece keeps its wave's issue port for ~60-150
// cycles, which a 64-cycle fp32 MFMA in flight covers -- in one block the eight pieces leave the matrix pipe of that
// wave empty for ~1 000 cycles per K tile (the kernel's time was the SUM of its no-load and load-only times).
//
// X3: the fp32 contraction on the bf16 matrix pipe (sbk_gemm_nt_f32x3).  v_mfma_f32_32x32x16_bf16 delivers 16x the
// flops of v_mfma_f32_32x32x2_f32 per cycle, and an fp32 number is EXACTLY the sum of three bf16 numbers (hi = bf16(x),
// mid = bf16(x - hi), lo = x - hi - mid, round to nearest even: the remainders are exact in fp32 and the last one has
// at most 7 significant bits).  a.w = sum of nine partial products; the six of relative size >= 2^-17 are kept (hi.hi,
// hi.mid, mid.hi, hi.lo, lo.hi, mid.mid: exact products, fp32 accumulation on the matrix core), the three dropped ones
// are <= 2^-26 |a||w| each and of either sign (rounded pieces: truncated ones would all carry the product's sign and
// add up), i.e. below the rounding error of ONE fp32 multiply-add -- measured against fp64 the result
// is as close as the fp32 MFMA chain's (tests/test_kernels.py::test_gemm_f32x3).  Six bf16 MFMAs replace sixteen
// fp32-MFMA-equivalents: a 2.67x higher ceiling (2.5 PF/s / 6 = 417 TF/s fp32-equivalent) for the same fp32 result.
// A stays fp32 in HBM and in LDS (same LDS-DMA image as the fp32 kernel) and is cut into its three pieces in registers
// after the operand fetch (two ds_read_b128 = 8 consecutive k of a row = one MFMA operand per piece; ~5 VALU
// instructions per element, which the other wave of the SIMD runs under this wave's MFMAs).  W arrives pre-split
// (sbk_split_bf16x3: [N][K/32][3 pieces][32 k] bf16, 192 contiguous bytes per row and K tile) and lands in LDS as
// [128 rows][3 pieces][4 slots of 8 k]; row pitch 192 B, slot s of row r at s ^ ((r>>2)&3) => the 16-lane groups of a
// ds_read_b128 touch 16 distinct 4-bank groups.
// MEAS (measurement builds of the X3 loop, wrong results): 2 = no operand split, 4 = the hi.hi products only
// (Tried and removed: the X3 panels through registers -- global_load_dwordx4 in front of the MFMA loop, ds_write_b128 behind it --
// instead of by LDS-DMA, whose pieces keep a wave's issue port for 60-180 cycles each.  It needs 40 staging registers: at the
// 256-register budget of two workgroups per CU hipcc spills them, at 512 it parks the accumulators in AGPRs and copies them per
// K tile: 415 us where the LDS-DMA kernel takes 170, profiles/r03_f32x3_sweep.log "g1256".)
template <int BT, bool IL, bool X3, int MEAS = 0>
__global__ void __launch_bounds__(256, 2) gemm_nt_sk_kernel(SkArgs s) {
  static_assert(!X3 || (BT == 128 && !IL), "the split-operand variant: 128-wide tiles, panel loads in one block");
  constexpr int BK = 32, PANEL = BT * BK, WPITCH = X3 ? 48 : BK, WPANEL = BT * WPITCH, STAGE = PANEL + WPANEL;  // floats
  constexpr int TS = BT / 64, WT = BT / 2, LI = BT / 32;      // 32x32 sub-tiles per wave and dimension, wave tile edge, loader instructions per wave and panel
  // the wave's sub-tiles: TM x TN of 32 x 32.  X3: one row block x four column blocks (a wave = 32 rows of the tile, all
  // its 128 columns): every A element is fetched and split by exactly ONE wave (2 x 2 sub-tiles split it twice)
  constexpr int TM = X3 ? 1 : TS, TN = X3 ? 4 : TS;
  constexpr int LIW = X3 ? BT * 3 / 64 : LI;                  // ... of the W panel (X3: 12 slots of 16 B per row)
  SBK_DYN_LDS(float, lds);  // [2][STAGE] + the ticket word (ONE LDS object: a second one de-pipelines the LDS-DMA loop)
  // kernel arguments into registers (a by-value struct whose address is taken is copied to scratch)
  const float* const gA = s.g.A;
  const float* const gW = s.g.W;
  const float* const gbias = s.g.bias;
  const float* const gR = s.g.R;
  float* const gC = s.g.C;
  const int lda = s.g.lda, ldw = s.g.ldw, ldr = s.g.ldr, ldc = s.g.ldc, M = s.g.M, N = s.g.N, act = s.g.act;
  const float alpha = s.g.alpha;
  const int32_t* const seq_len = s.g.seq_len;
  const int rows_per_seq = s.g.rows_per_seq;
  float* const slabs = s.slabs;
  int* const cnt = s.cnt;
  const int tiles_n = s.tiles_n, KT = s.KT, noload = s.noload;

  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int wm0 = X3 ? wave * 32 : (wave >> 1) * WT, wn0 = X3 ? 0 : (wave & 1) * WT;
  const int lrow = lane & 31, half = lane >> 5, sw = (lrow >> 1) & 7;
  // ---- this workgroup's segments: whole tiles of the XCD's range, then (or first) its share of the leftover tiles
  const int W = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;  // gridDim.x is a multiple of 8
  const int t0 = (int)((long)s.tiles * x / 8), t1 = (int)((long)s.tiles * (x + 1) / 8);
  const int nfull = sbk::uniform((t1 - t0) / W), R = (t1 - t0) - nfull * W, tb = t0 + nfull * W;
  const int UT = R * KT, ubase = sbk::uniform(UT / W), urem = UT - ubase * W;
  const int q0 = j * ubase + min(j, urem), q1 = q0 + ubase + (j < urem ? 1 : 0);
  int nt = 0, tileA = 0, loA = 0, hiA = 0, hiB = 0;
  if (q1 > q0) {
    const int ta = sbk::uniform(q0 / KT);
    tileA = tb + ta;
    loA = q0 - ta * KT;
    hiA = min(KT, loA + (q1 - q0));
    hiB = q1 - (ta + 1) * KT;  // > 0: the range runs on into the next leftover tile
    nt = hiB > 0 ? 2 : 1;
  }
  const int nseg_wg = nfull + nt;
  if (nseg_wg == 0) return;
  const bool tail_first = s.stagger && j >= (W >> 1);
  auto seg_get = [&](int sidx, int& tile, int& lo, int& hi) SBK_INLINE_LAMBDA {
    const int d = tail_first ? sidx - nt : sidx;
    if (d >= 0 && d < nfull) {
      tile = t0 + d * W + j;
      lo = 0;
      hi = KT;
    } else if ((tail_first ? sidx : sidx - nfull) == 0) {
      tile = tileA;
      lo = loA;
      hi = hiA;
    } else {
      tile = tileA + 1;
      lo = 0;
      hi = hiB;
    }
  };
  // workgroup (index within the XCD) that owns leftover unit q
  auto owner = [&](int q) SBK_INLINE_LAMBDA {
    const int big = urem * (ubase + 1);
    return sbk::uniform(q < big ? q / (ubase + 1) : urem + (q - big) / max(ubase, 1));
  };
  const int p = x * W + j;  // slab owner id

  // loader geometry: wave-instruction i of this wave covers rows (wave*4+i)*8 .. +7 of a panel, 8 slots of 16 B each
  int lrw[LI], lsl[LI];
#pragma unroll
  for (int i = 0; i < LI; ++i) {
    lrw[i] = (wave * LI + i) * 8 + (lane >> 3);
    lsl[i] = ((lane & 7) ^ ((lrw[i] >> 1) & 7)) * 4;  // source k offset (floats) of the slot this lane fills
  }
  // X3: the W panel is [BT rows][12 slots]; lane-linear slot q of the image = row q / 12, position q % 12 = piece * 4 +
  // (logical slot ^ ((row>>2)&3)); the source is the pre-split matrix [N][KT][3][32 bf16] addressed in float units
  int wrw[LIW], wsl[LIW];
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < LIW; ++i) {
      const int q = (wave * LIW + i) * 64 + lane;
      wrw[i] = q / 12;
      const int pos = q - wrw[i] * 12;
      wsl[i] = (pos >> 2) * 16 + (((pos & 3) ^ ((wrw[i] >> 2) & 3)) * 4);
    }
  }
  const float* ap[LI];
  const float* wp[LIW];
  auto setup = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * BT, n0 = (tile % tiles_n) * BT;
#pragma unroll
    for (int i = 0; i < LI; ++i)  // rows past the matrix re-read its last row (their outputs are never stored)
      ap[i] = gA + (size_t)min(m0 + lrw[i], M - 1) * lda + lsl[i];
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < LIW; ++i) wp[i] = gW + (size_t)min(n0 + wrw[i], N - 1) * (KT * 48) + wsl[i];
    } else {
#pragma unroll
      for (int i = 0; i < LI; ++i) wp[i] = gW + (size_t)min(n0 + lrw[i], N - 1) * ldw + lsl[i];
    }
  };
  auto issue = [&](int kt, int stage) SBK_INLINE_LAMBDA {
    float* base = lds + stage * STAGE + (wave * LI) * 256;
#pragma unroll
    for (int i = 0; i < LI; ++i) sbk::glds16(ap[i] + kt * BK, base + i * 256);
    float* wbase = lds + stage * STAGE + PANEL + (wave * LIW) * 256;
#pragma unroll
    for (int i = 0; i < LIW; ++i) sbk::glds16(wp[i] + kt * WPITCH, wbase + i * 256);
  };

  f32x16 acc[TM][TN];
  auto zero = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };
  auto compute = [&](int stage, bool fly, int nkt) SBK_INLINE_LAMBDA {
    const float* As = lds + stage * STAGE + (wm0 + lrow) * BK;
    const float* Ws = lds + stage * STAGE + PANEL + (wn0 + lrow) * WPITCH;
    if constexpr (X3) {
      const int wsw = (lrow >> 2) & 3;
#pragma unroll
      for (int gk = 0; gk < 2; ++gk) {  // 16 k per step: lanes 0-31 supply k = 16 gk .. +7, lanes 32-63 the next eight
        sbk::bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float4 x0 = *reinterpret_cast<const float4*>(As + i * 32 * BK + ((4 * gk + 2 * half) ^ sw) * 4);
          const float4 x1 = *reinterpret_cast<const float4*>(As + i * 32 * BK + ((4 * gk + 2 * half + 1) ^ sw) * 4);
          const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          unsigned h[4], m[4], l[4];
          if constexpr ((MEAS & 2) != 0) {  // measurement only: no operand split (wrong results)
#pragma unroll
            for (int p = 0; p < 4; ++p) h[p] = m[p] = l[p] = __float_as_uint(x[2 * p]) ^ __float_as_uint(x[2 * p + 1]);
          } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {  // x = hi + mid + lo exactly: 8 significand bits each, remainders exact in fp32
              h[p] = sbk::bf16_pair(x[2 * p], x[2 * p + 1]);
              const float r0 = x[2 * p] - __uint_as_float(h[p] << 16), r1 = x[2 * p + 1] - __uint_as_float(h[p] & 0xffff0000u);
              m[p] = sbk::bf16_pair(r0, r1);
              l[p] = sbk::bf16_pair(r0 - __uint_as_float(m[p] << 16), r1 - __uint_as_float(m[p] & 0xffff0000u));
            }
          }
          ah[i] = sbk::bf16x8_from_words(h[0], h[1], h[2], h[3]);
          am[i] = sbk::bf16x8_from_words(m[0], m[1], m[2], m[3]);
          al[i] = sbk::bf16x8_from_words(l[0], l[1], l[2], l[3]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float* wr = Ws + j * 32 * WPITCH + ((2 * gk + half) ^ wsw) * 4;
          bh[j] = *reinterpret_cast<const sbk::bf16x8*>(wr);
          bm[j] = *reinterpret_cast<const sbk::bf16x8*>(wr + 16);
          bl[j] = *reinterpret_cast<const sbk::bf16x8*>(wr + 32);
        }
        // smallest terms first; consecutive MFMAs go to different accumulators.  W is the FIRST operand: the wave
        // computes (W tile) . (A tile)^T, so a lane owns one row m of C and registers 4g .. 4g+3 hold four consecutive
        // columns n -- the epilogue loads residuals and stores results as 16-byte vectors (a quarter of the store
        // instructions of the column-per-lane layout: the store tail of a tile is issue-bound, MI355X_MICROARCH.md)
        if constexpr ((MEAS & 4) == 0) {  // (measurement only, bit 2: the hi.hi products alone)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bh[j], al[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bl[j], ah[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bm[j], am[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bh[j], am[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bm[j], ah[i], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bh[j], ah[i], acc[i][j]);
      }
      return;
    }
    float* nbase = lds + (stage ^ 1) * STAGE + (wave * LI) * 256;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      const int slot = ((2 * gk + half) ^ sw) * 4;
      float4 a[TS], b[TS];
#pragma unroll
      for (int i = 0; i < TS; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * BK + slot);
#pragma unroll
      for (int j = 0; j < TS; ++j) b[j] = *reinterpret_cast<const float4*>(Ws + j * 32 * BK + slot);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
          for (int j = 0; j < TS; ++j) {
            const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
            const float bw = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
            acc[i][j] = sbk::mfma_32x32x2(av, bw, acc[i][j]);
          }
      if constexpr (IL) {
        if (fly && gk < LI) {  // (uniform) behind this group's MFMAs
          sbk::glds16(ap[gk] + nkt * BK, nbase + gk * 256);
          sbk::glds16(wp[gk] + nkt * BK, nbase + PANEL + gk * 256);
        }
      }
    }
  };
  auto epilogue = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * BT, n0 = (tile % tiles_n) * BT;
    const bool interior = m0 + BT <= M && n0 + BT <= N;  // uniform: no per-element predicates
    if constexpr (X3) {  // transposed accumulators: lane = row m0 + wm0 + lrow, register 4g+e of block j = column j*32 + 8g + 4*half + e
      const int row = m0 + wm0 + lrow, rowc = min(row, M - 1);
      const bool row_ok = interior || row < M;
      const bool masked = seq_len && (rowc % rows_per_seq) >= seq_len[rowc / rows_per_seq];
      const float ra = masked ? 0.0f : alpha;
      float* crow = gC + (size_t)rowc * ldc;
      const float* rrow = gR ? gR + (size_t)rowc * ldr : nullptr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float4 bv[4], rv[4];
        bool ok[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // (N % 4 == 0: a vector is inside the matrix or outside as a whole)
          const int col = n0 + wn0 + j * 32 + 8 * g + 4 * half;
          ok[g] = row_ok && (interior || col < N);
          bv[g] = (gbias && ok[g]) ? *reinterpret_cast<const float4*>(gbias + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          rv[g] = (rrow && ok[g]) ? *reinterpret_cast<const float4*>(rrow + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[4 * g] = acc[0][j][4 * g] + bv[g].x;
          v[4 * g + 1] = acc[0][j][4 * g + 1] + bv[g].y;
          v[4 * g + 2] = acc[0][j][4 * g + 2] + bv[g].z;
          v[4 * g + 3] = acc[0][j][4 * g + 3] + bv[g].w;
        }
        switch (act) {  // uniform
          case SBK_ACT_SWISH:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            break;
          case SBK_ACT_GELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
            break;
          case SBK_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            break;
          case SBK_ACT_LEAKY_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.01f * v[r];
            break;
          default: break;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + wn0 + j * 32 + 8 * g + 4 * half;
          const float4 o = make_float4(masked ? rv[g].x : v[4 * g] * ra + rv[g].x, masked ? rv[g].y : v[4 * g + 1] * ra + rv[g].y,
                                       masked ? rv[g].z : v[4 * g + 2] * ra + rv[g].z, masked ? rv[g].w : v[4 * g + 3] * ra + rv[g].w);
          if (ok[g]) *reinterpret_cast<float4*>(crow + col) = o;
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 32 + lrow;
      const bool col_ok = interior || col < N;
      const float bv = (gbias && col_ok) ? gbias[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm0 + i * 32 + 4 * half;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
        switch (act) {  // uniform
          case SBK_ACT_SWISH:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            break;
          case SBK_ACT_GELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
            break;
          case SBK_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            break;
          case SBK_ACT_LEAKY_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.01f * v[r];
            break;
          default: break;
        }
        if (seq_len) {  // uniform
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
            v[r] = (row % rows_per_seq) >= seq_len[row / rows_per_seq] ? 0.0f : v[r] * alpha;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= alpha;
        }
        if (interior) {
          if (gR) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += gR[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldr + col];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) gC[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = v[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (col_ok && row < M) {
              if (gR) v[r] += gR[(size_t)row * ldr + col];
              gC[(size_t)row * ldc + col] = v[r];
            }
          }
        }
      }
    }
  };
  // a K range [kt_lo, kt_hi) of `tile` is complete in acc.  The ticket word: behind the stages, or (X3: two workgroups
  // of 2 x 40 KB fill the CU's 160 KB) the first word of the stage that was just multiplied -- every wave is past the
  // barrier behind its last read and the next panels go into it only after this function
  auto finish = [&](int tile, int kt_lo, int kt_hi, int stage_done) SBK_INLINE_LAMBDA {
    int* ticket = reinterpret_cast<int*>(lds + (X3 ? stage_done * STAGE : 2 * STAGE));
    bool store = true;
    if (kt_lo != 0 || kt_hi != KT) {  // partial: publish the slab, take a ticket; the last ticket sums the tile's slabs
      const int p_first = owner((tile - tb) * KT), p_last = owner((tile - tb + 1) * KT - 1);
      const int nseg = p_last - p_first + 1;
      float4* mine = reinterpret_cast<float4*>(slabs + (size_t)(2 * p + (kt_lo == 0 ? 1 : 0)) * (BT * BT));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            mine[((wave * (TM * TN) + i * TN + j) * 4 + r4) * 64 + lane] =
                make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
      sbk::vm_drain();
      __syncthreads();
      if (tid == 0) {
        sbk::release_agent();
        *ticket = sbk::atomic_add_agent(cnt + tile, 1);
      }
      __syncthreads();
      store = sbk::uniform(*ticket) == nseg - 1;
      __syncthreads();  // (the ticket word is rewritten by the next partial tile)
      if (store) {
        if (tid == 0) sbk::acquire_agent();
        __syncthreads();
        zero();
        for (int sgm = 0; sgm < nseg; ++sgm) {  // segment order = K order: the sum does not depend on who arrived last
          const float4* src =
              reinterpret_cast<const float4*>(slabs + (size_t)(2 * (x * W + p_first + sgm) + (sgm == 0 ? 1 : 0)) * (BT * BT));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = src[((wave * (TM * TN) + i * TN + j) * 4 + r4) * 64 + lane];
                acc[i][j][4 * r4] += v.x;
                acc[i][j][4 * r4 + 1] += v.y;
                acc[i][j][4 * r4 + 2] += v.z;
                acc[i][j][4 * r4 + 3] += v.w;
              }
        }
        if (tid == 0) sbk::atomic_store_agent(cnt + tile, 0);  // re-armed for the next launch on this stream
      }
    }
    if (store && !(noload & 8)) epilogue(tile);  // (bit 3, measurement only: no epilogue)
  };

  int sidx = 0, tile, lo, hi, stage = 0;
  seg_get(0, tile, lo, hi);
  int kt = lo;
  setup(tile);
  issue(kt, 0);
  zero();
  sbk::vm_drain();
  __syncthreads();
  for (;;) {
    const bool seg_ends = kt + 1 == hi;
    const bool has_next = !seg_ends || sidx + 1 < nseg_wg;
    int ntile = tile, nlo = lo, nhi = hi, nkt = kt + 1;
    if (seg_ends && has_next) {
      seg_get(sidx + 1, ntile, nlo, nhi);
      nkt = nlo;
    }
    if (has_next) {  // the next unit's panels fly while this one is multiplied
      if (seg_ends) setup(ntile);
      if (!IL && !(noload & 1)) issue(nkt, stage ^ 1);
    }
    compute(stage, has_next && !(noload & 1), nkt);
    sbk::vm_drain();   // this wave's share of the next panels has landed ...
    __syncthreads();   // ... and everybody's; every wave is done reading `stage`
    if (seg_ends) {
      finish(tile, lo, hi, stage);
      zero();
    }
    if (!has_next) break;
    if (seg_ends) {
      ++sidx;
      tile = ntile;
      lo = nlo;
      hi = nhi;
    }
    kt = nkt;
    stage ^= 1;
  }
}

// ---------------------------------------------------------------------------
// Skinny GEMM for the decoder steps (M = beams x utterances, a few hundred rows).
// With so few rows an LDS-tiled workgroup grid cannot fill 256 CUs, and the
// weights (L2/MALL resident) dominate traffic.  Here a workgroup owns one
// (TM*32) x 32 output tile and its 4 waves own 4 K slices of it: operands go
// straight from L2 into registers as 16-byte runs (lane (r, half) reads the
// `half` side of a 32-float chunk of row r; the two k-slices of the 32x32x2
// MFMA are fed from the two halves), double-buffered in registers, no LDS and
// no barrier in the main loop.  The four partial tiles meet in LDS and wave 0
// applies the epilogue (fixed summation order => run-to-run deterministic).
// For long K (FFN2) gridDim.y adds a second, global split whose partial tiles
// are combined by splitk_reduce_kernel.
// ---------------------------------------------------------------------------
// bf16 activations AND bf16 weights (sbk_gemm_nt_bf16a): C = epilogue(A . W^T) with A [M,K] and W [N,K] both bf16 in
// HBM, fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 and / or bf16 output.  With the operands already rounded
// the panels go global -> LDS by LDS-DMA exactly like the fp32 persistent kernel's: a 128-byte LDS row is 64 bf16 (a
// K tile of 64) instead of 32 floats, the same source-side slot swizzle makes the ds_read_b128 operand fetch (8
// consecutive k of one row = one MFMA operand) conflict-free.  The matrix pipe needs a K tile every 512 cycles per
// wave (16x the fp32 rate): NS stages (NS - 1 K tiles in flight, s_waitcnt vmcnt(8 x tiles issued after the one needed)
// -- loads retire in order), one barrier per K tile; default NS = 2 with two workgroups per CU (see launch_bf16dma).
// 256-register budget: with 512 the compiler keeps the accumulators in AGPRs and copies all 64 in and out of VGPRs
// every K tile.  Persistent over whole tiles (XCD-contiguous ranges, the K pipeline runs on across
// tile boundaries and under the epilogue); no K split -- the shapes that take this path have thousands of tiles.
struct Bf16DmaArgs {
  const unsigned short* A;
  const unsigned short* W;
  const float* bias;
  const float* R;      // fp32 residual (optional)
  float* C;            // fp32 output (optional)
  unsigned short* Cb;  // bf16 output (optional): the next contraction's operand
  int lda, ldw, ldr, ldc, ldcb, M, N, K, act;
  float alpha;
  int tiles_n, tiles, KT;
  int mode;  // measurement only (knob 29): 1 = no MFMA work, 2 = no panel loads after the prologue (wrong results)
};

template <int NS>
__global__ void __launch_bounds__(256, 2) gemm_nt_bf16dma_kernel(Bf16DmaArgs s) {
  constexpr int BKF = 32, PANEL = 128 * BKF, STAGE = 2 * PANEL;  // float units (one unit = two bf16)
  SBK_DYN_LDS(float, lds);  // [NS][A 128 rows | W 128 rows][64 bf16]
  const unsigned short* const gA = s.A;
  const unsigned short* const gW = s.W;
  const float* const gbias = s.bias;
  const float* const gR = s.R;
  float* const gC = s.C;
  unsigned short* const gCb = s.Cb;
  const int lda = s.lda, ldw = s.ldw, ldr = s.ldr, ldc = s.ldc, ldcb = s.ldcb, M = s.M, N = s.N, act = s.act;
  const float alpha = s.alpha;
  const int tiles_n = s.tiles_n, KT = s.KT, mode = s.mode;

  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int lrow = lane & 31, half = lane >> 5, sw = (lrow >> 1) & 7;
  // this workgroup's tiles: every W-th tile of the XCD's contiguous range
  const int W = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;  // gridDim.x is a multiple of 8
  const int t0 = (int)((long)s.tiles * x / 8), t1 = (int)((long)s.tiles * (x + 1) / 8);
  const int ntile = sbk::uniform(t0 + j < t1 ? (t1 - t0 - j + W - 1) / W : 0);
  if (ntile == 0) return;
  const int U = ntile * KT;

  int lrw[4], lsl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lrw[i] = (wave * 4 + i) * 8 + (lane >> 3);
    lsl[i] = ((lane & 7) ^ ((lrw[i] >> 1) & 7)) * 8;  // source k offset (bf16 elements) of the 16-byte slot this lane fills
  }
  const unsigned short* ap[4];
  const unsigned short* wp[4];
  auto setup = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // rows past the matrix re-read its last row (their outputs are never stored)
      ap[i] = gA + (size_t)min(m0 + lrw[i], M - 1) * lda + lsl[i];
      wp[i] = gW + (size_t)min(n0 + lrw[i], N - 1) * ldw + lsl[i];
    }
  };
  auto issue = [&](int kt, int stage) SBK_INLINE_LAMBDA {
    float* base = lds + stage * STAGE + (wave * 4) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) sbk::glds16(reinterpret_cast<const float*>(ap[i] + kt * 64), base + i * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) sbk::glds16(reinterpret_cast<const float*>(wp[i] + kt * 64), base + PANEL + i * 256);
  };
  f32x16 acc[2][2];
  auto zero = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;
  };
  auto compute = [&](int stage) SBK_INLINE_LAMBDA {
    const float* As = lds + stage * STAGE + (wm0 + lrow) * BKF;
    const float* Ws = lds + stage * STAGE + PANEL + (wn0 + lrow) * BKF;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {  // 16 k per step: lanes 0-31 supply k = 16 gk .. +7, lanes 32-63 the next eight
      const int slot = ((2 * gk + half) ^ sw) * 4;
      sbk::bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const sbk::bf16x8*>(As + i * 32 * BKF + slot);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) b[jj] = *reinterpret_cast<const sbk::bf16x8*>(Ws + jj * 32 * BKF + slot);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[i][jj] = sbk::mfma_32x32x16_bf16(a[i], b[jj], acc[i][jj]);
    }
  };
  auto epilogue = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    const bool interior = m0 + 128 <= M && n0 + 128 <= N;  // uniform: no per-element predicates
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int col = n0 + wn0 + jj * 32 + lrow;
      const bool col_ok = interior || col < N;
      const float bv = (gbias && col_ok) ? gbias[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rbase = m0 + wm0 + i * 32 + 4 * half;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][jj][r] + bv;
        switch (act) {  // uniform
          case SBK_ACT_SWISH:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            break;
          case SBK_ACT_GELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
            break;
          case SBK_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            break;
          case SBK_ACT_LEAKY_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.01f * v[r];
            break;
          default: break;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (interior || (col_ok && row < M)) {
            float o = v[r] * alpha;
            if (gR) o += gR[(size_t)row * ldr + col];
            if (gC) gC[(size_t)row * ldc + col] = o;
            if (gCb) gCb[(size_t)row * ldcb + col] = sbk::f32_to_bf16(o);
          }
        }
      }
    }
  };

  // ---- the K pipeline over this workgroup's units (tile ordinal, K tile): `issued` units are in flight or landed
  int i_ord = 0, i_kt = 0, issued = 0;
  setup(t0 + j);
  auto issue_next = [&]() SBK_INLINE_LAMBDA {
    if (!(mode & 2) || issued < NS - 1) issue(i_kt, issued % NS);
    ++issued;
    if (++i_kt == KT) {
      i_kt = 0;
      if (++i_ord < ntile) setup(t0 + j + i_ord * W);
    }
  };
  for (int pre = 0; pre < NS - 1 && issued < U; ++pre) issue_next();
  zero();
  int landed = -1, c_ord = 0, c_kt = 0;
  for (int n = 0; n < U; ++n) {
    if (n > landed) {  // unit n's panels: everything this wave issued up to it has landed once at most 8 x (units issued after it) loads are in flight
      const int newer = sbk::uniform(issued - 1 - n);
      if (newer <= 0) {
        sbk::vm_drain();
      } else if (newer == 1) {
        sbk::vm_wait<8>();
      } else if (newer == 2) {
        sbk::vm_wait<16>();
      } else {
        sbk::vm_wait<24>();
      }
      landed = n;
    }
    __syncthreads();  // ... and everybody's share of it; every wave is done with the stage of unit n - 1
    if (issued < U) issue_next();  // into the stage unit n - 1 occupied
    if (!(mode & 1)) compute(n % NS);
    if (++c_kt == KT) {
      epilogue(t0 + j + c_ord * W);
      zero();
      c_kt = 0;
      ++c_ord;
      // the epilogue's own loads / stores are younger than every K tile in flight: the next wait is a full one, after
      // which all of them have landed
      landed = n;
    }
  }
}

// fp8 (OCP e4m3) activations AND weights (sbk_gemm_nt_fp8a), each with one fp32 scale per row: C = epilogue(sa[m] sw[n]
// (A8 . W8^T)) with fp32 accumulation on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales: the 2 x-rate fp8 MFMA of
// gfx950, measured 4 267 TF/s from registers against 2 074 for the bf16 form).  The pipeline is gemm_nt_bf16dma_kernel's
// with bytes for elements: a 128-byte LDS row is 128 fp8 (a K tile of 128), panels by LDS-DMA with the same source-side
// slot swizzle, a lane's MFMA operand (32 consecutive bytes of its row: k block `half` of a 64-deep step) is two
// ds_read_b128 of neighbouring slots -- the same bank behaviour as the bf16 fetch; per K tile and wave 8 MFMAs of 64
// cycles where the bf16 kernel has 16 of 32: the same cadence for twice the K, i.e. half the panel bytes per flop.
// The scales are applied to the accumulators in the epilogue (rows of A: per activation row, written by
// sbk_layernorm_fp8o or a previous call's fp8 output; rows of W: per output channel, sbk_quant_rows_fp8 once per
// weight), so no element inside a row shares its scale with another row -- finer than per-tensor scaling, and free.
// Outputs: fp32 and / or bf16 (the attention kernel's operand) and / or fp8 with a FIXED scale (c8_scale: the hidden
// layer of a feed-forward pair, whose row maxima are not known before the last column tile; e4m3's 2^-9 .. 448 range
// at scale 1 covers GELU / Swish outputs of normalised inputs).
struct Fp8DmaArgs {
  const unsigned char* A;
  const unsigned char* W;
  const float* sa;     // [M] scale of each row of A (null: 1)
  const float* sw;     // [N] scale of each row of W (null: 1)
  const float* bias;
  const float* R;      // fp32 residual (optional)
  float* C;            // fp32 output (optional)
  unsigned short* Cb;  // bf16 output (optional)
  unsigned char* C8;   // fp8 output (optional): e4m3(o / c8_scale)
  float c8_scale;
  int lda, ldw, ldr, ldc, ldcb, ldc8, M, N, K, act;
  float alpha;
  int tiles_n, tiles, KT;
};

__global__ void __launch_bounds__(256, 2) gemm_nt_fp8dma_kernel(Fp8DmaArgs s) {
  constexpr int NS = 2, BKF = 32, PANEL = 128 * BKF, STAGE = 2 * PANEL;  // float units (one unit = four fp8)
  SBK_DYN_LDS(float, lds);  // [NS][A 128 rows | W 128 rows][128 fp8]
  const unsigned char* const gA = s.A;
  const unsigned char* const gW = s.W;
  const float* const gsa = s.sa;
  const float* const gsw = s.sw;
  const float* const gbias = s.bias;
  const float* const gR = s.R;
  float* const gC = s.C;
  unsigned short* const gCb = s.Cb;
  unsigned char* const gC8 = s.C8;
  const int lda = s.lda, ldw = s.ldw, ldr = s.ldr, ldc = s.ldc, ldcb = s.ldcb, ldc8 = s.ldc8, M = s.M, N = s.N, act = s.act;
  const float alpha = s.alpha, c8_inv = 1.0f / s.c8_scale;
  const int tiles_n = s.tiles_n, KT = s.KT;

  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int lrow = lane & 31, half = lane >> 5, sw = (lrow >> 1) & 7;
  const int W = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;  // gridDim.x is a multiple of 8
  const int t0 = (int)((long)s.tiles * x / 8), t1 = (int)((long)s.tiles * (x + 1) / 8);
  const int ntile = sbk::uniform(t0 + j < t1 ? (t1 - t0 - j + W - 1) / W : 0);
  if (ntile == 0) return;
  const int U = ntile * KT;

  int lrw[4], lsl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lrw[i] = (wave * 4 + i) * 8 + (lane >> 3);
    lsl[i] = ((lane & 7) ^ ((lrw[i] >> 1) & 7)) * 16;  // source k offset (bytes) of the 16-byte slot this lane fills
  }
  const unsigned char* ap[4];
  const unsigned char* wp[4];
  auto setup = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // rows past the matrix re-read its last row (their outputs are never stored)
      ap[i] = gA + (size_t)min(m0 + lrw[i], M - 1) * lda + lsl[i];
      wp[i] = gW + (size_t)min(n0 + lrw[i], N - 1) * ldw + lsl[i];
    }
  };
  auto issue = [&](int kt, int stage) SBK_INLINE_LAMBDA {
    float* base = lds + stage * STAGE + (wave * 4) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) sbk::glds16(reinterpret_cast<const float*>(ap[i] + kt * 128), base + i * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) sbk::glds16(reinterpret_cast<const float*>(wp[i] + kt * 128), base + PANEL + i * 256);
  };
  f32x16 acc[2][2];
  auto zero = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;
  };
  auto compute = [&](int stage) SBK_INLINE_LAMBDA {
    const float* As = lds + stage * STAGE + (wm0 + lrow) * BKF;
    const float* Ws = lds + stage * STAGE + PANEL + (wn0 + lrow) * BKF;
#pragma unroll
    for (int gk = 0; gk < 2; ++gk) {  // 64 k per step: lanes 0-31 supply bytes 64 gk .. +31 of their row, lanes 32-63 the next 32
      const int s0 = ((4 * gk + 2 * half) ^ sw) * 4, s1 = ((4 * gk + 2 * half + 1) ^ sw) * 4;
      sbk::i32x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = sbk::i32x8_from_u4(*reinterpret_cast<const uint4*>(As + i * 32 * BKF + s0), *reinterpret_cast<const uint4*>(As + i * 32 * BKF + s1));
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        b[jj] = sbk::i32x8_from_u4(*reinterpret_cast<const uint4*>(Ws + jj * 32 * BKF + s0), *reinterpret_cast<const uint4*>(Ws + jj * 32 * BKF + s1));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[i][jj] = sbk::mfma_32x32x64_fp8(a[i], b[jj], acc[i][jj]);
    }
  };
  auto epilogue = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    const bool interior = m0 + 128 <= M && n0 + 128 <= N;  // uniform: no per-element predicates
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = m0 + wm0 + i * 32 + 4 * half;
      float rs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rs[r] = gsa ? gsa[min(rbase + (r & 3) + 8 * (r >> 2), M - 1)] : 1.0f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int col = n0 + wn0 + jj * 32 + lrow;
        const bool col_ok = interior || col < N;
        const float bv = (gbias && col_ok) ? gbias[col] : 0.0f;
        const float cs = (gsw && col_ok) ? gsw[col] : 1.0f;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][jj][r] * (rs[r] * cs) + bv;
        switch (act) {  // uniform
          case SBK_ACT_SWISH:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            break;
          case SBK_ACT_GELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
            break;
          case SBK_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            break;
          case SBK_ACT_LEAKY_RELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.01f * v[r];
            break;
          default: break;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (interior || (col_ok && row < M)) {
            float o = v[r] * alpha;
            if (gR) o += gR[(size_t)row * ldr + col];
            if (gC) gC[(size_t)row * ldc + col] = o;
            if (gCb) gCb[(size_t)row * ldcb + col] = sbk::f32_to_bf16(o);
            if (gC8) gC8[(size_t)row * ldc8 + col] = (unsigned char)(sbk::f32x2_to_fp8(o * c8_inv, 0.0f) & 0xff);
          }
        }
      }
    }
  };

  // ---- the K pipeline over this workgroup's units (tile ordinal, K tile), as gemm_nt_bf16dma_kernel<2>
  int i_ord = 0, i_kt = 0, issued = 0;
  setup(t0 + j);
  auto issue_next = [&]() SBK_INLINE_LAMBDA {
    issue(i_kt, issued % NS);
    ++issued;
    if (++i_kt == KT) {
      i_kt = 0;
      if (++i_ord < ntile) setup(t0 + j + i_ord * W);
    }
  };
  for (int pre = 0; pre < NS - 1 && issued < U; ++pre) issue_next();
  zero();
  int landed = -1, c_ord = 0, c_kt = 0;
  for (int n = 0; n < U; ++n) {
    if (n > landed) {
      const int newer = sbk::uniform(issued - 1 - n);
      if (newer <= 0) {
        sbk::vm_drain();
      } else {
        sbk::vm_wait<8>();
      }
      landed = n;
    }
    __syncthreads();
    if (issued < U) issue_next();
    compute(n % NS);
    if (++c_kt == KT) {
      epilogue(t0 + j + c_ord * W);
      zero();
      c_kt = 0;
      ++c_ord;
      landed = n;
    }
  }
}

template <int NCH>  // 32-float K chunks fetched per batch (all of them in flight together)
__global__ void __launch_bounds__(256, 2) gemm_skinny_kernel(GemmArgs g, float* __restrict__ partial, int kper,
                                                          int tiles_m, int tiles_n) {
  constexpr int KC = 32;  // floats per row per chunk (16 per lane half)
  __shared__ float red[3][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware tile order: workgroup id = 8*q + x runs on XCD x (observed round-robin), so give XCD x
  // the column tiles nt = x (mod 8): every weight row is then fetched into ONE XCD's L2 only.
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;  // column tiles per XCD (last group may be ragged)
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;  // uniform per workgroup: no barrier is skipped by a subset
  }
  const int r = lane & 31, half = lane >> 5;
  const int ks = blockIdx.y * 4 + wave;
  const int k_begin = min(g.K, ks * kper), k_end = min(g.K, k_begin + kper);
  const float* wrow = g.W + (size_t)min(nt * 32 + r, g.N - 1) * g.ldw + half * (KC / 2);
  const float* arow = g.A + (size_t)min(mt * 32 + r, g.M - 1) * g.lda + half * (KC / 2);

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;

  for (int k0 = k_begin; k0 < k_end; k0 += NCH * KC) {
    float4 a[NCH][4], w[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = min(k0 + c * KC, g.K - KC);  // chunks past the slice end re-read a valid chunk and are skipped below
#pragma unroll
      for (int v = 0; v < 4; ++v) w[c][v] = *reinterpret_cast<const float4*>(wrow + k + 4 * v);
#pragma unroll
      for (int v = 0; v < 4; ++v) a[c][v] = *reinterpret_cast<const float4*>(arow + k + 4 * v);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (k0 + c * KC < k_end) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc = sbk::mfma_32x32x2(a[c][v].x, w[c][v].x, acc);
          acc = sbk::mfma_32x32x2(a[c][v].y, w[c][v].y, acc);
          acc = sbk::mfma_32x32x2(a[c][v].z, w[c][v].z, acc);
          acc = sbk::mfma_32x32x2(a[c][v].w, w[c][v].w, acc);
        }
      }
    }
  }

  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave - 1][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  const int col = nt * 32 + r;
  const bool to_partial = gridDim.y > 1;
  const float bv = (!to_partial && g.bias && col < g.N) ? g.bias[col] : 0.0f;
  float* P = partial + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    const int row = mt * 32 + rr;
    const float sum = ((acc[q] + red[0][rr][r]) + red[1][rr][r]) + red[2][rr][r];
    if (row >= g.M || col >= g.N) continue;
    if (to_partial) {
      P[(size_t)row * g.N + col] = sum;
    } else {
      float v = apply_act(sum + bv, g.act) * g.alpha;
      if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
      if (g.R) v += g.R[(size_t)row * g.ldr + col];
      g.C[(size_t)row * g.ldc + col] = v;
    }
  }
}

// The same contraction when a wave's K slice is exactly ONE fetch batch (kper == NCH*32: K = 512 with the four
// waves, K = 2048 with the four-way global split, ...).  No loop and no predicates: all 8*NCH 16-byte loads of a
// lane are issued back to back (the sched_barrier keeps the compiler from sinking them next to their MFMAs --
// in the looped kernel above it pairs every four MFMAs with a fresh load round trip, which makes a 128-deep
// slice cost ~16 dependent L2 latencies), then the MFMA chain runs off registers.
template <int NCH>
__global__ void __launch_bounds__(256, 2) gemm_skinny_flat_kernel(GemmArgs g, float* __restrict__ partial, int tiles_m,
                                                               int tiles_n) {
  constexpr int KC = 32;
  __shared__ float red[3][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;
  }
  const int r = lane & 31, half = lane >> 5;
  const int k_begin = (blockIdx.y * 4 + wave) * NCH * KC;
  const float* wrow = g.W + (size_t)min(nt * 32 + r, g.N - 1) * g.ldw + half * (KC / 2) + k_begin;
  const float* arow = g.A + (size_t)min(mt * 32 + r, g.M - 1) * g.lda + half * (KC / 2) + k_begin;
  float4 a[NCH][4], w[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int v = 0; v < 4; ++v) w[c][v] = *reinterpret_cast<const float4*>(wrow + c * KC + 4 * v);
#pragma unroll
    for (int v = 0; v < 4; ++v) a[c][v] = *reinterpret_cast<const float4*>(arow + c * KC + 4 * v);
  }
  sbk::sched_fence();
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      acc = sbk::mfma_32x32x2(a[c][v].x, w[c][v].x, acc);
      acc = sbk::mfma_32x32x2(a[c][v].y, w[c][v].y, acc);
      acc = sbk::mfma_32x32x2(a[c][v].z, w[c][v].z, acc);
      acc = sbk::mfma_32x32x2(a[c][v].w, w[c][v].w, acc);
    }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave - 1][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    v[q] = ((acc[q] + red[0][rr][r]) + red[1][rr][r]) + red[2][rr][r];
  }
  if (gridDim.y > 1) {  // partial tile of a global K split: combined by splitk_reduce_kernel
    float* P = partial + (size_t)blockIdx.y * g.M * g.N;
    const int col = nt * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      if (row < g.M && col < g.N) P[(size_t)row * g.N + col] = v[q];
    }
    return;
  }
  tile_epilogue_32x32(g, v, mt, nt, r, half);
}

// 64 x 64 tile of the same scheme for M >= ~600 rows (a 128-utterance batch, or a grouped search over several
// recipe-sized batches): the four waves still split K four ways and hold their whole slice in registers, but each
// owns a 2 x 2 block of 32x32 accumulators.  Per MFMA that halves the bytes pulled from L2 (the 32x32 kernel moves
// 128 KB per workgroup for 64 MFMAs per wave and is bound by that traffic from ~500 workgroups on: measured 2.2 us
// per extra 100 workgroups), and the four accumulator chains are independent, so no MFMA waits on its predecessor.
template <int NCH>
__global__ void __launch_bounds__(256) gemm_skinny_flat64_kernel(GemmArgs g, float* __restrict__ partial, int tiles_m,
                                                                 int tiles_n) {
  constexpr int KC = 32;
  __shared__ float red[4][4][32][33];  // [wave][sub-tile][row][col]: partial tiles of the four K slices
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;
  }
  const int r = lane & 31, half = lane >> 5;
  const int k_begin = (blockIdx.y * 4 + wave) * NCH * KC;
  const float* arow[2];
  const float* wrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = g.A + (size_t)min(mt * 64 + i * 32 + r, g.M - 1) * g.lda + half * (KC / 2) + k_begin;
    wrow[i] = g.W + (size_t)min(nt * 64 + i * 32 + r, g.N - 1) * g.ldw + half * (KC / 2) + k_begin;
  }
  float4 a[2][NCH][4], w[2][NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int v = 0; v < 4; ++v) w[i][c][v] = *reinterpret_cast<const float4*>(wrow[i] + c * KC + 4 * v);
#pragma unroll
      for (int v = 0; v < 4; ++v) a[i][c][v] = *reinterpret_cast<const float4*>(arow[i] + c * KC + 4 * v);
    }
  sbk::sched_fence();
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float av = e == 0 ? a[i][c][v].x : e == 1 ? a[i][c][v].y : e == 2 ? a[i][c][v].z : a[i][c][v].w;
            const float wv = e == 0 ? w[j][c][v].x : e == 1 ? w[j][c][v].y : e == 2 ? w[j][c][v].z : w[j][c][v].w;
            acc[i][j] = sbk::mfma_32x32x2(av, wv, acc[i][j]);
          }
  // every wave publishes its four partial sub-tiles; wave s then owns sub-tile s = 2*i + j (fixed summation order)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) red[wave][2 * i + j][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[i][j][q];
  __syncthreads();
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    v[q] = ((red[0][wave][rr][r] + red[1][wave][rr][r]) + red[2][wave][rr][r]) + red[3][wave][rr][r];
  }
  const int sub_m = mt * 2 + (wave >> 1), sub_n = nt * 2 + (wave & 1);  // this wave's 32x32 tile in 32-row/col units
  if (gridDim.y > 1) {
    float* P = partial + (size_t)blockIdx.y * g.M * g.N;
    const int col = sub_n * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = sub_m * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      if (row < g.M && col < g.N) P[(size_t)row * g.N + col] = v[q];
    }
    return;
  }
  tile_epilogue_32x32(g, v, sub_m, sub_n, r, half);
}

// LayerNorm fused into the skinny GEMM:  C = epilogue( LN(A) . W^T ) for K = NCH*128 (one fetch batch
// per wave, so the workgroup's four waves hold complete rows of A in registers).  gamma/beta are
// pre-folded into the operands by the caller:  Wf[n,k] = W[n,k]*gamma[k],  bf[n] = b[n] + sum_k W[n,k]*beta[k],
// hence  LN(x).W^T + b = rstd * ((x - mean) . Wf^T) + bf.  Row statistics are the two-pass form of
// csrc/norm.hip (mean, then sum of squared deviations), reduced across the waves through LDS.
template <int NCH>
__global__ void __launch_bounds__(256, 2) gemm_skinny_ln_kernel(GemmArgs g, float eps, int tiles_m, int tiles_n) {
  constexpr int KC = 32;
  __shared__ float red[3][32][33];
  __shared__ float stat[2][4][32];
  __shared__ float rstd_s[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;
  }
  const int r = lane & 31, half = lane >> 5;
  const int k_begin = wave * NCH * KC;  // K == 4 * NCH * KC
  const float* wrow = g.W + (size_t)min(nt * 32 + r, g.N - 1) * g.ldw + half * (KC / 2);
  const float* arow = g.A + (size_t)min(mt * 32 + r, g.M - 1) * g.lda + half * (KC / 2);
  float4 a[NCH][4], w[NCH][4];
  // the activation rows first: the row statistics below wait for them only, the weight loads stay in flight
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) a[c][v] = *reinterpret_cast<const float4*>(arow + k_begin + c * KC + 4 * v);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) w[c][v] = *reinterpret_cast<const float4*>(wrow + k_begin + c * KC + 4 * v);
  sbk::sched_fence();
  // mean over the full row: lane partial -> both halves -> the four waves
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) s += (a[c][v].x + a[c][v].y) + (a[c][v].z + a[c][v].w);
  s += sbk::shfl_xor(s, 32);
  if (half == 0) stat[0][wave][r] = s;
  __syncthreads();
  const float mean = ((stat[0][0][r] + stat[0][1][r]) + (stat[0][2][r] + stat[0][3][r])) / (float)g.K;
  float q2 = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      a[c][v].x -= mean; a[c][v].y -= mean; a[c][v].z -= mean; a[c][v].w -= mean;
      q2 += (a[c][v].x * a[c][v].x + a[c][v].y * a[c][v].y) + (a[c][v].z * a[c][v].z + a[c][v].w * a[c][v].w);
    }
  q2 += sbk::shfl_xor(q2, 32);
  if (half == 0) stat[1][wave][r] = q2;

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      acc = sbk::mfma_32x32x2(a[c][v].x, w[c][v].x, acc);
      acc = sbk::mfma_32x32x2(a[c][v].y, w[c][v].y, acc);
      acc = sbk::mfma_32x32x2(a[c][v].z, w[c][v].z, acc);
      acc = sbk::mfma_32x32x2(a[c][v].w, w[c][v].w, acc);
    }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave - 1][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  if (half == 0) {
    const float var = ((stat[1][0][r] + stat[1][1][r]) + (stat[1][2][r] + stat[1][3][r])) / (float)g.K;
    rstd_s[r] = rsqrtf(var + eps);
  }
  sbk::wave_sync();
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    v[q] = (((acc[q] + red[0][rr][r]) + red[1][rr][r]) + red[2][rr][r]) * rstd_s[rr];
  }
  tile_epilogue_32x32(g, v, mt, nt, r, half);
}

// C = epilogue(sum_ks partial[ks]) ; fixed summation order => run-to-run deterministic.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs g, const float* __restrict__ partial, int SK) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / g.N), col = (int)(i % g.N);
    float acc = partial[i];
    for (int ks = 1; ks < SK; ++ks) acc += partial[(size_t)ks * total + i];
    float v = apply_act(acc + (g.bias ? g.bias[col] : 0.0f), g.act) * g.alpha;
    if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
    if (g.R) v += g.R[(size_t)row * g.ldr + col];
    g.C[(size_t)row * g.ldc + col] = v;
  }
}

}  // namespace
namespace sbk {
int g_gemm_vec_lds = 0;  // tuning knob (key 9): 1 = the 16-byte LDS operand variant (gemm_nt_v4_kernel)
}
namespace {
template <int BM, int BN, int BK, int WM, int WN>
int launch_gemm(const GemmArgs& g, bool vec, hipStream_t st) {
  dim3 grid(sbk::cdiv(g.N, BN), sbk::cdiv(g.M, BM));
  dim3 block((BM / WM) * (BN / WN) * 64);
  static const char* kName = (BM == 256 && BN == 128) ? (WM == 128 ? "gemm_nt_256x128w" : "gemm_nt_256x128")
                             : (BM == 128 && BN == 256) ? "gemm_nt_128x256"
                             : BM == 128 ? "gemm_nt_128x128" : (BM == 64 ? "gemm_nt_64x64" : "gemm_nt_32x64");
  sbk::ProfScope prof(kName, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N), st);
  if (!sbk::g_gemm_vec_lds) {  // default: scalar LDS operand reads at pitch BK+1 (measured faster)
    if (vec) {
      SBK_LAUNCH((gemm_nt_kernel<BM, BN, BK, WM, WN, true>), grid, block, 0, st, g);
    } else {
      SBK_LAUNCH((gemm_nt_kernel<BM, BN, BK, WM, WN, false>), grid, block, 0, st, g);
    }
  } else if (vec) {
    SBK_LAUNCH((gemm_nt_v4_kernel<BM, BN, BK, WM, WN, true>), grid, block, 0, st, g);
  } else {
    SBK_LAUNCH((gemm_nt_v4_kernel<BM, BN, BK, WM, WN, false>), grid, block, 0, st, g);
  }
  return sbk::launch_status("sbk_gemm_nt_f32");
}

}  // namespace

namespace sbk {
int g_skinny_nch = 0;  // tuning knob (0 = automatic): K chunks fetched per batch by the skinny kernel
int g_skinny_off = 0;  // tuning knob: 1 = route few-row GEMMs to the LDS-tiled kernels
int g_skinny_looped = 0;  // tuning knob (key 10): 1 = always the looped skinny kernel (the round-1 schedule)
int g_flat64_min_rows = 1 << 30;  // tuning knob (key 11): from this many rows on the register-operand path uses 64x64 tiles
                                  // (off by default: measured slower in situ, 29 vs 23 us at M = 1280)
int g_tiled_splitk = 256;  // tuning knob (key 14): from this many rows on, K >= 2048 shapes take 64x64 LDS tiles with a
                           // 4-way K split instead of the register-operand path (0 = off).  Measured (tools/microbench.py
                           // --ffn2, N = 512, K = 2048): 160 rows 17.9 -> 21.6 us, 320: 24.2 -> 20.1, 1280: 53.0 -> 37.8,
                           // 2560: 94.4 -> 57.9; N = 768, K = 3072 at 1280 rows: 115.7 -> 63.5
// tuning knob (key 36): 1 = the tiled split-K GEMM reduces in the last-arriving workgroup of a tile.  OFF: measured slower in
// the bench (9 815 vs 10 416 audio-s/s with knobs 36 + 37 on / off, profiles/r03_last_arriver_reductions_ab.log): the
// agent-scope release every workgroup needs is an L2 write-back on an 8-XCD part, paid 640 times per launch here
int g_splitk_fused = 0;
int g_tiled_splitk_short = 0;  // tuning knob (key 15): K split of the same kernel for 512 <= K < 2048 (0 = not used)
int g_skinny_reach = 0;   // tuning knob (key 12): 1 = the register-operand path also takes the mid-M shapes that go to
                          // the LDS-tiled kernels by default (M*N >= 1.9 M with K <= 1024)
int g_gemm_tile = 0;   // tuning knob (key 6) for the large-M path: 0 = 128x128, 1 = 256x128 (8 waves of 64x64),
                       // 2 = 128x256 (8 waves), 3 = 256x128 (4 waves of 128x64), 4 = 128x128 with 64-deep K tiles
int gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
            int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st);
int* tile_tickets(hipStream_t st, long tiles);  // this stream's zeroed arrival counters (nullptr: no workspace registered for the stream, or too many tiles)
int sk_route(int M, int N, int K, int* bt = nullptr);  // workgroups (and tile edge) of the persistent kernel for this shape (0: tile-grid / register-operand paths)
// Internal C++ entry shared with the fused pipelines (decoder step, encoder).
// Skinny path: M <= 512 rows, K a multiple of 64, 16-byte aligned rows.  `ws` (optional) holds the
// split-K partials: SK * M * N floats.
int gemm_nt_ws(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
               int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq,
               float* ws, size_t ws_floats, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  // (measured, tools/microbench.py --attn --gemm: from ~1.9 M outputs with a short K the LDS-tiled kernels win:
  //  M=1280 N=1536 41.6 -> 27.6 us, M=640 N=5000 63 -> 46 us; a long K still needs the split of the skinny path)
  const bool big_short = !g_skinny_reach && (long)M * N >= 1900000 && K <= 1024;  // (K = 768: the TransformerLM scorer's projections)
  const bool skinny_ok = !big_short && (M <= 512 || (long)cdiv(M, 128) * cdiv(N, 128) < (g_skinny_reach ? 2048 : 256)) && M <= (g_skinny_reach ? 8192 : 4096) && K % 32 == 0 && lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) && aligned16(W);
  if (!skinny_ok || g_skinny_off || (aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && sk_route(M, N, K) > 0))
    return gemm_nt(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq, st);
  GemmArgs g{A, W, bias, R, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, seq_len, rows_per_seq > 0 ? rows_per_seq : 1};
  // (K = 768, the TransformerLM scorer's projections: 2-way split, 34.4 -> 24.6 us at 1280 rows; K = 512: no gain)
  const int short_sk = K < 2048 && (long)M * N < 1900000 ? (g_tiled_splitk_short ? (K >= 512 ? g_tiled_splitk_short : 0) : (K >= 768 && M >= 1024 ? 2 : 0)) : 0;
  if (g_tiled_splitk && ws && M >= g_tiled_splitk && (K >= 2048 || short_sk) && K % 128 == 0) {  // long K at ~1 K rows: 64x64 LDS tiles, K split 4-way
    const int SK = short_sk ? short_sk : 4, kper = K / SK;
    if ((size_t)SK * M * N <= ws_floats) {
      ProfScope prof("gemm_skinny", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
      dim3 grid(cdiv(N, 64), cdiv(M, 64), SK), block(256);
      int* tickets = g_splitk_fused ? tile_tickets(st, (long)grid.x * grid.y) : nullptr;
      if (tickets) {  // the reduction runs in the last-arriving K slice of every tile
        SBK_LAUNCH((gemm_nt_splitk_fused_kernel<64, 64, 32, 32, 32, true>), grid, block, 0, st, g, ws, kper, tickets);
        return launch_status("gemm_splitk_fused");
      }
      SBK_LAUNCH((gemm_nt_splitk_kernel<64, 64, 32, 32, 32, true>), grid, block, 0, st, g, ws, kper);
      int rc = launch_status("gemm_splitk_tiled");
      if (rc) return rc;
      const size_t total = (size_t)M * N;
      SBK_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, g, (const float*)ws, SK);
      return launch_status("splitk_reduce");
    }
  }
  const int tiles_m = cdiv(M, 32), tiles_n = cdiv(N, 32);
  // a second, global K split when the tile grid alone leaves SIMDs idle (needs `ws` for the partial tiles)
  int SKg = 1;
  if (ws) {
    // (only worth the extra reduce launch for long K: K = 512 slices are 128 deep already)
    while (K / (4 * SKg) > 128 && K % (4 * SKg * 2 * 32) == 0 && tiles_m * tiles_n * SKg < 1024 &&
           (size_t)(SKg * 2) * M * N <= ws_floats)
      SKg *= 2;
  }
  const int kper = cdiv(cdiv(K, 4 * SKg), 32) * 32;
  ProfScope prof("gemm_skinny", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
  dim3 grid(8 * tiles_m * cdiv(tiles_n, 8), SKg), block(256);
  const int nch = g_skinny_nch ? g_skinny_nch : (kper >= 128 ? 4 : (kper >= 64 ? 2 : 1));
  const bool flat = !g_skinny_looped && K == 4 * SKg * kper && (kper == 128 || kper == 64 || kper == 32);
  if (flat && (kper == 128 || kper == 64) && M >= g_flat64_min_rows) {  // 2 x 2 accumulators per wave (see the kernel)
    const int tm64 = cdiv(M, 64), tn64 = cdiv(N, 64);
    dim3 grid64(8 * tm64 * cdiv(tn64, 8), SKg);
    if (kper == 128) {
      SBK_LAUNCH((gemm_skinny_flat64_kernel<4>), grid64, block, 0, st, g, ws, tm64, tn64);
    } else {
      SBK_LAUNCH((gemm_skinny_flat64_kernel<2>), grid64, block, 0, st, g, ws, tm64, tn64);
    }
  } else if (flat) {  // one fetch batch per wave: every load in flight before the first MFMA
    if (kper == 128) {
      SBK_LAUNCH((gemm_skinny_flat_kernel<4>), grid, block, 0, st, g, ws, tiles_m, tiles_n);
    } else if (kper == 64) {
      SBK_LAUNCH((gemm_skinny_flat_kernel<2>), grid, block, 0, st, g, ws, tiles_m, tiles_n);
    } else {
      SBK_LAUNCH((gemm_skinny_flat_kernel<1>), grid, block, 0, st, g, ws, tiles_m, tiles_n);
    }
  } else if (nch >= 4) {
    SBK_LAUNCH((gemm_skinny_kernel<4>), grid, block, 0, st, g, ws, kper, tiles_m, tiles_n);
  } else if (nch >= 2) {
    SBK_LAUNCH((gemm_skinny_kernel<2>), grid, block, 0, st, g, ws, kper, tiles_m, tiles_n);
  } else {
    SBK_LAUNCH((gemm_skinny_kernel<1>), grid, block, 0, st, g, ws, kper, tiles_m, tiles_n);
  }
  int rc = launch_status("gemm_skinny");
  if (rc || SKg == 1) return rc;
  const size_t total = (size_t)M * N;
  SBK_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, g, (const float*)ws, SKg);
  return launch_status("splitk_reduce");
}

// C = epilogue(LN(A) . Wf^T + bf) with gamma/beta pre-folded into (Wf, bf); returns -1 when the shape is
// not eligible (the caller then runs LayerNorm + gemm_nt_ws with the unfolded weights).
int gemm_ln_nt(const float* A, int lda, const float* Wf, int ldw, const float* bf, const float* R, int ldr, float* C,
               int ldc, int M, int N, int K, float eps, int act, float alpha, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  const bool ok = (K == 512 || K == 256 || K == 128) && (M <= 512 || (long)cdiv(M, 128) * cdiv(N, 128) < 256) &&
                  M <= 4096 && (long)M * N < 1900000 && lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) && aligned16(Wf) &&
                  !g_skinny_off;
  if (!ok) return -1;
  GemmArgs g{A, Wf, bf, R, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, nullptr, 1};
  const int tiles_m = cdiv(M, 32), tiles_n = cdiv(N, 32);
  ProfScope prof("gemm_skinny_ln", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
  dim3 grid(8 * tiles_m * cdiv(tiles_n, 8)), block(256);
  if (K == 512) {
    SBK_LAUNCH((gemm_skinny_ln_kernel<4>), grid, block, 0, st, g, eps, tiles_m, tiles_n);
  } else if (K == 256) {
    SBK_LAUNCH((gemm_skinny_ln_kernel<2>), grid, block, 0, st, g, eps, tiles_m, tiles_n);
  } else {
    SBK_LAUNCH((gemm_skinny_ln_kernel<1>), grid, block, 0, st, g, eps, tiles_m, tiles_n);
  }
  return launch_status("gemm_skinny_ln");
}

// ---- stream-K launch: per-stream workspace (slabs + tile tickets), registered by the caller (sbk_stream_workspace_set)
int g_sk_mode = 1;        // tuning knob (key 18): 0 = tile-grid kernels only, 1 = routed by shape (sk_route), 2 = always (tests), 3 = always from 8 tiles on (A/B)
int g_sk_grid = 0;        // tuning knob (key 19): workgroups of a stream-K launch (0 = two per CU)
int g_sk_noload = 0;      // measurement knob (key 22)
int g_sk_stagger = 1;     // tuning knob (key 23): upper half of each XCD's workgroups runs its tail share first
int g_sk_min_rows = 2048;  // tuning knob (key 24): fewer rows than this never take the persistent kernel in routed mode
int g_sk_min_units = 4;   // tuning knob (key 21): fewer units per workgroup than this shrinks the grid
int g_sk_interleave = 0;  // tuning knob (key 30): 1 = the next K tile's LDS-DMA pieces are issued between the MFMA groups
int g_sk64_min_rows = 0;  // tuning knob (key 25): from this many rows on (and below g_sk_min_rows) the 64x64-tile persistent kernel; 0 = off
int g_sk64_units = 16;    // tuning knob (key 26): K units (64x64x32) per workgroup the 64-tile grid is sized for
int g_bf16a_stages = 2;   // tuning knob (key 27): LDS stages of gemm_nt_bf16dma_kernel (2, 3 or 4)
int g_bf16a_grid = 0;     // tuning knob (key 28): its workgroups (0 = as many as fit: two per CU with 2 stages, one with 3 / 4)
int g_bf16a_mode = 0;     // measurement knob (key 29)
int g_x3_grid = 0;        // tuning knob (key 31): workgroups of the split-operand kernel (0 = two per CU from one tile per CU on)
namespace {
constexpr int kSkMaxGrid = 512, kSkMaxGrid64 = 1024, kSkMaxTiles = 1 << 16;  // (both grids fit the same slab area)
int sk_cus();
}
// Workgroups of the persistent kernel for this shape, 0 = the tile-grid kernels.  Measured on MI355X (tools/microbench.py
// --sk, profiles/r03_gemm_persistent_sweep.log): the persistent kernel wins once every workgroup gets about a tile's
// worth of units (T >= W per XCD keeps the leftover share small); two workgroups per CU from ~500 tiles on, one below;
// narrow short-K shapes (N <= 512, K <= 512: four column tiles, 16 K steps per tile -- the epilogue and the partial
// tiles weigh most there) stay on the tile grid below ~400 tiles (in situ -- tools/microbench.py --enc-layer,
// profiles/r03_encoder_in_situ_gemm_routing.log -- the persistent kernel already wins there from 500 tiles on).
int sk_route(int M, int N, int K, int* bt) {
  int unused;
  if (!bt) bt = &unused;
  *bt = 128;
  if (!g_sk_mode || K % 32 != 0 || K < 64) return 0;
  const long T = (long)cdiv(M, 128) * cdiv(N, 128), U = T * (K / 32);
  const int cus = sk_cus();
  int G = g_sk_grid;
  if (g_sk_mode == 1 && M < g_sk_min_rows) {
    // decode-step shapes (a few hundred to ~1 300 rows): 64x64 tiles, one workgroup per g_sk64_units K units -- a whole
    // tile at K = 512, a quarter of one at K = 2 048 (split K through the slabs, reduced by the last arriver)
    if (!g_sk64_min_rows || M < g_sk64_min_rows) return 0;
    *bt = 64;
    const long U64 = (long)cdiv(M, 64) * cdiv(N, 64) * (K / 32);
    if (!G) G = (int)std::min<long>(kSkMaxGrid64, U64 / std::max(1, g_sk64_units));
    return G >= 8 ? (G / 8) * 8 : 0;
  }
  if (g_sk_mode == 1) {
    const bool narrow_short = N <= 512 && K <= 512;
    if (narrow_short ? 2 * T < 3L * cus : U < 16L * cus) return 0;
    if (!G) G = U >= 32L * cus ? 2 * cus : cus;
  } else {
    if (g_sk_mode == 3 && T < 8) return 0;
    if (!G) G = 2 * cus;
    if (U / g_sk_min_units < G) G = (int)(U / g_sk_min_units);  // short launches: fewer, longer ranges
  }
  if (G > kSkMaxGrid) G = kSkMaxGrid;
  return G >= 8 ? (G / 8) * 8 : 8;  // W workgroups on each of the 8 XCDs
}
namespace {
struct SkWorkspace {
  float* slabs;
  int* cnt;
};
// Stream workspaces are CALLER-OWNED device memory (sbk_stream_workspace_set, include/sbk.h): the library allocates
// nothing.  One per (device, stream): launches of one stream are ordered, so they can share the slabs and the tickets.
std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, SkWorkspace> g_sk_ws;
int g_sk_cus[64] = {};

int cur_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}

// two partial-tile slabs per workgroup: 512 workgroups x 128 x 128 (this file's kernels) or 256 x 256 x 256 (gemm_x3p.hip)
constexpr size_t kSkSlabBytes = (size_t)2 * 256 * 256 * 256 * sizeof(float);
static_assert(kSkSlabBytes >= (size_t)2 * kSkMaxGrid * 128 * 128 * sizeof(float), "slab area");
constexpr size_t kSkTicketBytes = (size_t)kSkMaxTiles * sizeof(int);

// the stream's workspace; false when the caller has registered none for it (the tile-grid kernels run instead)
bool sk_workspace(hipStream_t st, SkWorkspace* out) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  auto it = g_sk_ws.find(std::make_pair(cur_device(), st));
  if (it == g_sk_ws.end()) return false;
  *out = it->second;
  return true;
}

}  // namespace
int* tile_tickets(hipStream_t st, long tiles) {
  SkWorkspace w;
  if (tiles > kSkMaxTiles || !sk_workspace(st, &w)) return nullptr;
  return w.cnt;
}
bool stream_ws(hipStream_t st, float** slabs, int** cnt) {  // (gemm_x3p.hip)
  SkWorkspace w;
  if (!sk_workspace(st, &w)) return false;
  *slabs = w.slabs;
  *cnt = w.cnt;
  return true;
}
namespace {

int sk_cus() {
  const int dev = cur_device() & 63;
  if (!g_sk_cus[dev]) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    g_sk_cus[dev] = cus > 0 ? cus : 256;
  }
  return g_sk_cus[dev];
}

}  // namespace
int device_cus() { return sk_cus(); }
namespace {
int launch_sk(const GemmArgs& g, int G, int bt, hipStream_t st, bool x3 = false) {
  SkArgs s;
  s.g = g;
  s.tiles_n = cdiv(g.N, bt);
  s.tiles = cdiv(g.M, bt) * s.tiles_n;
  s.KT = g.K / 32;
  if (s.tiles > kSkMaxTiles || (long)s.tiles * s.KT > (1L << 30)) return -1;
  SkWorkspace w;
  if (!sk_workspace(st, &w)) return -1;
  s.slabs = w.slabs;
  s.cnt = w.cnt;
  s.stagger = g_sk_stagger;
  s.noload = g_sk_noload;
  const size_t lds = x3 ? (size_t)2 * (128 * 32 + 128 * 48) * sizeof(float) : (size_t)(2 * 2 * bt * 32 + 4) * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, false, false>), (size_t)(2 * 2 * 128 * 32 + 4) * sizeof(float));
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, true, false>), (size_t)(2 * 2 * 128 * 32 + 4) * sizeof(float));
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, false, true>), (size_t)2 * (128 * 32 + 128 * 48) * sizeof(float));
    once = true;
  }
  const double flops = 2.0 * g.M * g.N * g.K, bytes = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
  if (x3) {
    // (fp32-equivalent flops: the six bf16 partial products of an element pair count as ONE multiply-add)
    ProfScope prof("gemm_nt_f32x3", flops, bytes + 2.0 * (double)g.N * g.K, st);
    const int meas = (g_sk_noload >> 1) & 3;
    s.noload = g_sk_noload & 9;
    if (meas == 0) {
      SBK_LAUNCH((gemm_nt_sk_kernel<128, false, true>), dim3((unsigned)G), dim3(256), lds, st, s);
    } else {
      static bool once_meas = false;
      if (!once_meas) {
        (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, false, true, 2>), lds);
        (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, false, true, 4>), lds);
        (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, false, true, 6>), lds);
        once_meas = true;
      }
      if (meas == 1) {
        SBK_LAUNCH((gemm_nt_sk_kernel<128, false, true, 2>), dim3((unsigned)G), dim3(256), lds, st, s);
      } else if (meas == 2) {
        SBK_LAUNCH((gemm_nt_sk_kernel<128, false, true, 4>), dim3((unsigned)G), dim3(256), lds, st, s);
      } else {
        SBK_LAUNCH((gemm_nt_sk_kernel<128, false, true, 6>), dim3((unsigned)G), dim3(256), lds, st, s);
      }
    }
  } else if (bt == 64) {
    ProfScope prof("gemm_nt_persistent64", flops, bytes, st);
    SBK_LAUNCH((gemm_nt_sk_kernel<64, false, false>), dim3((unsigned)G), dim3(256), lds, st, s);
  } else {
    ProfScope prof("gemm_nt_persistent", flops, bytes, st);
    if (g_sk_interleave) {
      SBK_LAUNCH((gemm_nt_sk_kernel<128, true, false>), dim3((unsigned)G), dim3(256), lds, st, s);
    } else {
      SBK_LAUNCH((gemm_nt_sk_kernel<128, false, false>), dim3((unsigned)G), dim3(256), lds, st, s);
    }
  }
  return launch_status("sbk_gemm_nt_f32 (stream-K)");
}
}  // namespace

// internal callers (the search's memory / CTC / vocabulary projections): from 1 024 rows and 192 tiles on (M = 1 280,
// N = 5 000: 64.7 vs 87.9 us; M = 640: 44.9 vs 47.4, left on the tile grid)
int g_x3_route_rows = 1024, g_x3_route_tiles = 192;  // knobs 34 / 35 (tests lower them to reach these calls with small models)
bool x3_routed(int M, int N, int K) {
  return M >= g_x3_route_rows && K % 32 == 0 && K >= 64 && N % 4 == 0 && (long)cdiv(M, 128) * cdiv(N, 128) >= g_x3_route_tiles;
}

// fp32 contraction with a pre-split W (sbk_split_bf16x3) on the bf16 matrix pipe; -1: no workspace for this stream
int gemm_nt_x3(const float* A, int lda, const uint16_t* W3, const float* bias, const float* R, int ldr, float* C, int ldc,
               int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  GemmArgs g{A, reinterpret_cast<const float*>(W3), bias, R, C, lda, 0, ldr, ldc, M, N, K, act, alpha, seq_len,
             rows_per_seq > 0 ? rows_per_seq : 1};
  const long T = (long)cdiv(M, 128) * cdiv(N, 128), U = T * (K / 32);
  const int cus = sk_cus();
  // Measured on MI355X (tools/microbench.py --x3, profiles/r03_f32x3_sweep.log): two workgroups per CU from two tiles per
  // CU on, one below (M = 4 032: N = 1 024 31 vs 50 us, N = 1 536 59 vs 69 us with one)
  int G = g_x3_grid > 0 ? g_x3_grid : (T >= 2L * cus ? 2 * cus : cus);
  if (U / g_sk_min_units < G) G = (int)(U / g_sk_min_units);  // short launches: fewer, longer ranges
  if (G > kSkMaxGrid) G = kSkMaxGrid;
  G = G >= 8 ? (G / 8) * 8 : 8;
  return launch_sk(g, G, 128, st, true);
}

int gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
            int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  GemmArgs g{A, W, bias, R, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, seq_len, rows_per_seq > 0 ? rows_per_seq : 1};
  const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && aligned16(A) && aligned16(W);
  // Tile choice: keep >= ~1 workgroup per CU where the problem allows it.
  const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128);
  const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
  if (vec) {
    int bt = 128;
    const int G = sk_route(M, N, K, &bt);
    if (G > 0) {
      const int rc = launch_sk(g, G, bt, st);
      if (rc != -1) return rc;  // -1: the caller registered no workspace for this stream
    }
  }
  const bool big = tiles128 >= 768 || (g_gemm_tile & 16);  // +16: take the variant at any size (tests)
  if (big && (g_gemm_tile & 15) == 1) return launch_gemm<256, 128, 32, 64, 64>(g, vec, st);
  if (big && (g_gemm_tile & 15) == 2) return launch_gemm<128, 256, 32, 64, 64>(g, vec, st);
  if (big && (g_gemm_tile & 15) == 3) return launch_gemm<256, 128, 32, 128, 64>(g, vec, st);
  if (big && (g_gemm_tile & 15) == 4) return launch_gemm<128, 128, 64, 64, 64>(g, vec, st);  // one barrier pair per 128 MFMAs
  if (tiles128 >= 384) return launch_gemm<128, 128, 32, 64, 64>(g, vec, st);
  if (tiles64 >= 256 || M > 256) return launch_gemm<64, 64, 32, 32, 32>(g, vec, st);
  return launch_gemm<32, 64, 32, 32, 32>(g, vec, st);
}
}  // namespace sbk

extern "C" size_t sbk_stream_workspace_bytes(void) { return sbk::kSkSlabBytes + sbk::kSkTicketBytes; }

extern "C" int sbk_stream_workspace_set(sbk_stream_t stream, void* workspace, size_t workspace_bytes) {
  SBK_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "stream workspace: null or not 256-byte aligned");
  SBK_REQUIRE(workspace_bytes >= sbk_stream_workspace_bytes(), "stream workspace: %zu bytes given, %zu needed", workspace_bytes,
              sbk_stream_workspace_bytes());
  hipStream_t st = sbk::as_stream(stream);
  sbk::SkWorkspace w{reinterpret_cast<float*>(workspace),
                     reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + sbk::kSkSlabBytes)};
  // tickets start at zero (ordered before the first launch on this stream); every launch leaves them at zero
  const hipError_t e = hipMemsetAsync(w.cnt, 0, sbk::kSkTicketBytes, st);
  if (e != hipSuccess) return sbk::fail((int)e, "stream workspace: memset: %s", hipGetErrorString(e));
  std::lock_guard<std::mutex> lk(sbk::g_sk_mu);
  sbk::g_sk_ws[std::make_pair(sbk::cur_device(), st)] = w;
  return 0;
}

extern "C" int sbk_stream_workspace_release(sbk_stream_t stream) {
  std::lock_guard<std::mutex> lk(sbk::g_sk_mu);
  sbk::g_sk_ws.erase(std::make_pair(sbk::cur_device(), sbk::as_stream(stream)));
  return 0;
}

extern "C" int sbk_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                               const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                               float alpha, const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(A && W && C, "gemm: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  // lda < K is allowed: rows of A then overlap (a strided window over a time-major signal = a 1-D convolution read in place)
  SBK_REQUIRE(lda > 0 && ldw >= K && ldc >= N, "gemm: leading dimension smaller than the row");
  SBK_REQUIRE(!residual || ldr >= N, "gemm: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm: seq_len given without rows_per_seq");
  return sbk::gemm_nt_ws(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq,
                         nullptr, 0, sbk::as_stream(stream));
}

extern "C" int sbk_gemm_ln_nt_f32(const float* A, int lda, const float* Wf, int ldw, const float* bf,
                                  const float* residual, int ldr, float* C, int ldc, int M, int N, int K, float eps,
                                  int act, float alpha, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(A && Wf && C, "gemm_ln: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "gemm_ln: bad shape");
  const int rc = sbk::gemm_ln_nt(A, lda, Wf, ldw, bf, residual, ldr, C, ldc, M, N, K, eps, act, alpha,
                                 sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_ln: shape M=%d N=%d K=%d not eligible for the fused kernel", M, N, K);
  return rc;
}

extern "C" int sbk_gemm_nt_splitk_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                                      const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                                      float alpha, float* workspace, size_t workspace_floats, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(A && W && C, "gemm: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  SBK_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gemm: leading dimension smaller than the row");
  SBK_REQUIRE(!residual || ldr >= N, "gemm: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm: unknown activation %d", act);
  return sbk::gemm_nt_ws(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, nullptr, 0, workspace,
                         workspace ? workspace_floats : 0, sbk::as_stream(stream));
}


// Measurement helper: `iters` back-to-back launches of the same contraction between two events on
// `stream` (host launch overhead amortised); returns the mean time per launch in microseconds.
extern "C" int sbk_prof_gemm_repeat_f32(const float* A, const float* W, float* C, int M, int N, int K, float* workspace,
                                        size_t workspace_floats, int iters, float* us_per_launch,
                                        sbk_stream_t stream) {
  SBK_REQUIRE(A && W && C && us_per_launch && iters > 0, "gemm_repeat: bad arguments");
  hipStream_t st = sbk::as_stream(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return sbk::fail(1, "event create");
  int rc = 0;
  for (int i = 0; i < 3 && !rc; ++i)
    rc = sbk::gemm_nt_ws(A, K, W, K, nullptr, nullptr, 0, C, N, M, N, K, SBK_ACT_NONE, 1.0f, nullptr, 0, workspace,
                         workspace_floats, st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters && !rc; ++i)
    rc = sbk::gemm_nt_ws(A, K, W, K, nullptr, nullptr, 0, C, N, M, N, K, SBK_ACT_NONE, 1.0f, nullptr, 0, workspace,
                         workspace_floats, st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = ms * 1000.0f / iters;
  return rc;
}

extern "C" void sbk_prof_set_knob(int key, int value) {
  if (key == 1) sbk::g_skinny_nch = value;
  if (key == 2) sbk::g_skinny_off = value;
  if (key == 3) sbk::g_attn_prefetch = value;
  if (key == 16) sbk::g_rope_flash_lds = value;
  if (key == 17) sbk::g_relpos_flash_t = value;
  if (key == 4) sbk::g_cross_rows = value;
  if (key == 5) sbk::g_kv_head_major = value;
  if (key == 6) sbk::g_gemm_tile = value;
  if (key == 7) sbk::g_ctc_tpt = value;
  if (key == 8) sbk::g_cross_fc256 = value;
  if (key == 9) sbk::g_gemm_vec_lds = value;
  if (key == 10) sbk::g_skinny_looped = value;
  if (key == 11) sbk::g_flat64_min_rows = value;
  if (key == 12) sbk::g_skinny_reach = value;
  if (key == 14) sbk::g_tiled_splitk = value;
  if (key == 15) sbk::g_tiled_splitk_short = value;
  if (key == 13) sbk::g_self_group_off = value;
  if (key == 18) sbk::g_sk_mode = value;
  if (key == 19) sbk::g_sk_grid = value;
  if (key == 21) sbk::g_sk_min_units = value > 0 ? value : 1;
  if (key == 22) sbk::g_sk_noload = value;
  if (key == 23) sbk::g_sk_stagger = value;
  if (key == 24) sbk::g_sk_min_rows = value;
  if (key == 25) sbk::g_sk64_min_rows = value;
  if (key == 30) sbk::g_sk_interleave = value;
  if (key == 26) sbk::g_sk64_units = value > 0 ? value : 1;
  if (key == 27) sbk::g_bf16a_stages = value;
  if (key == 28) sbk::g_bf16a_grid = value;
  if (key == 29) sbk::g_bf16a_mode = value;
  if (key == 31) sbk::g_x3_grid = value;
  if (key == 39) sbk::g_x3p_tile = value;
  if (key == 40) sbk::g_score_fused = value;
  if (key == 41) sbk::g_x3r_mode = value;
  if (key == 42) sbk::g_x3r_min_rows = value;
  if (key == 45) sbk::g_x3r_ln = value;
  if (key == 47) sbk::g_persist = value;
  if (key == 48) sbk::g_persist_grid = value;
  if (key == 49) sbk::g_persist_stamps = value;
  if (key == 36) sbk::g_splitk_fused = value;
  if (key == 37) sbk::g_cross_fused_merge = value;
  if (key == 34) sbk::g_x3_route_rows = value;
  if (key == 35) sbk::g_x3_route_tiles = value;
}


// ---- fp32 contraction on the bf16 matrix pipe (exact three-way operand split) ----------------------------------
extern "C" int sbk_split_bf16x3(const float* W, int ldw, uint16_t* W3, int N, int K, sbk_stream_t stream) {
  if (N == 0) return 0;
  SBK_REQUIRE(W && W3 && N > 0 && K > 0 && K % 32 == 0 && ldw >= K, "split_bf16x3: bad arguments (K must be a multiple of 32)");
  const long n = (long)N * K, blocks = (n + 255) / 256;
  SBK_LAUNCH(split_bf16x3_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, sbk::as_stream(stream), W, ldw,
             reinterpret_cast<unsigned short*>(W3), n, K);
  return sbk::launch_status("sbk_split_bf16x3");
}

extern "C" int sbk_gemm_nt_f32x3(const float* A, int lda, const uint16_t* W3, const float* bias, const float* residual,
                                 int ldr, float* C, int ldc, int M, int N, int K, int act, float alpha,
                                 const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && W3 && C, "gemm_f32x3: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K >= 64 && K % 32 == 0, "gemm_f32x3: bad shape M=%d N=%d K=%d (K: a multiple of 32, >= 64)", M, N, K);
  SBK_REQUIRE(lda > 0 && lda % 4 == 0 && ldc >= N && sbk::aligned16(A) && sbk::aligned16(W3),
              "gemm_f32x3: operand rows must be 16-byte aligned (lda=%d)", lda);
  SBK_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && sbk::aligned16(C) && (!bias || sbk::aligned16(bias)),
              "gemm_f32x3: N and ldc must be multiples of 4, C / bias 16-byte aligned (rows are stored as 16-byte vectors)");
  SBK_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && sbk::aligned16(residual)), "gemm_f32x3: residual stride / alignment");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_f32x3: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm_f32x3: seq_len given without rows_per_seq");
  const int rc = sbk::gemm_nt_x3(A, lda, W3, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq,
                                 sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_f32x3: no workspace registered for this stream (sbk_stream_workspace_set) or too many tiles");
  return rc;
}

// ---- bf16-operand fast entry points ---------------------------------------------------------------
extern "C" int sbk_f32_to_bf16(const float* x, uint16_t* y, long n, sbk_stream_t stream) {
  if (n == 0) return 0;
  SBK_REQUIRE(x && y && n > 0, "f32_to_bf16: bad arguments");
  const long blocks = (n + 255) / 256;
  SBK_LAUNCH(f32_to_bf16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, sbk::as_stream(stream), x,
             reinterpret_cast<unsigned short*>(y), n);
  return sbk::launch_status("sbk_f32_to_bf16");
}

namespace {
int launch_bf16dma(const Bf16DmaArgs& a0, hipStream_t st) {
  Bf16DmaArgs a = a0;
  a.tiles_n = sbk::cdiv(a.N, 128);
  a.tiles = sbk::cdiv(a.M, 128) * a.tiles_n;
  a.KT = a.K / 64;
  a.mode = sbk::g_bf16a_mode;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus <= 0) cus = 256;
  // Measured on MI355X (tools/microbench.py --bf16a, profiles/r03_bf16_activation_gemm.log): two stages and two
  // workgroups per CU (634-827 TF/s at 12 000 rows) beat three / four stages with one (460-630): a second workgroup's
  // MFMAs cover the ~100-cycle issue of each LDS-DMA piece better than a deeper pipeline of one wave per SIMD does
  const int ns = sbk::g_bf16a_stages == 3 ? 3 : (sbk::g_bf16a_stages == 4 ? 4 : 2);
  int G = sbk::g_bf16a_grid > 0 ? sbk::g_bf16a_grid : (ns == 2 ? 2 * cus : cus);
  if (G > a.tiles) G = a.tiles;
  G = G >= 8 ? (G / 8) * 8 : 8;
  const size_t lds = (size_t)ns * 2 * 128 * 32 * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS(gemm_nt_bf16dma_kernel<2>, (size_t)2 * 2 * 128 * 32 * sizeof(float));
    (void)SBK_ALLOW_DYN_LDS(gemm_nt_bf16dma_kernel<3>, (size_t)3 * 2 * 128 * 32 * sizeof(float));
    (void)SBK_ALLOW_DYN_LDS(gemm_nt_bf16dma_kernel<4>, (size_t)4 * 2 * 128 * 32 * sizeof(float));
    once = true;
  }
  sbk::ProfScope prof("gemm_nt_bf16a", 2.0 * a.M * a.N * a.K,
                      2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (a.C ? 4.0 : 0.0) * a.M * a.N + (a.Cb ? 2.0 : 0.0) * a.M * a.N +
                          (a.R ? 4.0 : 0.0) * a.M * a.N, st);
  if (ns == 2) {
    SBK_LAUNCH(gemm_nt_bf16dma_kernel<2>, dim3((unsigned)G), dim3(256), lds, st, a);
  } else if (ns == 3) {
    SBK_LAUNCH(gemm_nt_bf16dma_kernel<3>, dim3((unsigned)G), dim3(256), lds, st, a);
  } else {
    SBK_LAUNCH(gemm_nt_bf16dma_kernel<4>, dim3((unsigned)G), dim3(256), lds, st, a);
  }
  return sbk::launch_status("sbk_gemm_nt_bf16a");
}

int launch_lp(int dt, const float* A, int lda, const void* Wq, int ldw, const float* bias, const float* residual, int ldr,
              float* C, int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq,
              const float* a_absmax, float w_scale, hipStream_t st) {
  GemmBf16Args g{A, Wq, bias, residual, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, seq_len, rows_per_seq > 0 ? rows_per_seq : 1,
                 a_absmax, w_scale};
  const long tiles128 = (long)sbk::cdiv(M, 128) * sbk::cdiv(N, 128);
  const char* name = dt == 0 ? "gemm_nt_bf16" : (dt == 1 ? "gemm_nt_f16" : "gemm_nt_fp8");
  sbk::ProfScope prof(name, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)M * N) + (dt == 2 ? 1.0 : 2.0) * (double)N * K, st);
  const dim3 g128(sbk::cdiv(N, 128), sbk::cdiv(M, 128)), g64(sbk::cdiv(N, 64), sbk::cdiv(M, 64)), block(256);
#define SBK_LP(DT)                                                           \
  if (tiles128 >= 256) {                                                     \
    SBK_LAUNCH((gemm_nt_lp_kernel<128, 128, DT>), g128, block, 0, st, g);    \
  } else {                                                                   \
    SBK_LAUNCH((gemm_nt_lp_kernel<64, 64, DT>), g64, block, 0, st, g);       \
  }
  if (dt == 0) {
    SBK_LP(0)
  } else if (dt == 1) {
    SBK_LP(1)
  } else {
    SBK_LP(2)
  }
#undef SBK_LP
  return sbk::launch_status(name);
}
}  // namespace

extern "C" int sbk_gemm_nt_bf16(const float* A, int lda, const uint16_t* Wb, int ldw, const float* bias,
                                const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                                float alpha, const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && Wb && C, "gemm_bf16: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0 && K % 8 == 0, "gemm_bf16: bad shape M=%d N=%d K=%d (K must be a multiple of 8)", M, N, K);
  SBK_REQUIRE(lda > 0 && ldw >= K && ldc >= N && lda % 4 == 0 && ldw % 8 == 0, "gemm_bf16: leading dimensions");
  SBK_REQUIRE(sbk::aligned16(A) && sbk::aligned16(Wb), "gemm_bf16: operands must be 16-byte aligned");
  SBK_REQUIRE(!residual || ldr >= N, "gemm_bf16: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_bf16: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm_bf16: seq_len given without rows_per_seq");
  return launch_lp(0, A, lda, Wb, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq, nullptr, 1.0f,
                   sbk::as_stream(stream));
}

extern "C" int sbk_gemm_nt_bf16a(const uint16_t* A, int lda, const uint16_t* Wb, int ldw, const float* bias,
                                 const float* residual, int ldr, float* C, int ldc, uint16_t* Cb, int ldcb, int M, int N,
                                 int K, int act, float alpha, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && Wb && (C || Cb), "gemm_bf16a: null operand");
  SBK_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_bf16a: K must be a multiple of 64 (M=%d N=%d K=%d)", M, N, K);
  SBK_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && sbk::aligned16(A) && sbk::aligned16(Wb),
              "gemm_bf16a: operand rows must be 16-byte aligned (lda=%d ldw=%d)", lda, ldw);
  SBK_REQUIRE((!C || ldc >= N) && (!Cb || ldcb >= N) && (!residual || ldr >= N), "gemm_bf16a: leading dimension smaller than the row");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_bf16a: unknown activation %d", act);
  Bf16DmaArgs a{A, Wb, bias, residual, C, Cb, lda, ldw, ldr, ldc, ldcb, M, N, K, act, alpha, 0, 0, 0, 0};
  return launch_bf16dma(a, sbk::as_stream(stream));
}

extern "C" int sbk_gemm_nt_fp8a(const uint8_t* A8, int lda, const float* a_scale, const uint8_t* W8, int ldw, const float* w_scale,
                                const float* bias, const float* residual, int ldr, float* C, int ldc, uint16_t* Cb, int ldcb,
                                uint8_t* C8, int ldc8, float c8_scale, int M, int N, int K, int act, float alpha,
                                sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A8 && W8 && (C || Cb || C8), "gemm_fp8a: null operand");
  SBK_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8a: K must be a multiple of 128 (M=%d N=%d K=%d)", M, N, K);
  SBK_REQUIRE(lda >= K && ldw >= K && lda % 16 == 0 && ldw % 16 == 0 && sbk::aligned16(A8) && sbk::aligned16(W8),
              "gemm_fp8a: operand rows must be 16-byte aligned (lda=%d ldw=%d)", lda, ldw);
  SBK_REQUIRE((!C || ldc >= N) && (!Cb || ldcb >= N) && (!C8 || (ldc8 >= N && c8_scale > 0.0f)) && (!residual || ldr >= N),
              "gemm_fp8a: leading dimension smaller than the row / non-positive fp8 output scale");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_fp8a: unknown activation %d", act);
  Fp8DmaArgs a{A8, W8, a_scale, w_scale, bias, residual, C, Cb, C8, C8 ? c8_scale : 1.0f, lda, ldw, ldr, ldc, ldcb, ldc8,
               M, N, K, act, alpha, 0, 0, 0};
  a.tiles_n = sbk::cdiv(N, 128);
  a.tiles = sbk::cdiv(M, 128) * a.tiles_n;
  a.KT = K / 128;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus <= 0) cus = 256;
  int G = 2 * cus;  // two stages, two workgroups per CU (launch_bf16dma's measured choice)
  if (G > a.tiles) G = a.tiles;
  G = G >= 8 ? (G / 8) * 8 : 8;
  const size_t lds = (size_t)2 * 2 * 128 * 32 * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS(gemm_nt_fp8dma_kernel, lds);
    once = true;
  }
  hipStream_t st = sbk::as_stream(stream);
  sbk::ProfScope prof("gemm_nt_fp8a", 2.0 * M * (double)N * K,
                      1.0 * ((double)M * K + (double)N * K) + ((C ? 4.0 : 0.0) + (Cb ? 2.0 : 0.0) + (C8 ? 1.0 : 0.0) + (residual ? 4.0 : 0.0)) * M * (double)N, st);
  SBK_LAUNCH(gemm_nt_fp8dma_kernel, dim3((unsigned)G), dim3(256), lds, st, a);
  return sbk::launch_status("sbk_gemm_nt_fp8a");
}

extern "C" int sbk_gemm_nt_f16(const float* A, int lda, const uint16_t* Wh, int ldw, const float* bias,
                               const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                               float alpha, const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && Wh && C, "gemm_f16: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0 && K % 8 == 0, "gemm_f16: bad shape M=%d N=%d K=%d (K must be a multiple of 8)", M, N, K);
  SBK_REQUIRE(lda > 0 && ldw >= K && ldc >= N && lda % 4 == 0 && ldw % 8 == 0, "gemm_f16: leading dimensions");
  SBK_REQUIRE(sbk::aligned16(A) && sbk::aligned16(Wh), "gemm_f16: operands must be 16-byte aligned");
  SBK_REQUIRE(!residual || ldr >= N, "gemm_f16: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_f16: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm_f16: seq_len given without rows_per_seq");
  return launch_lp(1, A, lda, Wh, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq, nullptr, 1.0f,
                   sbk::as_stream(stream));
}

extern "C" int sbk_gemm_nt_fp8(const float* A, int lda, const float* a_absmax, const uint8_t* Wq, int ldw, float w_scale,
                               const float* bias, const float* residual, int ldr, float* C, int ldc, int M, int N, int K,
                               int act, float alpha, const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && Wq && C && a_absmax, "gemm_fp8: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0 && K % 16 == 0, "gemm_fp8: bad shape M=%d N=%d K=%d (K must be a multiple of 16)", M, N, K);
  SBK_REQUIRE(lda > 0 && ldw >= K && ldc >= N && lda % 4 == 0 && ldw % 16 == 0, "gemm_fp8: leading dimensions");
  SBK_REQUIRE(sbk::aligned16(A) && sbk::aligned16(Wq), "gemm_fp8: operands must be 16-byte aligned");
  SBK_REQUIRE(!residual || ldr >= N, "gemm_fp8: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_fp8: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm_fp8: seq_len given without rows_per_seq");
  SBK_REQUIRE(w_scale > 0.0f, "gemm_fp8: w_scale must be positive");
  return launch_lp(2, A, lda, Wq, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq, a_absmax, w_scale,
                   sbk::as_stream(stream));
}

extern "C" int sbk_f32_to_f16(const float* x, uint16_t* y, long n, sbk_stream_t stream) {
  if (n == 0) return 0;
  SBK_REQUIRE(x && y && n > 0, "f32_to_f16: bad arguments");
  const long blocks = (n + 255) / 256;
  SBK_LAUNCH(f32_to_f16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, sbk::as_stream(stream), x,
             reinterpret_cast<unsigned short*>(y), n);
  return sbk::launch_status("sbk_f32_to_f16");
}

// y = e4m3(x * mul) (OCP e4m3fn, round to nearest even, saturating at +-448); n even
extern "C" int sbk_f32_to_fp8(const float* x, uint8_t* y, long n, float mul, sbk_stream_t stream) {
  if (n == 0) return 0;
  SBK_REQUIRE(x && y && n > 0 && n % 2 == 0, "f32_to_fp8: bad arguments (n must be even)");
  const long blocks = (n / 2 + 255) / 256;
  SBK_LAUNCH(f32_to_fp8_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, sbk::as_stream(stream), x,
             reinterpret_cast<unsigned short*>(y), n / 2, mul);
  return sbk::launch_status("sbk_f32_to_fp8");
}

// out[0] = max |x[i]| (device float; the activation scale of sbk_gemm_nt_fp8)
extern "C" int sbk_absmax_f32(const float* x, long n, float* out, sbk_stream_t stream) {
  SBK_REQUIRE(x && out && n > 0, "absmax: bad arguments");
  hipStream_t st = sbk::as_stream(stream);
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return sbk::fail(1, "absmax: memset");
  const long blocks = (n + 255) / 256;
  SBK_LAUNCH(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, x, reinterpret_cast<int*>(out), n);
  return sbk::launch_status("sbk_absmax_f32");
}
