// Reduced-precision contractions (opt-in: run_opts precision "bf16" / "fp16" / "fp8"; SURVEY 8b "fast entry points", BASELINE
// configs[4]): sbk_gemm_nt_bf16 / _f16 / _fp8 (fp32 activations rounded on their way into LDS), sbk_gemm_nt_bf16a (bf16
// activations, LDS-DMA panels), sbk_gemm_nt_fp8a (e4m3 activations and weights on v_mfma_scale_f32_32x32x64_f8f6f4) and the
// conversion kernels.  Moved out of gemm.hip in round 5; reference: the Linear layers under the reference's
// inference autocast (inference/interfaces.py:295-298, integrations/huggingface/whisper.py:318-353).
#include "common.h"
#include "internal.h"
#include "gemm_common.h"

namespace {

using sbk::f32x16;

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
  const int tiles_n = s.tiles_n, KT = s.KT;

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
            for (int r = 0; r < 16; ++r) v[r] = sbk::gelu_erfc(v[r]);
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
    compute(n % NS);
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
            for (int r = 0; r < 16; ++r) v[r] = sbk::gelu_erfc(v[r]);
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

}  // namespace

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
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus <= 0) cus = 256;
  // Measured on MI355X (tools/microbench.py --bf16a, profiles/r03_bf16_activation_gemm.log): two stages and two
  // workgroups per CU (634-827 TF/s at 12 000 rows) beat three / four stages with one (460-630): a second workgroup's
  // MFMAs cover the ~100-cycle issue of each LDS-DMA piece better than a deeper pipeline of one wave per SIMD does
  // (three / four stages with one workgroup per CU were the knobs 27 / 28 of rounds 3-4; removed with their instantiations)
  int G = 2 * cus;
  if (G > a.tiles) G = a.tiles;
  G = G >= 8 ? (G / 8) * 8 : 8;
  const size_t lds = (size_t)2 * 2 * 128 * 32 * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS(gemm_nt_bf16dma_kernel<2>, lds);
    once = true;
  }
  sbk::ProfScope prof("gemm_nt_bf16a", 2.0 * a.M * a.N * a.K,
                      2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (a.C ? 4.0 : 0.0) * a.M * a.N + (a.Cb ? 2.0 : 0.0) * a.M * a.N +
                          (a.R ? 4.0 : 0.0) * a.M * a.N, st);
  SBK_LAUNCH(gemm_nt_bf16dma_kernel<2>, dim3((unsigned)G), dim3(256), lds, st, a);
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
  {  // the large shapes: 256 x 256 tiles (csrc/gemm_lp256.hip), the same sums in the same order
    const sbk::Lp256Args a{reinterpret_cast<const unsigned char*>(A), reinterpret_cast<const unsigned char*>(Wb), nullptr, nullptr, bias,
                           residual, C, Cb, nullptr, 1.0f, 2L * lda, 2L * ldw, ldr, ldc, ldcb, 0, M, N, act, alpha, K / 64, 0, 0, 0, 0};
    if (sbk::lp256_routed(a)) {
      hipStream_t st = sbk::as_stream(stream);
      sbk::ProfScope prof("gemm_nt_bf16a", 2.0 * M * (double)N * K,
                          2.0 * ((double)M * K + (double)N * K) + ((C ? 4.0 : 0.0) + (Cb ? 2.0 : 0.0) + (residual ? 4.0 : 0.0)) * M * (double)N, st);
      return sbk::gemm_nt_lp256(a, false, st);
    }
  }
  Bf16DmaArgs a{A, Wb, bias, residual, C, Cb, lda, ldw, ldr, ldc, ldcb, M, N, K, act, alpha, 0, 0, 0};
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
  {  // the large shapes: 256 x 256 tiles (csrc/gemm_lp256.hip), the same sums in the same order
    const sbk::Lp256Args a{A8, W8, a_scale, w_scale, bias, residual, C, Cb, C8, C8 ? c8_scale : 1.0f, (long)lda, (long)ldw, ldr, ldc, ldcb, ldc8,
                           M, N, act, alpha, K / 128, 0, 0, 0, 0};
    if (sbk::lp256_routed(a)) {
      hipStream_t st = sbk::as_stream(stream);
      sbk::ProfScope prof("gemm_nt_fp8a", 2.0 * M * (double)N * K,
                          1.0 * ((double)M * K + (double)N * K) + ((C ? 4.0 : 0.0) + (Cb ? 2.0 : 0.0) + (C8 ? 1.0 : 0.0) + (residual ? 4.0 : 0.0)) * M * (double)N, st);
      return sbk::gemm_nt_lp256(a, true, st);
    }
  }
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
