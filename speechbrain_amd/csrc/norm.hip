// Row-wise normalisation kernels (HBM-bound; one wavefront per row).
//
//  sbk_layernorm_f32        y = act(LayerNorm(x))  over the last dimension
//  sbk_input_norm_global_f32 y = (x - mean[c]) / max(std[c], eps)
//  sbk_input_norm_stats_f32  the same with the mean / std of the utterance ("sentence") or of the batch ("batch")
//
// A row of d floats (d <= a few thousand on this path) is read by the 64 lanes
// of one wave with 16-byte loads, reduced with wave shuffles (no LDS, no
// barrier) and written once: 2*d*4 bytes of HBM traffic per row.
#include "common.h"

namespace {

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == SBK_ACT_SWISH) return v / (1.0f + expf(-v));
  if (act == SBK_ACT_LEAKY_RELU) return v > 0.0f ? v : 0.01f * v;
  if (act == SBK_ACT_RELU) return v > 0.0f ? v : 0.0f;
  return v;
}

// MAXV = float4 slots kept in registers per lane (d <= 256*MAXV).
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        unsigned short* __restrict__ yb, int rows, int d, float eps,
                                                        int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = row < rows;
  const int r = live ? row : rows - 1;  // idle waves shadow the last row so shuffles stay full-width
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)r * d);
  const int nv = d >> 2;
  float4 v[MAXV];
  float s = 0.0f;
  // the whole row requested first: unconditional loads of clamped slots (a load under a lane mask is a branch and a full wait each;
  // in source order load / use / load / use the compiler keeps that order); slots past the row are zeros
#pragma unroll
  for (int i = 0; i < MAXV; ++i) v[i] = xr[min(lane + i * 64, nv - 1)];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c >= nv) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = sbk::wave_sum(s) / (float)d;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + dd * dd);
    }
  }
  const float rstd = rsqrtf(sbk::wave_sum(q) / (float)d + eps);
  if (!live) return;
  float4* yr = reinterpret_cast<float4*>(y + (size_t)row * d);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const float4 g = g4[min(c, nv - 1)], b = b4[min(c, nv - 1)];
    if (c < nv) {
      float4 o;
      o.x = act_f((v[i].x - mean) * rstd * g.x + b.x, act);
      o.y = act_f((v[i].y - mean) * rstd * g.y + b.y, act);
      o.z = act_f((v[i].z - mean) * rstd * g.z + b.z, act);
      o.w = act_f((v[i].w - mean) * rstd * g.w + b.w, act);
      if (yb) {  // (uniform) bf16 output: the next contraction's operand, rounded to nearest even
        uint2 pk;
        pk.x = (unsigned)sbk::f32_to_bf16(o.x) | ((unsigned)sbk::f32_to_bf16(o.y) << 16);
        pk.y = (unsigned)sbk::f32_to_bf16(o.z) | ((unsigned)sbk::f32_to_bf16(o.w) << 16);
        reinterpret_cast<uint2*>(yb + (size_t)row * d)[c] = pk;
      } else {
        yr[c] = o;
      }
    }
  }
}

// y = act(LayerNorm(x)) written as the PANEL image of the [rows, d] result (csrc/gemm_x3p.hip: [rows/64][d/16][3 pieces]
// [2 k-halves][64 rows][8 k] bf16, x = hi + mid + lo exactly): the A operand of sbk_gemm_nt_x3p made by its producer --
// the contraction then reads 6 B per element once instead of the fp32 row being written, re-read and split by a pass of
// its own.  A wave = one row, a lane = NV runs of 8 consecutive k (one 16-byte slot of a chunk per piece); a workgroup =
// 8 consecutive rows, i.e. the eight 16-byte slots of every 128-byte line of the image are written by the same
// workgroup at the same time (they merge in its XCD's L2).  Rows past `rows` up to the next multiple of 64 are written
// as zeros (the contraction may read them).
template <int NV>
__global__ void __launch_bounds__(512) layernorm_x3p_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, uint4* __restrict__ P, int rows,
                                                            int d, float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 6);
  const bool live = row < rows;
  const float* xr = x + (size_t)(live ? row : rows - 1) * d;  // padding rows shadow the last row: shuffles stay full-width
  const int nu = d >> 3, KB = d >> 4;
  float v[NV][8];
  float s = 0.0f;
  float4 xa[NV], xb[NV];  // (the whole row requested first, unconditionally, on clamped runs: see layernorm_kernel)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int uc = min(lane + 64 * i, nu - 1);
    xa[i] = *reinterpret_cast<const float4*>(xr + uc * 8);
    xb[i] = *reinterpret_cast<const float4*>(xr + uc * 8 + 4);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int u = lane + 64 * i;
    float4 a = xa[i], b = xb[i];
    if (u >= nu) a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    v[i][0] = a.x, v[i][1] = a.y, v[i][2] = a.z, v[i][3] = a.w, v[i][4] = b.x, v[i][5] = b.y, v[i][6] = b.z, v[i][7] = b.w;
    s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
  }
  const float mean = sbk::wave_sum(s) / (float)d;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nu) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (v[i][e] - mean) * (v[i][e] - mean);
      q += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
  const float rstd = rsqrtf(sbk::wave_sum(q) / (float)d + eps);
  const int rb = row >> 6, rr = row & 63;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int u = lane + 64 * i, uc = min(u, nu - 1);
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + uc * 8), g1 = *reinterpret_cast<const float4*>(gamma + uc * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + uc * 8), b1 = *reinterpret_cast<const float4*>(beta + uc * 8 + 4);
    if (u >= nu) continue;
    float o[8];
    {
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = live ? act_f((v[i][e] - mean) * rstd * g[e] + bb[e], act) : 0.0f;
    }
    unsigned hi[4], mi[4], lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // x = hi + mid + lo exactly (split_x3p_kernel's arithmetic)
      hi[p] = sbk::bf16_pair(o[2 * p], o[2 * p + 1]);
      const float r0 = o[2 * p] - __uint_as_float(hi[p] << 16), r1 = o[2 * p + 1] - __uint_as_float(hi[p] & 0xffff0000u);
      mi[p] = sbk::bf16_pair(r0, r1);
      lo[p] = sbk::bf16_pair(r0 - __uint_as_float(mi[p] << 16), r1 - __uint_as_float(mi[p] & 0xffff0000u));
    }
    uint4* dst = P + ((size_t)(rb * KB + (u >> 1)) * 6 + (u & 1)) * 64 + rr;  // chunk (rb, kb, piece 0, half), slot rr
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[128] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
    dst[256] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// The same for d <= 1024 with FULL-LINE stores: a wave = 8 consecutive rows, lane = (row r = lane >> 3, column group
// u = lane & 7) holds the runs q = u + 8 j of 8 consecutive k of its row; the statistics are reduced over the 8 lanes of a
// row (DPP); a store instruction then writes, for each u, the 16-byte slots of 8 consecutive rows = one whole 128-byte
// line of the image (the one-row-per-wave kernel above writes 64 scattered 16-byte pieces per instruction: measured
// 22.3 us at 12 800 x 512 -- no faster than LayerNorm + the split pass; profiles/r04_f_*).
template <int NJ>
__global__ void __launch_bounds__(256) layernorm_x3p_rows8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, uint4* __restrict__ P,
                                                                  int rows, int d, float eps, int act) {
  const int lane = threadIdx.x & 63, r = lane >> 3, u = lane & 7;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + r;
  const bool live = row < rows;
  const float* xr = x + (size_t)(live ? row : rows - 1) * d;
  const int nu = d >> 3, KB = d >> 4;
  float v[NJ][8];
  float s = 0.0f;
  // the whole row requested first: unconditional loads of clamped pieces (a load under a lane mask is a branch and a full wait
  // each, NJ round trips in a row; and in source order load / use / load / use the compiler keeps that order)
  float4 xa[NJ], xb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int qc = min(u + 8 * j, nu - 1);
    xa[j] = *reinterpret_cast<const float4*>(xr + qc * 8);
    xb[j] = *reinterpret_cast<const float4*>(xr + qc * 8 + 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int q = u + 8 * j;
    float4 a = xa[j], b = xb[j];
    if (q >= nu) a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    v[j][0] = a.x, v[j][1] = a.y, v[j][2] = a.z, v[j][3] = a.w, v[j][4] = b.x, v[j][5] = b.y, v[j][6] = b.z, v[j][7] = b.w;
    s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
  }
  const float mean = sbk::group_sum<8>(s) / (float)d;
  float qs = 0.0f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (u + 8 * j < nu) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (v[j][e] - mean) * (v[j][e] - mean);
      qs += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
  const float rstd = rsqrtf(sbk::group_sum<8>(qs) / (float)d + eps);
  const int rb = row >> 6, rr = row & 63;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int q = u + 8 * j, qc = min(q, nu - 1);
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + qc * 8), g1 = *reinterpret_cast<const float4*>(gamma + qc * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + qc * 8), b1 = *reinterpret_cast<const float4*>(beta + qc * 8 + 4);
    if (q >= nu) continue;
    float o[8];
    {
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = live ? act_f((v[j][e] - mean) * rstd * g[e] + bb[e], act) : 0.0f;
    }
    unsigned hi[4], mi[4], lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      hi[p] = sbk::bf16_pair(o[2 * p], o[2 * p + 1]);
      const float r0 = o[2 * p] - __uint_as_float(hi[p] << 16), r1 = o[2 * p + 1] - __uint_as_float(hi[p] & 0xffff0000u);
      mi[p] = sbk::bf16_pair(r0, r1);
      lo[p] = sbk::bf16_pair(r0 - __uint_as_float(mi[p] << 16), r1 - __uint_as_float(mi[p] & 0xffff0000u));
    }
    uint4* dst = P + ((size_t)(rb * KB + (q >> 1)) * 6 + (q & 1)) * 64 + rr;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[128] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
    dst[256] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// ---- fp8 (OCP e4m3) rows with one fp32 scale per row: q[r][c] = e4m3(v[r][c] / scale[r]), scale[r] = max_c |v[r][c]| / 448
// (1 for an all-zero row) -- the operand format of sbk_gemm_nt_fp8a.  LN = true: v = act(LayerNorm(x)) (the row is in
// registers anyway: the maximum is one more wave reduction); LN = false: v = x (weights: one scale per output channel).
// One wave per row, d % 4 == 0, d <= 256 * MAXV.
// BF16IN: x holds bf16 rows (the attention kernel's context), ldx in bf16 elements
template <int MAXV, bool LN, bool BF16IN = false>
__global__ void __launch_bounds__(256) rows_fp8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, unsigned char* __restrict__ q,
                                                       float* __restrict__ scale, int rows, int d, int ldx, float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = row < rows;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)(live ? row : rows - 1) * ldx);
  const uint2* xb = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + (size_t)(live ? row : rows - 1) * ldx);
  const int nv = d >> 2;
  float4 v[MAXV];
  float s = 0.0f;
  uint2 ub[BF16IN ? MAXV : 1];  // (the whole row requested first, unconditionally, on clamped slots: see layernorm_kernel)
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int cl = min(lane + i * 64, nv - 1);
    if constexpr (BF16IN) {
      ub[i] = xb[cl];
    } else {
      v[i] = xr[cl];
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if constexpr (BF16IN) {
      uint2 u = ub[i];
      if (c >= nv) u = make_uint2(0u, 0u);
      v[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                         __uint_as_float(u.y & 0xffff0000u));
    } else {
      if (c >= nv) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  if constexpr (LN) {  // (layernorm_kernel's arithmetic)
    const float mean = sbk::wave_sum(s) / (float)d;
    float qs = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + i * 64 < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
        qs += (a * a + b * b) + (cc * cc + dd * dd);
      }
    const float rstd = rsqrtf(sbk::wave_sum(qs) / (float)d + eps);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      const float4 g = g4[min(c, nv - 1)], b = b4[min(c, nv - 1)];
      if (c < nv) {
        v[i].x = act_f((v[i].x - mean) * rstd * g.x + b.x, act);
        v[i].y = act_f((v[i].y - mean) * rstd * g.y + b.y, act);
        v[i].z = act_f((v[i].z - mean) * rstd * g.z + b.z, act);
        v[i].w = act_f((v[i].w - mean) * rstd * g.w + b.w, act);
      }
    }
  }
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + i * 64 < nv) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
  m = sbk::wave_max(m);
  const float sc = m > 0.0f ? m / 448.0f : 1.0f, inv = 1.0f / sc;
  if (!live) return;
  if (lane == 0) scale[row] = sc;
  unsigned* qr = reinterpret_cast<unsigned*>(q + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv)
      qr[c] = (unsigned)sbk::f32x2_to_fp8(v[i].x * inv, v[i].y * inv) | ((unsigned)sbk::f32x2_to_fp8(v[i].z * inv, v[i].w * inv) << 16);
  }
}

// Any d (scalar loads, three passes over an L1/L2-resident row).
__global__ void __launch_bounds__(256) layernorm_generic_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                unsigned short* __restrict__ yb, int rows, int d,
                                                                float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = row < rows;
  const int r = live ? row : rows - 1;
  const float* xr = x + (size_t)r * d;
  float s = 0.0f;
  for (int c = lane; c < d; c += 64) s += xr[c];
  const float mean = sbk::wave_sum(s) / (float)d;
  float q = 0.0f;
  for (int c = lane; c < d; c += 64) {
    const float a = xr[c] - mean;
    q += a * a;
  }
  const float rstd = rsqrtf(sbk::wave_sum(q) / (float)d + eps);
  if (!live) return;
  float* yr = y + (size_t)row * d;
  for (int c = lane; c < d; c += 64) {
    const float o = act_f((xr[c] - mean) * rstd * gamma[c] + beta[c], act);
    if (yb) {
      yb[(size_t)row * d + c] = sbk::f32_to_bf16(o);
    } else {
      yr[c] = o;
    }
  }
}

__global__ void __launch_bounds__(256) input_norm_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ sd, float* __restrict__ y,
                                                         long n, int C, float eps) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    y[i] = (x[i] - mean[c]) / fmaxf(sd[c], eps);
  }
}

// global statistics with avoid_padding_norm (features.py:1447-1449): the padded frames (t >= n_valid[b]) keep their
// values (mean 0, std 1); x [B,T,C]
__global__ void __launch_bounds__(256) input_norm_masked_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ sd, const int* __restrict__ n_valid,
                                                                float* __restrict__ y, long n, int T, int C, float eps) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long row = i / C;
    const bool valid = (int)(row % T) < n_valid[row / T];
    y[i] = valid ? (x[i] - mean[c]) / fmaxf(sd[c], eps) : (x[i] - 0.0f) / fmaxf(1.0f, eps);
  }
}

// ---- InputNormalization with statistics of the input itself (norm_type "sentence" / "batch") -------------------
// x [B,T,C]; n_valid[b] frames of utterance b count.  Two-pass moments like the reference (mean first, then the mean
// of squared deviations), reduced in a fixed order: S time-splits per utterance write partial sums that every
// consumer adds up in the same order, so the result does not depend on the launch schedule.
constexpr int NORM_SPLITS = 8;

// part[b][split][c] = sum over the split's valid frames of  x - centre[c]  (squared when `centred`)
//   centred = 0: plain sums (first pass);  centred = 1: squared deviations from the mean of the first pass
//   per_batch: the mean is that of the whole batch (all utterances' valid frames) instead of utterance b's own
__global__ void __launch_bounds__(256) norm_partial_kernel(const float* __restrict__ x, const int* __restrict__ n_valid,
                                                           const float* __restrict__ sums, float* __restrict__ part,
                                                           int B, int T, int C, int centred, int per_batch) {
  __shared__ float red[4][64];
  const int b = blockIdx.x, split = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = n_valid[b] < T ? n_valid[b] : T;
  const int per = (T + NORM_SPLITS - 1) / NORM_SPLITS;
  const int t0 = split * per, t1 = (t0 + per < n) ? t0 + per : n;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    float centre = 0.0f;
    if (centred && c < C) {
      float tot = 0.0f;
      long cnt = 0;
      if (per_batch) {
        for (int u = 0; u < B; ++u) {
          for (int k = 0; k < NORM_SPLITS; ++k) tot += sums[((size_t)u * NORM_SPLITS + k) * C + c];
          cnt += n_valid[u] < T ? n_valid[u] : T;
        }
      } else {
        for (int k = 0; k < NORM_SPLITS; ++k) tot += sums[((size_t)b * NORM_SPLITS + k) * C + c];
        cnt = n;
      }
      centre = tot / (float)cnt;
    }
    float acc = 0.0f;
    if (c < C) {
      for (int t = t0 + wave; t < t1; t += 4) {
        const float v = x[((size_t)b * T + t) * C + c] - centre;
        acc += centred ? v * v : v;
      }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < C)
      part[((size_t)b * NORM_SPLITS + split) * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) norm_apply_kernel(const float* __restrict__ x, const int* __restrict__ n_valid,
                                                         const float* __restrict__ sums, const float* __restrict__ sq,
                                                         float* __restrict__ y, int B, int T, int C, int per_batch,
                                                         int std_norm, float eps, int avoid_padding) {
  SBK_DYN_LDS(float, stat);  // mean[C], std[C] of this block's utterance
  const int b = blockIdx.x;
  const int n = n_valid[b] < T ? n_valid[b] : T;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float tot = 0.0f, dev = 0.0f;
    long cnt = 0;
    if (per_batch) {
      for (int u = 0; u < B; ++u) {
        for (int k = 0; k < NORM_SPLITS; ++k) {
          tot += sums[((size_t)u * NORM_SPLITS + k) * C + c];
          dev += sq[((size_t)u * NORM_SPLITS + k) * C + c];
        }
        cnt += n_valid[u] < T ? n_valid[u] : T;
      }
    } else {
      for (int k = 0; k < NORM_SPLITS; ++k) {
        tot += sums[((size_t)b * NORM_SPLITS + k) * C + c];
        dev += sq[((size_t)b * NORM_SPLITS + k) * C + c];
      }
      cnt = n;
    }
    const float var = dev / (float)cnt;
    stat[c] = tot / (float)cnt;
    // features.py:1437-1440: "batch" clamps the variance at eps before the root, "sentence" takes the root as it is
    stat[C + c] = std_norm ? sqrtf(per_batch ? fmaxf(var, eps) : var) : 1.0f;
  }
  __syncthreads();
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int t0 = blockIdx.y * per, t1 = (t0 + per < T) ? t0 + per : T;
  const size_t base = (size_t)b * T * C;
  for (size_t i = (size_t)t0 * C + threadIdx.x; i < (size_t)t1 * C; i += blockDim.x) {
    const int c = (int)(i % C);
    const bool pad = avoid_padding && (int)(i / C) >= n;  // padded frames: mean 0, std 1 (features.py:1451-1453)
    const float m = pad ? 0.0f : stat[c];
    const float sd = pad ? 1.0f : stat[C + c];
    y[base + i] = (x[base + i] - m) / fmaxf(sd, eps);
  }
}

}  // namespace

namespace sbk {
int layernorm_any(const float* x, const float* gamma, const float* beta, float* y, unsigned short* yb, int rows, int d,
                  float eps, int act, hipStream_t st) {
  if (rows == 0) return 0;
  dim3 grid(cdiv(rows, 4)), block(256);
  ProfScope prof("layernorm", 8.0 * rows * d, (yb ? 6.0 : 8.0) * rows * d, st);
  const bool vec = (d % 4 == 0) && aligned16(x) && (yb ? (reinterpret_cast<uintptr_t>(yb) & 7) == 0 : aligned16(y)) &&
                   aligned16(gamma) && aligned16(beta);
  if (vec && d <= 256 * 1) {
    SBK_LAUNCH((layernorm_kernel<1>), grid, block, 0, st, x, gamma, beta, y, yb, rows, d, eps, act);
  } else if (vec && d <= 256 * 2) {
    SBK_LAUNCH((layernorm_kernel<2>), grid, block, 0, st, x, gamma, beta, y, yb, rows, d, eps, act);
  } else if (vec && d <= 256 * 4) {
    SBK_LAUNCH((layernorm_kernel<4>), grid, block, 0, st, x, gamma, beta, y, yb, rows, d, eps, act);
  } else if (vec && d <= 256 * 5) {  // d = 1 280 (Whisper large)
    SBK_LAUNCH((layernorm_kernel<5>), grid, block, 0, st, x, gamma, beta, y, yb, rows, d, eps, act);
  } else {
    SBK_LAUNCH(layernorm_generic_kernel, grid, block, 0, st, x, gamma, beta, y, yb, rows, d, eps, act);
  }
  return launch_status("sbk_layernorm_f32");
}
int layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int d, float eps, int act,
              hipStream_t st) {
  return layernorm_any(x, gamma, beta, y, nullptr, rows, d, eps, act, st);
}
}  // namespace sbk

extern "C" int sbk_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int d,
                                 float eps, int act, sbk_stream_t stream) {
  if (rows == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(x && gamma && beta && y, "layernorm: null operand");
  SBK_REQUIRE(rows >= 0 && d > 0, "layernorm: bad shape rows=%d d=%d", rows, d);
  return sbk::layernorm(x, gamma, beta, y, rows, d, eps, act, sbk::as_stream(stream));
}

extern "C" int sbk_layernorm_bf16o(const float* x, const float* gamma, const float* beta, uint16_t* y, int rows, int d,
                                   float eps, int act, sbk_stream_t stream) {
  if (rows == 0) return 0;
  SBK_REQUIRE(x && gamma && beta && y, "layernorm_bf16o: null operand");
  SBK_REQUIRE(rows >= 0 && d > 0, "layernorm_bf16o: bad shape rows=%d d=%d", rows, d);
  return sbk::layernorm_any(x, gamma, beta, nullptr, y, rows, d, eps, act, sbk::as_stream(stream));
}

namespace {
template <bool LN>
int launch_rows_fp8(const float* x, int ldx, const float* gamma, const float* beta, unsigned char* q, float* scale, int rows, int d,
                    float eps, int act, hipStream_t st) {
  dim3 grid(sbk::cdiv(rows, 4)), block(256);
  sbk::ProfScope prof(LN ? "layernorm_fp8" : "quant_rows_fp8", 8.0 * rows * d, 5.0 * rows * d, st);
  if (d <= 256) {
    SBK_LAUNCH((rows_fp8_kernel<1, LN>), grid, block, 0, st, x, gamma, beta, q, scale, rows, d, ldx, eps, act);
  } else if (d <= 512) {
    SBK_LAUNCH((rows_fp8_kernel<2, LN>), grid, block, 0, st, x, gamma, beta, q, scale, rows, d, ldx, eps, act);
  } else if (d <= 1024) {
    SBK_LAUNCH((rows_fp8_kernel<4, LN>), grid, block, 0, st, x, gamma, beta, q, scale, rows, d, ldx, eps, act);
  } else if (d <= 1280) {
    SBK_LAUNCH((rows_fp8_kernel<5, LN>), grid, block, 0, st, x, gamma, beta, q, scale, rows, d, ldx, eps, act);
  } else if (d <= 2048) {
    SBK_LAUNCH((rows_fp8_kernel<8, LN>), grid, block, 0, st, x, gamma, beta, q, scale, rows, d, ldx, eps, act);
  } else {
    SBK_LAUNCH((rows_fp8_kernel<20, LN>), grid, block, 0, st, x, gamma, beta, q, scale, rows, d, ldx, eps, act);
  }
  return sbk::launch_status(LN ? "sbk_layernorm_fp8o" : "sbk_quant_rows_fp8");
}
}  // namespace

extern "C" int sbk_layernorm_fp8o(const float* x, const float* gamma, const float* beta, uint8_t* q, float* scale, int rows,
                                  int d, float eps, int act, sbk_stream_t stream) {
  if (rows == 0) return 0;
  SBK_REQUIRE(x && gamma && beta && q && scale, "layernorm_fp8o: null operand");
  SBK_REQUIRE(rows > 0 && d >= 4 && d % 4 == 0 && d <= 5120, "layernorm_fp8o: rows=%d d=%d (d %% 4 == 0, d <= 5120)", rows, d);
  SBK_REQUIRE(sbk::aligned16(x) && sbk::aligned16(gamma) && sbk::aligned16(beta) && (reinterpret_cast<uintptr_t>(q) & 3) == 0,
              "layernorm_fp8o: operand alignment");
  return launch_rows_fp8<true>(x, d, gamma, beta, q, scale, rows, d, eps, act, sbk::as_stream(stream));
}

extern "C" int sbk_quant_rows_bf16_fp8(const uint16_t* xb, int ldx, uint8_t* q, float* scale, int rows, int d, sbk_stream_t stream) {
  if (rows == 0) return 0;
  SBK_REQUIRE(xb && q && scale, "quant_rows_bf16_fp8: null operand");
  SBK_REQUIRE(rows > 0 && d >= 4 && d % 4 == 0 && d <= 2048 && ldx >= d && ldx % 4 == 0,
              "quant_rows_bf16_fp8: rows=%d d=%d ldx=%d (d %% 4 == 0, d <= 2048)", rows, d, ldx);
  SBK_REQUIRE((reinterpret_cast<uintptr_t>(xb) & 7) == 0 && (reinterpret_cast<uintptr_t>(q) & 3) == 0, "quant_rows_bf16_fp8: operand alignment");
  hipStream_t st = sbk::as_stream(stream);
  const float* x = reinterpret_cast<const float*>(xb);
  dim3 grid(sbk::cdiv(rows, 4)), block(256);
  sbk::ProfScope prof("quant_rows_fp8", 2.0 * rows * d, 3.0 * rows * d, st);
  if (d <= 512) {
    SBK_LAUNCH((rows_fp8_kernel<2, false, true>), grid, block, 0, st, x, nullptr, nullptr, q, scale, rows, d, ldx, 0.0f, 0);
  } else if (d <= 1280) {
    SBK_LAUNCH((rows_fp8_kernel<5, false, true>), grid, block, 0, st, x, nullptr, nullptr, q, scale, rows, d, ldx, 0.0f, 0);
  } else {
    SBK_LAUNCH((rows_fp8_kernel<8, false, true>), grid, block, 0, st, x, nullptr, nullptr, q, scale, rows, d, ldx, 0.0f, 0);
  }
  return sbk::launch_status("sbk_quant_rows_bf16_fp8");
}

extern "C" int sbk_quant_rows_fp8(const float* x, int ldx, uint8_t* q, float* scale, int rows, int d, sbk_stream_t stream) {
  if (rows == 0) return 0;
  SBK_REQUIRE(x && q && scale, "quant_rows_fp8: null operand");
  SBK_REQUIRE(rows > 0 && d >= 4 && d % 4 == 0 && d <= 5120 && ldx >= d && ldx % 4 == 0,
              "quant_rows_fp8: rows=%d d=%d ldx=%d (d %% 4 == 0, d <= 5120)", rows, d, ldx);
  SBK_REQUIRE(sbk::aligned16(x) && (reinterpret_cast<uintptr_t>(q) & 3) == 0, "quant_rows_fp8: operand alignment");
  return launch_rows_fp8<false>(x, ldx, nullptr, nullptr, q, scale, rows, d, 0.0f, 0, sbk::as_stream(stream));
}

extern "C" int sbk_layernorm_x3p(const float* x, const float* gamma, const float* beta, uint16_t* P, int rows, int d,
                                 float eps, int act, sbk_stream_t stream) {
  if (rows == 0) return 0;
  SBK_REQUIRE(x && gamma && beta && P, "layernorm_x3p: null operand");
  SBK_REQUIRE(rows > 0 && d >= 16 && d % 16 == 0 && d <= 2048, "layernorm_x3p: rows=%d d=%d (d %% 16 == 0, d <= 2048)", rows, d);
  SBK_REQUIRE(sbk::aligned16(x) && sbk::aligned16(gamma) && sbk::aligned16(beta) && sbk::aligned16(P),
              "layernorm_x3p: operands must be 16-byte aligned");
  hipStream_t st = sbk::as_stream(stream);
  const int rows64 = ((rows + 63) / 64) * 64;
  sbk::ProfScope prof("layernorm_x3p", 8.0 * rows * d, (4.0 * rows + 6.0 * rows64) * d, st);
  uint4* P4 = reinterpret_cast<uint4*>(P);
  const dim3 grid(rows64 / 8), block(512);
  // (few rows: the 8-rows-per-wave kernel leaves the chip empty -- 32 us at 1 280 x 512, 40 workgroups -- so the
  // row-per-wave kernel takes them; profiles/r04_i_*)
  if (d <= 512 && rows >= 4096) {
    SBK_LAUNCH(layernorm_x3p_rows8_kernel<8>, dim3(rows64 / 32), dim3(256), 0, st, x, gamma, beta, P4, rows, d, eps, act);
  } else if (d > 512 && d <= 1024 && rows >= 4096) {
    SBK_LAUNCH(layernorm_x3p_rows8_kernel<16>, dim3(rows64 / 32), dim3(256), 0, st, x, gamma, beta, P4, rows, d, eps, act);
  } else if (d <= 512) {
    SBK_LAUNCH(layernorm_x3p_kernel<1>, grid, block, 0, st, x, gamma, beta, P4, rows, d, eps, act);
  } else if (d <= 1024) {
    SBK_LAUNCH(layernorm_x3p_kernel<2>, grid, block, 0, st, x, gamma, beta, P4, rows, d, eps, act);
  } else {
    SBK_LAUNCH(layernorm_x3p_kernel<4>, grid, block, 0, st, x, gamma, beta, P4, rows, d, eps, act);
  }
  return sbk::launch_status("sbk_layernorm_x3p");
}

extern "C" size_t sbk_input_norm_stats_workspace_bytes(int B, int C) {
  return (size_t)2 * (B > 0 ? B : 0) * NORM_SPLITS * (C > 0 ? C : 0) * sizeof(float);
}

extern "C" int sbk_input_norm_stats_f32(const float* x, const int32_t* n_valid, float* y, float* workspace, int B, int T,
                                        int C, int per_batch, int std_norm, float eps, int avoid_padding_norm,
                                        sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;  // empty batch: nothing to launch
  SBK_REQUIRE(x && n_valid && y && workspace, "input_norm_stats: null operand");
  SBK_REQUIRE(B > 0 && T > 0 && C > 0 && C <= 4096, "input_norm_stats: bad shape B=%d T=%d C=%d", B, T, C);
  hipStream_t st = sbk::as_stream(stream);
  float* sums = workspace;
  float* sq = workspace + (size_t)B * NORM_SPLITS * C;
  const double bytes = 4.0 * B * (double)T * C;
  {
    sbk::ProfScope prof("input_norm_stats", 0.0, 4.0 * bytes, st);
    SBK_LAUNCH(norm_partial_kernel, dim3(B, NORM_SPLITS), dim3(256), 0, st, x, n_valid, sums, sums, B, T, C, 0, per_batch);
    SBK_LAUNCH(norm_partial_kernel, dim3(B, NORM_SPLITS), dim3(256), 0, st, x, n_valid, sums, sq, B, T, C, 1, per_batch);
    const int ysplit = T >= 512 ? 16 : (T >= 64 ? 4 : 1);
    SBK_LAUNCH(norm_apply_kernel, dim3(B, ysplit), dim3(256), 2 * C * sizeof(float), st, x, n_valid, sums, sq, y, B, T, C,
               per_batch, std_norm, eps, avoid_padding_norm);
  }
  return sbk::launch_status("sbk_input_norm_stats_f32");
}

extern "C" int sbk_input_norm_global_masked_f32(const float* x, const float* mean, const float* std, const int32_t* n_valid,
                                                float* y, int B, int T, int C, float eps, sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;
  SBK_REQUIRE(x && mean && std && n_valid && y, "input_norm_masked: null operand");
  SBK_REQUIRE(B > 0 && T > 0 && C > 0, "input_norm_masked: bad shape B=%d T=%d C=%d", B, T, C);
  const long n = (long)B * T * C;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  SBK_LAUNCH(input_norm_masked_kernel, dim3(blocks), dim3(256), 0, sbk::as_stream(stream), x, mean, std, n_valid, y, n, T, C, eps);
  return sbk::launch_status("sbk_input_norm_global_masked_f32");
}

extern "C" int sbk_input_norm_global_f32(const float* x, const float* mean, const float* std, float* y, int rows,
                                         int C, float eps, sbk_stream_t stream) {
  if (rows == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(x && mean && std && y, "input_norm: null operand");
  SBK_REQUIRE(rows >= 0 && C > 0, "input_norm: bad shape");
  const long n = (long)rows * C;
  if (n == 0) return 0;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  SBK_LAUNCH(input_norm_kernel, dim3(blocks), dim3(256), 0, sbk::as_stream(stream), x, mean, std, y, n, C, eps);
  return sbk::launch_status("sbk_input_norm_global_f32");
}
