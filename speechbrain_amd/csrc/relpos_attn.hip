// Fused relative-position self-attention (RelPosMHAXL, Transformer-XL style)
//                                                      (sbk_relpos_attention_f32)
//   out = softmax( (q+u)K^T s + shift((q+v)P^T s) , keys < key_len ) V
//
// Roofline: MFMA fp32 (three 32x32 contractions per key tile: AC, BD x2, then
// P.V).  HBM traffic is the fused minimum: q/k/v are read from the interleaved
// in_proj output [B,T,H,(q|k|v)] exactly as the projection GEMM wrote it, the
// context is written as [B,T,H*Dh] for out_proj; the [T,2T-1] position-score
// matrix and the [T,T] score matrix of the reference never exist in HBM.
//
// One workgroup = one (batch, head, 32-query tile); its 4 waves split the key
// tiles.  The relative shift is done in registers: for a 32x32 (query,key)
// tile the needed position rows are the 63 consecutive rows
// r = T-1-i+j, so the wave computes G = (Q+v) P[rbase..rbase+63]^T with two
// MFMA column tiles and every lane adds its G[i][r] straight into
// S[i][j = r + i - 31 + j0] (each S element is touched by exactly one lane).
// The whole score strip S[32][T] lives in LDS (T <= ~1100), so the softmax is
// the exact two-pass form (max, exp, sum, divide) of the reference.
//
// MFMA operand trick: the two k-slices of v_mfma_f32_32x32x2 are fed from the
// two halves of the head dimension (k=0 -> columns [0,Dh/2), k=1 -> [Dh/2,Dh))
// so each lane's K / P operand is one contiguous Dh/2-float run of a single
// row: vector loads straight from HBM/L2 into registers, no LDS staging.
#include "common.h"
#include "internal.h"

namespace {

using sbk::f32x16;

struct AttnArgs {
  const float* qkv;       // [B,T,H,3*DH]
  const float* pos;       // [2T-1, H*DH]  = linear_pos(RelPosEncXL table)
  const float* bias_u;    // [H*DH]  (pos_bias_u storage viewed as (H,DH))
  const float* bias_v;
  const int32_t* key_len; // [B] or null
  float* out;             // [B,T,H*DH]
  float* attn;            // [B,H,T,T] or null
  int B, T, H, SP;
  float scale;
  // Dynamic Chunk attention mask (TransformerASR.py:47-103, make_transformer_src_mask): chunk > 0 restricts query i
  // (chunk c = i / chunk) to the keys [max(0, (c - left) * chunk), (c + 1) * chunk); left < 0 = unlimited left context.
  int chunk, left;
  int out_bf16;  // rope_flash_t_bf16_kernel only: `out` holds bf16 (the next contraction's operand) instead of fp32
};

// allowed key range [lo, hi) of query row i under the chunk mask and the key padding length
__device__ __forceinline__ void key_range(const AttnArgs& a, int i, int klen, int& lo, int& hi) {
  lo = 0;
  hi = klen;
  if (a.chunk > 0) {
    const int c = i / a.chunk;
    hi = min((c + 1) * a.chunk, klen);
    if (a.left >= 0) lo = max(0, (c - a.left) * a.chunk);
  }
}

template <int N>
__device__ __forceinline__ void load_run(float (&dst)[N], const float* __restrict__ p) {
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const float4 v = reinterpret_cast<const float4*>(p)[i];
      dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
  } else if constexpr (N % 2 == 0) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float2 v = reinterpret_cast<const float2*>(p)[i];
      dst[2 * i] = v.x; dst[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = p[i];
  }
}

// ROPE = true is RoPEMHA (nnet/attention.py:1167-1392): no position-score term; q and k are rotated
// pair-wise by the angle of their own frame on the way into the MFMA operands (a.pos = cosines
// [>=T][DH], a.bias_u = signed sines [>=T][DH] of PrecomputedRoPESinusoids, attention.py:955-1053;
// x'[c] = x[c]*cos[t][c] + x[c^1]*sin[t][c]).  The rotation pairs (2i, 2i+1) never straddle the two
// halves of the head dimension, so the half-split operand runs rotate in registers.
// NW = waves per workgroup (4 or 8).  The score strip pins one workgroup per CU once T' > ~350, so 8 waves
// put two on every SIMD behind the same strip: one wave's operand loads hide behind the other's MFMA chain.
template <int DH, bool ROPE, bool PF, int NW>
__global__ void __launch_bounds__(NW * 64) relpos_attn_kernel(AttnArgs a) {
  constexpr int DH2 = DH / 2;
  constexpr int QP = DH + 1;             // LDS pitch of the Q tiles (odd)
  constexpr int NC = (DH + 31) / 32;     // 32-wide column tiles of the context
  constexpr int NPART = NW / NC;         // waves sharing one column tile in P.V
  constexpr int NT = NW * 64;
  SBK_DYN_LDS(float, lds);
  float* Qu = lds;                       // [32][QP]   (q + u) * scale
  float* Qv = Qu + 32 * QP;              // [32][QP]   (q + v) * scale
  float* red = Qv + 32 * QP;             // [NW-NC][32][33] partial contexts
  float* S = red + (NW - NC) * 32 * 33;  // [32][SP]   scores -> probabilities

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 31, half = lane >> 5;
  const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, SP = a.SP, d = a.H * DH;
  const size_t row3 = (size_t)3 * d;
  const float* qkv_b = a.qkv + (size_t)b * T * row3 + (size_t)h * 3 * DH;

  for (int idx = tid; idx < 32 * DH; idx += NT) {
    const int i = idx / DH, c = idx % DH;
    const int row = min(i0 + i, T - 1);
    const float q = qkv_b[(size_t)row * row3 + c];
    if constexpr (ROPE) {
      const float qs = qkv_b[(size_t)row * row3 + (c ^ 1)];
      Qu[i * QP + c] = (q * a.pos[(size_t)row * DH + c] + qs * a.bias_u[(size_t)row * DH + c]) * a.scale;
    } else {
      Qu[i * QP + c] = (q + a.bias_u[h * DH + c]) * a.scale;
      Qv[i * QP + c] = (q + a.bias_v[h * DH + c]) * a.scale;
    }
  }
  __syncthreads();

  // ---- phase 1: S = AC + shifted BD, key tiles round-robin over the waves.  The operand runs of a
  // wave's NEXT key tile (K, P[rbase..], P[rbase+32..]; for RoPE: K, cosines, sines) are requested
  // before the MFMA chains of the current one, so their latency hides behind ~100 MFMAs.
  const int nkt = (T + 31) / 32;
  auto fetch = [&](int kt, float (&kr)[DH2], float (&q0)[DH2], float (&q1)[DH2]) {
    const int j0 = kt * 32;
    const int krow = min(j0 + jl, T - 1);
    load_run<DH2>(kr, qkv_b + (size_t)krow * row3 + DH + half * DH2);
    if constexpr (ROPE) {
      load_run<DH2>(q0, a.pos + (size_t)krow * DH + half * DH2);
      load_run<DH2>(q1, a.bias_u + (size_t)krow * DH + half * DH2);
    } else {
      const int rbase = (T - 1) - i0 - 31 + j0;
      const int prow0 = min(max(rbase + jl, 0), 2 * T - 2), prow1 = min(max(rbase + 32 + jl, 0), 2 * T - 2);
      load_run<DH2>(q0, a.pos + (size_t)prow0 * d + h * DH + half * DH2);
      load_run<DH2>(q1, a.pos + (size_t)prow1 * d + h * DH + half * DH2);
    }
  };
  auto score_tile = [&](int kt, float (&kreg)[DH2], float (&p0reg)[DH2], float (&p1reg)[DH2]) {
    const int j0 = kt * 32;
    if constexpr (ROPE) {  // p0reg / p1reg carry this frame's cosines / sines
#pragma unroll
      for (int s2 = 0; s2 < DH2; s2 += 2) {
        const float k0 = kreg[s2], k1 = kreg[s2 + 1];
        kreg[s2] = k0 * p0reg[s2] + k1 * p1reg[s2];
        kreg[s2 + 1] = k1 * p0reg[s2 + 1] + k0 * p1reg[s2 + 1];
      }
    }
    f32x16 acc;
    {  // AC = (Q+u) K^T
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < DH2; ++s) acc = sbk::mfma_32x32x2(Qu[jl * QP + s + half * DH2], kreg[s], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        S[i * SP + j0 + jl] = acc[r];
      }
    }
    sbk::wave_sync();
    if constexpr (!ROPE) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {  // BD: G = (Q+v) P[rbase + 32*pt ...]^T, added along the skew
        const int rl = pt * 32 + jl;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < DH2; ++s)
          acc = sbk::mfma_32x32x2(Qv[jl * QP + s + half * DH2], pt == 0 ? p0reg[s] : p1reg[s], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int jloc = rl + i - 31;
          if (jloc >= 0 && jloc < 32) S[i * SP + j0 + jloc] += acc[r];
        }
      }
      sbk::wave_sync();
    }
  };
  if constexpr (PF) {
    float kA[DH2], pA0[DH2], pA1[DH2], kB[DH2], pB0[DH2], pB1[DH2];
    if (wave < nkt) fetch(wave, kA, pA0, pA1);
    for (int kt = wave; kt < nkt; kt += 2 * NW) {
      const bool more = kt + NW < nkt;
      if (more) fetch(kt + NW, kB, pB0, pB1);
      score_tile(kt, kA, pA0, pA1);
      if (more) {
        if (kt + 2 * NW < nkt) fetch(kt + 2 * NW, kA, pA0, pA1);
        score_tile(kt + NW, kB, pB0, pB1);
      }
    }
  } else {
    float kA[DH2], pA0[DH2], pA1[DH2];
    for (int kt = wave; kt < nkt; kt += NW) {
      fetch(kt, kA, pA0, pA1);
      score_tile(kt, kA, pA0, pA1);
    }
  }
  __syncthreads();

  // ---- phase 2: exact softmax over valid keys, 32 / NW rows per wave
  int klen = T;
  if (a.key_len) klen = min(max(a.key_len[b], 1), T);
  const int kend = nkt * 32;
  for (int ii = 0; ii < 32 / NW; ++ii) {
    const int i = wave * (32 / NW) + ii;
    float* Srow = S + i * SP;
    int lo, hi;
    key_range(a, min(i0 + i, T - 1), klen, lo, hi);
    float m = -INFINITY;
    for (int j = lo + lane; j < hi; j += 64) m = fmaxf(m, Srow[j]);
    m = sbk::wave_max(m);
    float sum = 0.0f;
    for (int j = lo + lane; j < hi; j += 64) {
      const float e = expf(Srow[j] - m);
      Srow[j] = e;
      sum += e;
    }
    sum = sbk::wave_sum(sum);
    // (a row whose keys are all masked gets zero weights, like the reference's post-softmax masked_fill, :713-728)
    for (int j = lane; j < kend; j += 64) Srow[j] = (j >= lo && j < hi) ? Srow[j] / sum : 0.0f;
    if (a.attn && i0 + i < T) {
      float* arow = a.attn + (((size_t)b * a.H + h) * T + (i0 + i)) * T;
      for (int j = lane; j < T; j += 64) arow[j] = Srow[j];
    }
  }
  __syncthreads();

  // ---- phase 3: context = P . V ; wave -> (column tile, key range)
  {
    const int ct = wave % NC, part = wave / NC;
    const int kspan = (klen + 1) & ~1;
    int chunk = (kspan + NPART - 1) / NPART;
    chunk = (chunk + 1) & ~1;
    const int t_begin = min(part * chunk, kspan), t_end = min(t_begin + chunk, kspan);
    const int col = ct * 32 + jl;
    const bool col_ok = col < DH;
    const float* vbase = qkv_b + 2 * DH + (col_ok ? col : 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // 8 k-steps per round (8 V loads in flight per lane); the next round's operands are requested
    // before the current round's MFMAs
    auto fetch_pv = [&](int t0, float (&pv)[8], float (&vv)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 2 * u + half;
        const bool ok = t0 + 2 * u < t_end;
        vv[u] = (ok && col_ok) ? vbase[(size_t)min(t, T - 1) * row3] : 0.0f;
        pv[u] = ok ? S[jl * SP + min(t, SP - 1)] : 0.0f;
      }
    };
    float pvA[8], vvA[8], pvB[8], vvB[8];
    fetch_pv(t_begin, pvA, vvA);
    for (int t0 = t_begin; t0 < t_end; t0 += 32) {
      fetch_pv(t0 + 16, pvB, vvB);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = sbk::mfma_32x32x2(pvA[u], vvA[u], acc);
      fetch_pv(t0 + 32, pvA, vvA);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = sbk::mfma_32x32x2(pvB[u], vvB[u], acc);
    }
    if (part > 0) {
      float* dst = red + ((part - 1) * NC + ct) * 32 * 33;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + jl] = acc[r];
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        for (int p = 1; p < NPART; ++p) v += red[(((p - 1) * NC + ct) * 32 + i) * 33 + jl];
        if (col_ok && i0 + i < T) a.out[((size_t)b * T + i0 + i) * d + h * DH + col] = v;
      }
    }
  }
}


// (The first strip-free kernel -- one wave per 32-query tile with a 32 x 33 LDS score tile, rounds 1-2, later knobs 16 / 17 -- was
// replaced by the transposed-score kernels below in round 3: 608 -> 491 us at B = 64, T' = 440; 777 -> 617 us at B = 32, T' = 750
// (profiles/r03_stream_calibration_and_streamk_v1_sweep.log).  Removed in round 5.)

// ---------------------------------------------------------------------------------------------
// Attention without a position-score term (RoPEMHA; plain MHA = an identity rotation table), transposed scores.
// The wave computes S^T = K Q^T instead of Q K^T: in the MFMA result layout a lane then holds ONE query (its column)
// and 16 of the tile's 32 keys, so
//   * the softmax statistics of a query are lane-local (16 registers + one shuffle with the other half-wave),
//   * the probabilities are already the B operand of the second product O^T = V^T P^T: accumulator register r of
//     half-wave h holds key (r&3) + 8(r>>2) + 4h, which is exactly the k-slice MFMA number r contracts when V^T is
//     fed with the same key order -- P never leaves the registers,
//   * the running rescale exp(m_old - m_new) is one scalar per lane.
// No LDS, no wave barrier, no workgroup barrier: a wave is one 32-query tile walking the key tiles.
template <int DH>
__device__ __forceinline__ void rotate_run(float (&x)[DH / 2], const float* __restrict__ cs, const float* __restrict__ sn) {
  float c[DH / 2], sg[DH / 2];
  load_run<DH / 2>(c, cs);
  load_run<DH / 2>(sg, sn);
#pragma unroll
  for (int s2 = 0; s2 < DH / 2; s2 += 2) {
    const float x0 = x[s2], x1 = x[s2 + 1];
    x[s2] = x0 * c[s2] + x1 * sg[s2];
    x[s2 + 1] = x1 * c[s2 + 1] + x0 * sg[s2 + 1];
  }
}

template <int DH>
__global__ void __launch_bounds__(256, 2) rope_flash_t_kernel(AttnArgs a) {
  constexpr int DH2 = DH / 2;
  constexpr int NC = (DH + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 31, half = lane >> 5;
  const int i0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, d = a.H * DH;
  if (i0 >= T) return;
  const size_t row3 = (size_t)3 * d;
  const float* qkv_b = a.qkv + (size_t)b * T * row3 + (size_t)h * 3 * DH;
  const int qrow = min(i0 + jl, T - 1);  // this lane's query (B operand column / output row)

  const bool rot = a.pos != nullptr;  // no tables: plain scaled-dot-product attention (uniform for the launch)
  float qu[DH2];
  load_run<DH2>(qu, qkv_b + (size_t)qrow * row3 + half * DH2);
  if (rot) rotate_run<DH>(qu, a.pos + (size_t)qrow * DH + half * DH2, a.bias_u + (size_t)qrow * DH + half * DH2);
#pragma unroll
  for (int s = 0; s < DH2; ++s) qu[s] *= a.scale;

  int klen = T;
  if (a.key_len) klen = min(max(a.key_len[b], 1), T);
  const int nkt = (klen + 31) / 32;
  int lo, hi;  // allowed keys of this lane's query
  key_range(a, qrow, klen, lo, hi);
  int kt_begin = 0, kt_end = nkt;
  if (a.chunk > 0) {
    kt_end = min(nkt, ((min(i0 + 31, T - 1) / a.chunk + 1) * a.chunk + 31) / 32);
    if (a.left >= 0) kt_begin = max(0, (i0 / a.chunk - a.left) * a.chunk) / 32;
  }
  float m_run = -INFINITY, l_run = 0.0f;
  f32x16 o[NC];
#pragma unroll
  for (int ct = 0; ct < NC; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int j0 = kt * 32;
    float kreg[DH2];
    {
      const int krow = min(j0 + jl, T - 1);
      load_run<DH2>(kreg, qkv_b + (size_t)krow * row3 + DH + half * DH2);
      if (rot) rotate_run<DH>(kreg, a.pos + (size_t)krow * DH + half * DH2, a.bias_u + (size_t)krow * DH + half * DH2);
    }
    f32x16 acc;  // S^T: row = key (r&3) + 8(r>>2) + 4*half of the tile, column = query jl
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DH2; ++s) acc = sbk::mfma_32x32x2(kreg[s], qu[s], acc);
    // V^T operand of the second product, in the key order of the accumulator registers (requested before the softmax)
    float vv[NC][16];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const int col = ct * 32 + jl;
      const float* vbase = qkv_b + 2 * DH + (col < DH ? col : 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = min(j0 + (r & 3) + 8 * (r >> 2) + 4 * half, T - 1);
        vv[ct][r] = col < DH ? vbase[(size_t)t * row3] : 0.0f;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (!(key >= lo && key < hi)) acc[r] = -INFINITY;
      mx = fmaxf(mx, acc[r]);
    }
    mx = fmaxf(mx, sbk::shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = m_new == -INFINITY ? 1.0f : expf(m_run - m_new);
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = acc[r] == -INFINITY ? 0.0f : expf(acc[r] - m_new);
      sum += acc[r];
    }
    sum += sbk::shfl_xor(sum, 32);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) o[ct] = sbk::mfma_32x32x2(vv[ct][r], acc[r], o[ct]);
  }
  // O^T: column = this lane's query, row = channel (r&3) + 8(r>>2) + 4*half + 32*ct
  if (i0 + jl < T) {
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;  // no allowed key at all: zero context
    float* orow = a.out + ((size_t)b * T + i0 + jl) * d + h * DH;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < DH) orow[c] = o[ct][r] * inv;
      }
  }
}

// RelPosMHAXL with the same transposed scores.  The position term BD[i][j] = (q_i + v) . p[T-1-i+j] is the one thing that
// cannot stay in registers: G^T[r'][i] = p[rbase + r'] . (q_i + v) comes out of the MFMA with a lane owning one query
// column, and the row that query needs for key j is r' = 31 - i + j -- a per-lane shift.  So G^T goes through a 64-row
// ring in LDS (each key tile adds the 32 rows the previous one did not have; 16 stores per lane) and every lane
// gathers its 16 entries back (pitch 32: the 32 lanes of a half-wave hit 32 different banks).  Scores, softmax,
// probabilities and the context product are those of rope_flash_t_kernel.
template <int DH, bool E2 = true>  // E2: the softmax weights as 2^(.) on one v_exp_f32 (false: libm's expf -- the A/B of knob 60)
__global__ void __launch_bounds__(256, 2) relpos_flash_t_kernel(AttnArgs a) {
  constexpr int DH2 = DH / 2;
  constexpr int NC = (DH + 31) / 32;
  __shared__ float Gs[4][64][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 31, half = lane >> 5;
  const int i0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, d = a.H * DH;
  if (i0 >= T) return;
  const size_t row3 = (size_t)3 * d;
  const float* qkv_b = a.qkv + (size_t)b * T * row3 + (size_t)h * 3 * DH;
  const int qrow = min(i0 + jl, T - 1);
  float (*G)[32] = Gs[wave];

  float qu[DH2], qv[DH2];
  {
    float qr[DH2], bu[DH2], bv[DH2];
    load_run<DH2>(qr, qkv_b + (size_t)qrow * row3 + half * DH2);
    load_run<DH2>(bu, a.bias_u + h * DH + half * DH2);
    load_run<DH2>(bv, a.bias_v + h * DH + half * DH2);
#pragma unroll
    for (int s = 0; s < DH2; ++s) {
      qu[s] = (qr[s] + bu[s]) * a.scale;
      qv[s] = (qr[s] + bv[s]) * a.scale;
    }
  }
  int klen = T;
  if (a.key_len) klen = min(max(a.key_len[b], 1), T);
  const int nkt = (klen + 31) / 32;
  int lo, hi;
  key_range(a, qrow, klen, lo, hi);
  int kt_begin = 0, kt_end = nkt;
  if (a.chunk > 0) {
    kt_end = min(nkt, ((min(i0 + 31, T - 1) / a.chunk + 1) * a.chunk + 31) / 32);
    if (a.left >= 0) kt_begin = max(0, (i0 / a.chunk - a.left) * a.chunk) / 32;
  }
  float m_run = -INFINITY, l_run = 0.0f;
  f32x16 o[NC];
#pragma unroll
  for (int ct = 0; ct < NC; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;

  // G^T rows [rbase + 32*blk, +32) of the position table against the wave's queries -> ring rows of parity `slot`
  auto g_block = [&](int first_row, int slot) {
    float preg[DH2];
    const int prow = min(max(first_row + jl, 0), 2 * T - 2);
    load_run<DH2>(preg, a.pos + (size_t)prow * d + h * DH + half * DH2);
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DH2; ++s) g = sbk::mfma_32x32x2(preg[s], qv[s], g);
#pragma unroll
    for (int r = 0; r < 16; ++r) G[slot * 32 + (r & 3) + 8 * (r >> 2) + 4 * half][jl] = g[r];
  };
  // key tile kt needs rows rbase .. rbase+62 with rbase = T-1-i0-31+32*kt: its lower 32 rows are the upper 32 of tile
  // kt-1.  Ring slot of the block starting at rbase + 32*m: (kt + m) & 1.
  g_block((T - 1) - i0 - 31 + 32 * kt_begin, kt_begin & 1);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int j0 = kt * 32;
    const int rbase = (T - 1) - i0 - 31 + j0;
    float kreg[DH2];
    load_run<DH2>(kreg, qkv_b + (size_t)min(j0 + jl, T - 1) * row3 + DH + half * DH2);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DH2; ++s) acc = sbk::mfma_32x32x2(kreg[s], qu[s], acc);
    g_block(rbase + 32, (kt + 1) & 1);  // the upper 32 rows (the slot last read two tiles ago)
    float vv[NC][16];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const int col = ct * 32 + jl;
      const float* vbase = qkv_b + 2 * DH + (col < DH ? col : 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = min(j0 + (r & 3) + 8 * (r >> 2) + 4 * half, T - 1);
        vv[ct][r] = col < DH ? vbase[(size_t)t * row3] : 0.0f;
      }
    }
    sbk::wave_sync();  // the ring holds both blocks of this tile
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jc = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int rr = 31 - jl + jc;  // 0 .. 62: row of this tile's window for (query jl, key jc)
      acc[r] += G[(((kt & 1) + (rr >> 5)) & 1) * 32 + (rr & 31)][jl];
      const int key = j0 + jc;
      if (!(key >= lo && key < hi)) acc[r] = -INFINITY;
      mx = fmaxf(mx, acc[r]);
    }
    sbk::wave_sync();  // everybody has read the lower block before the next tile overwrites its slot
    mx = fmaxf(mx, sbk::shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    // e^(s - m) as 2^(s log2(e) - m log2(e)): one FMA + one v_exp_f32 per score (libm's expf: a scaling, the instruction and its range
    // fix-ups); a masked score (-inf) gives exactly 0; a row whose keys are all masked so far keeps m = -inf: guarded
    constexpr float kL2E = 1.44269504088896340736f;
    const bool none = m_new == -INFINITY;
    const float nm = none ? 0.0f : -m_new * kL2E;
    float alpha, sum = 0.0f;
    if constexpr (E2) {
      alpha = none ? 1.0f : sbk::exp2_raw(fmaf(m_run, kL2E, nm));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = none ? 0.0f : sbk::exp2_raw(fmaf(acc[r], kL2E, nm));
        sum += acc[r];
      }
    } else {
      alpha = none ? 1.0f : expf(m_run - m_new);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = acc[r] == -INFINITY ? 0.0f : expf(acc[r] - m_new);
        sum += acc[r];
      }
    }
    sum += sbk::shfl_xor(sum, 32);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (sbk::wave_any(alpha != 1.0f)) {  // (a new maximum is rare after the first tiles)
#pragma unroll
      for (int ct = 0; ct < NC; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) o[ct] = sbk::mfma_32x32x2(vv[ct][r], acc[r], o[ct]);
  }
  if (i0 + jl < T) {
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    float* orow = a.out + ((size_t)b * T + i0 + jl) * d + h * DH;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < DH) orow[c] = o[ct][r] * inv;
      }
  }
}

// The same walk with bf16 MFMA operands (v_mfma_f32_32x32x16_bf16: 8 k-values per lane and instruction) for the
// opt-in reduced-precision path: q, k, v and the probabilities are rounded to bf16 on their way into the matrix
// cores, scores / softmax statistics / context accumulate in fp32.  head_dim 64: four instructions per score tile and
// four per context update instead of 32 + 32.
__global__ void __launch_bounds__(256, 2) rope_flash_t_bf16_kernel(AttnArgs a) {
  constexpr int DH = 64;
  using sbk::bf16x8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 31, half = lane >> 5;
  const int i0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, d = a.H * DH;
  if (i0 >= T) return;
  const size_t row3 = (size_t)3 * d;
  const float* qkv_b = a.qkv + (size_t)b * T * row3 + (size_t)h * 3 * DH;
  const int qrow = min(i0 + jl, T - 1);

  // operand chunk of MFMA step t: channels 16t + 8*half .. +7 of a row (rotation pairs stay inside a chunk)
  auto load_chunks = [&](const float* row, int frame, float scale, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c0 = 16 * t + 8 * half;
      float x[8];
      load_run<8>(x, row + c0);
      if (a.pos != nullptr) {  // rotary tables given; otherwise plain attention
        float cs[8], sn[8];
        load_run<8>(cs, a.pos + (size_t)frame * DH + c0);
        load_run<8>(sn, a.bias_u + (size_t)frame * DH + c0);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float x0 = x[e], x1 = x[e + 1];
          x[e] = x0 * cs[e] + x1 * sn[e];
          x[e + 1] = x1 * cs[e + 1] + x0 * sn[e + 1];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] *= scale;
      dst[t] = sbk::cvt_bf16x8(x);
    }
  };
  bf16x8 qb[4];
  load_chunks(qkv_b + (size_t)qrow * row3, qrow, a.scale, qb);

  int klen = T;
  if (a.key_len) klen = min(max(a.key_len[b], 1), T);
  const int nkt = (klen + 31) / 32;
  int lo, hi;
  key_range(a, qrow, klen, lo, hi);
  int kt_begin = 0, kt_end = nkt;
  if (a.chunk > 0) {
    kt_end = min(nkt, ((min(i0 + 31, T - 1) / a.chunk + 1) * a.chunk + 31) / 32);
    if (a.left >= 0) kt_begin = max(0, (i0 / a.chunk - a.left) * a.chunk) / 32;
  }
  float m_run = -INFINITY, l_run = 0.0f;
  f32x16 o[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int j0 = kt * 32;
    bf16x8 kb[4];
    {
      const int krow = min(j0 + jl, T - 1);
      load_chunks(qkv_b + (size_t)krow * row3 + DH, krow, 1.0f, kb);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = sbk::mfma_32x32x16_bf16(kb[t], qb[t], acc);
    // V^T operands: step s contracts the 8 keys the accumulator registers 8s .. 8s+7 of this half-wave belong to
    bf16x8 vb[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float* vbase = qkv_b + 2 * DH + ct * 32 + jl;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * s + e;
          const int t = min(j0 + (r & 3) + 8 * (r >> 2) + 4 * half, T - 1);
          x[e] = vbase[(size_t)t * row3];
        }
        vb[ct][s] = sbk::cvt_bf16x8(x);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (!(key >= lo && key < hi)) acc[r] = -INFINITY;
      mx = fmaxf(mx, acc[r]);
    }
    mx = fmaxf(mx, sbk::shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = m_new == -INFINITY ? 1.0f : expf(m_run - m_new);
    float sum = 0.0f;
    bf16x8 pb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * s + e;
        x[e] = acc[r] == -INFINITY ? 0.0f : expf(acc[r] - m_new);
        sum += x[e];
      }
      pb[s] = sbk::cvt_bf16x8(x);
    }
    sum += sbk::shfl_xor(sum, 32);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) o[ct] = sbk::mfma_32x32x16_bf16(vb[ct][s], pb[s], o[ct]);
  }
  if (i0 + jl < T) {
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    if (a.out_bf16) {  // (uniform) four consecutive channels per register quad: 8-byte stores
      unsigned short* orow = reinterpret_cast<unsigned short*>(a.out) + ((size_t)b * T + i0 + jl) * d + h * DH;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint2 pk;
          pk.x = (unsigned)sbk::f32_to_bf16(o[ct][4 * q] * inv) | ((unsigned)sbk::f32_to_bf16(o[ct][4 * q + 1] * inv) << 16);
          pk.y = (unsigned)sbk::f32_to_bf16(o[ct][4 * q + 2] * inv) | ((unsigned)sbk::f32_to_bf16(o[ct][4 * q + 3] * inv) << 16);
          *reinterpret_cast<uint2*>(orow + ct * 32 + 8 * q + 4 * half) = pk;
        }
    } else {
      float* orow = a.out + ((size_t)b * T + i0 + jl) * d + h * DH;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) orow[ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = o[ct][r] * inv;
    }
  }
}

// ---- plain attention on bf16 q / k / v rows shared through LDS (the Whisper encoder under precision "bf16") ------------
// qkv [B,T,H,3,64] bf16 as written by the bf16-activation QKV contraction; vT [B,H,64,Tp] bf16 = the values transposed
// (v_transpose_bf16_kernel, Tp = T rounded up to 64, zero padded).  A workgroup = one (utterance, head) and 128 queries
// (four waves of 32); the K rows and V^T rows of a 64-key tile go global -> LDS by LDS-DMA once per workgroup (double
// buffer, 128-byte rows, the 16-byte slot s of row r lands in slot s ^ ((r>>1)&7): the ds_read_b128 operand fetch of the
// GEMM kernels) instead of once per wave from L2.  Scores transposed as in rope_flash_t_kernel: S^T = K Q^T with Q in
// registers; MFMA row i of a 32-key sub-tile holds key i with bits 2 and 3 swapped, so that the 8 probabilities a lane
// packs for MFMA step s of the context product belong to the 8 CONSECUTIVE keys 16s + 8*half .. +7 -- one
// ds_read_b128 of a V^T row.  Online softmax in fp32 (scores scaled after the MFMA), probabilities rounded to bf16,
// context accumulated in fp32 and written as bf16 (the output projection's operand).
struct AttnLdsArgs {
  const unsigned short* qkv;
  const unsigned short* vT;
  const int32_t* key_len;
  unsigned short* out;  // [B,T,H*64] bf16
  int B, T, Tp, H;
  float scale;
};

__global__ void __launch_bounds__(256) v_transpose_bf16_kernel(const unsigned short* __restrict__ qkv,
                                                               unsigned short* __restrict__ vT, int T, int Tp, int H) {
  __shared__ unsigned short tile[64][66];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const size_t row3 = (size_t)3 * H * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int t = e >> 6, c = e & 63;
    tile[t][c] = t0 + t < T ? qkv[((size_t)b * T + t0 + t) * row3 + (size_t)h * 192 + 128 + c] : (unsigned short)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e >> 6, t = e & 63;
    vT[(((size_t)b * H + h) * 64 + c) * Tp + t0 + t] = tile[t][c];
  }
}

__global__ void __launch_bounds__(256, 2) attn_lds_bf16_kernel(AttnLdsArgs a) {
  using sbk::bf16x8;
  constexpr int TILE = 64 * 32;  // floats: 64 rows x 128 bytes
  SBK_DYN_LDS(float, lds);       // [2 stages][K tile | V^T tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int jl = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, d = a.H * 64;
  const int i0 = blockIdx.x * 128 + wave * 32;
  const size_t row3 = (size_t)3 * d;
  const unsigned short* qkv_b = a.qkv + (size_t)b * T * row3 + (size_t)h * 192;
  const unsigned short* vT_b = a.vT + ((size_t)b * a.H + h) * 64 * a.Tp;
  int klen = T;
  if (a.key_len) klen = min(max(a.key_len[b], 1), T);
  const int ntile = (klen + 63) / 64;

  // Q operands: channels 16t + 8*half .. +7 of this lane's query row
  bf16x8 qb[4];
  {
    const unsigned short* qrow = qkv_b + (size_t)min(i0 + jl, T - 1) * row3;
#pragma unroll
    for (int t = 0; t < 4; ++t) qb[t] = *reinterpret_cast<const bf16x8*>(qrow + 16 * t + 8 * half);
  }
  // loader: wave w fills rows 16w .. 16w+15 of the K tile and of the V^T tile (two 8-row pieces each)
  const int lr = lane >> 3, ls = lane & 7;
  auto issue = [&](int tile, int stage) SBK_INLINE_LAMBDA {
    float* base = lds + stage * 2 * TILE;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int r = wave * 16 + pc * 8 + lr;  // row of the tile: a key (K) / a channel (V^T)
      const int sl = (ls ^ ((r >> 1) & 7)) * 8;  // source offset (bf16 elements) of the slot this lane fills
      const int key = min(tile * 64 + r, T - 1);  // keys past the utterance re-read its last row (masked below)
      sbk::glds16(reinterpret_cast<const float*>(qkv_b + (size_t)key * row3 + 64 + sl), base + (wave * 16 + pc * 8) * 32);
      sbk::glds16(reinterpret_cast<const float*>(vT_b + (size_t)r * a.Tp + tile * 64 + sl), base + TILE + (wave * 16 + pc * 8) * 32);
    }
  };
  f32x16 o[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[c][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  // MFMA row jl of a score sub-tile = key jl with bits 2 and 3 swapped
  const int krow = (jl & 19) | ((jl & 4) << 1) | ((jl & 8) >> 1);
  const int ksw = (krow >> 1) & 7, vsw = (jl >> 1) & 7;  // (rows + 32 have the same swizzle)

  issue(0, 0);
  sbk::vm_drain();
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < ntile) issue(kt + 1, stage ^ 1);
    const float* Kt = lds + stage * 2 * TILE;
    const float* Vt = Kt + TILE;
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 kb = *reinterpret_cast<const bf16x8*>(Kt + (32 * u + krow) * 32 + (((2 * t + half) ^ ksw) << 2));
        acc[u] = sbk::mfma_32x32x16_bf16(kb, qb[t], acc[u]);
      }
    }
    // acc[u][r]: key 64 kt + 32u + 16 (r>>3) + 8 half + (r&7), query jl.  The softmax runs on the RAW scores in base 2 (round 6):
    // p = 2^((s - m) c) with c = scale * log2(e) is one FMA + one v_exp_f32 per score where expf((s * scale) - m) was a multiply, a
    // subtraction and libm's exp (a scaling, the instruction and its range fix-ups); the running maximum m is kept in raw units.
    float mx = -INFINITY;
    const bool tail = (kt + 1) * 64 > klen;  // uniform: only the last tile has keys to mask
    if (tail) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 64 + 32 * u + 16 * (r >> 3) + 8 * half + (r & 7) >= klen) acc[u][r] = -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[u][r]);
    mx = fmaxf(mx, sbk::shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: key 64 kt is inside the utterance
    const float c2 = a.scale * 1.44269504088896340736f, mc = -m_new * c2;
    const float alpha = sbk::exp2_raw((m_run - m_new) * c2);  // (2^-inf = 0 at the first tile)
    float sum = 0.0f;
    bf16x8 pb[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          x[e] = sbk::exp2_raw(fmaf(acc[u][8 * s2 + e], c2, mc));
          sum += x[e];
        }
        pb[u][s2] = sbk::cvt_bf16x8(x);
      }
    sum += sbk::shfl_xor(sum, 32);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (sbk::wave_any(alpha != 1.0f)) {  // (after the first tiles a new maximum is rare: most tiles skip the 32 rescaling multiplies)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
    }
    // O^T[channel 32c + row][query] += V^T[channel][keys 32u + 16s + 8 half .. +7] . P^T
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 vb = *reinterpret_cast<const bf16x8*>(Vt + (32 * c + jl) * 32 + (((4 * u + 2 * s2 + half) ^ vsw) << 2));
          o[c] = sbk::mfma_32x32x16_bf16(vb, pb[u][s2], o[c]);
        }
    sbk::vm_drain();   // this wave's pieces of the next tile have landed ...
    __syncthreads();   // ... and everybody's; every wave is done with `stage`
  }
  if (i0 + jl < T) {
    const float inv = 1.0f / l_run;
    unsigned short* orow = a.out + ((size_t)b * T + i0 + jl) * d + h * 64;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 pk;
        pk.x = (unsigned)sbk::f32_to_bf16(o[c][4 * q] * inv) | ((unsigned)sbk::f32_to_bf16(o[c][4 * q + 1] * inv) << 16);
        pk.y = (unsigned)sbk::f32_to_bf16(o[c][4 * q + 2] * inv) | ((unsigned)sbk::f32_to_bf16(o[c][4 * q + 3] * inv) << 16);
        *reinterpret_cast<uint2*>(orow + c * 32 + 8 * q + 4 * half) = pk;
      }
  }
}

}  // namespace
namespace sbk {
int g_attn_exp2 = 1;  // key 60: 1 (default) = relpos_flash_t_kernel computes its softmax weights as 2^(.) on v_exp_f32, 0 = libm expf
}
namespace {
template <int DH, bool ROPE>
int launch_flash(const AttnArgs& a, hipStream_t st) {
  sbk::ProfScope prof(ROPE ? "rope_attention" : "relpos_attention", (ROPE ? 4.0 : 6.0) * a.B * a.H * (double)a.T * a.T * DH,
                      4.0 * a.B * a.T * (4.0 * a.H * DH) + 4.0 * (2.0 * a.T - 1) * a.H * DH, st);
  if constexpr (ROPE) {  // transposed scores, no LDS
    SBK_LAUNCH((rope_flash_t_kernel<DH>), dim3((a.T + 127) / 128, a.H, a.B), dim3(256), 0, st, a);
    return sbk::launch_status("sbk_rope_attention_f32");
  } else {  // transposed scores, position term through a 64-row LDS ring
    if (sbk::g_attn_exp2) {
      SBK_LAUNCH((relpos_flash_t_kernel<DH, true>), dim3((a.T + 127) / 128, a.H, a.B), dim3(256), 0, st, a);
    } else {
      SBK_LAUNCH((relpos_flash_t_kernel<DH, false>), dim3((a.T + 127) / 128, a.H, a.B), dim3(256), 0, st, a);
    }
    return sbk::launch_status("sbk_relpos_attention_f32");
  }
}

template <int DH, bool ROPE, bool PF, int NW>
int launch_attn_pf(const AttnArgs& a, hipStream_t st);

template <int DH, bool ROPE>
int launch_attn(const AttnArgs& a, hipStream_t st) {
  // Schedule (measured on MI355X, tools/microbench.py --attn, B=64 T'=440 Dh=64): once the score strip
  // pins one workgroup per CU (> 80 KB of LDS), 8 waves per workgroup are 1.43x (RelPos) / 1.28x (RoPE)
  // faster than 4; prefetching the next key tile's operands helps only RoPE (1.18x more; RelPos spills).
  if (!a.attn) return launch_flash<DH, ROPE>(a, st);  // (the strip kernel below serves the attention-weights output)
  const size_t lds4 = ((size_t)2 * 32 * (DH + 1) + 3 * 32 * 33 + (size_t)32 * a.SP) * sizeof(float);
  const bool eight = lds4 > 80 * 1024;
  const bool pf = ROPE && eight;
  if (eight) return pf ? launch_attn_pf<DH, ROPE, true, 8>(a, st) : launch_attn_pf<DH, ROPE, false, 8>(a, st);
  return pf ? launch_attn_pf<DH, ROPE, true, 4>(a, st) : launch_attn_pf<DH, ROPE, false, 4>(a, st);
}

template <int DH, bool ROPE, bool PF, int NW>
int launch_attn_pf(const AttnArgs& a, hipStream_t st) {
  constexpr int NCt = (DH + 31) / 32;
  const size_t lds = ((size_t)2 * 32 * (DH + 1) + (NW - NCt) * 32 * 33 + (size_t)32 * a.SP) * sizeof(float);
  if (lds > 160 * 1024) return sbk::fail(SBK_EINVAL, "relpos_attention: T=%d needs %zu B of LDS (max 160 KiB)", a.T, lds);
  if (lds > 64 * 1024) {
    hipError_t e = SBK_ALLOW_DYN_LDS((relpos_attn_kernel<DH, ROPE, PF, NW>), lds);
    if (e != hipSuccess) return sbk::fail((int)e, "relpos_attention: cannot raise the LDS window to %zu B", lds);
  }
  sbk::ProfScope prof(ROPE ? "rope_attention" : "relpos_attention", (ROPE ? 4.0 : 6.0) * a.B * a.H * (double)a.T * a.T * DH,
                      4.0 * a.B * a.T * (4.0 * a.H * DH) + 4.0 * (2.0 * a.T - 1) * a.H * DH, st);
  SBK_LAUNCH((relpos_attn_kernel<DH, ROPE, PF, NW>), dim3((a.T + 31) / 32, a.H, a.B), dim3(NW * 64), lds, st, a);
  return sbk::launch_status(ROPE ? "sbk_rope_attention_f32" : "sbk_relpos_attention_f32");
}

}  // namespace

namespace sbk {

int relpos_attention(const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                     const int32_t* key_len, float* out, float* attn, int B, int T, int H, int Dh, float scale,
                     hipStream_t st, int chunk, int left) {
  if (B == 0 || T == 0) return 0;
  const int SP = ((T + 31) / 32) * 32 + 1;  // odd pitch: the 32 rows of a P.V operand read hit 32 banks
  AttnArgs a{qkv, pos, bias_u, bias_v, key_len, out, attn, B, T, H, SP, scale, chunk, left};
  switch (Dh) {
    case 64: return launch_attn<64, false>(a, st);
    case 36: return launch_attn<36, false>(a, st);
    case 32: return launch_attn<32, false>(a, st);
    case 16: return launch_attn<16, false>(a, st);
    case 8: return launch_attn<8, false>(a, st);
    default: return fail(SBK_EINVAL, "relpos_attention: head_dim %d not instantiated (8,16,32,36,64)", Dh);
  }
}

int rope_attention(const float* qkv, const float* cosines, const float* sines, const int32_t* key_len, float* out,
                   float* attn, int B, int T, int H, int Dh, float scale, hipStream_t st, int chunk, int left) {
  if (B == 0 || T == 0) return 0;
  const int SP = ((T + 31) / 32) * 32 + 1;
  AttnArgs a{qkv, cosines, sines, nullptr, key_len, out, attn, B, T, H, SP, scale, chunk, left};
  switch (Dh) {
    case 64: return launch_attn<64, true>(a, st);
    case 36: return launch_attn<36, true>(a, st);
    case 32: return launch_attn<32, true>(a, st);
    case 16: return launch_attn<16, true>(a, st);
    case 8: return launch_attn<8, true>(a, st);
    default: return fail(SBK_EINVAL, "rope_attention: head_dim %d not instantiated (8,16,32,36,64)", Dh);
  }
}
}  // namespace sbk

extern "C" int sbk_rope_attention_f32(const float* qkv, const float* cosines, const float* sines,
                                      const int32_t* key_len, float* out, float* attn, int B, int T, int H, int Dh,
                                      int table_rows, float scale, int chunk_size, int left_chunks,
                                      sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(qkv && out && (cosines != nullptr) == (sines != nullptr), "rope_attention: null operand");
  SBK_REQUIRE(cosines || !attn, "rope_attention: plain attention (no rotary tables) does not return attention weights");
  SBK_REQUIRE(B >= 0 && T >= 0 && H > 0 && Dh > 0 && Dh % 2 == 0, "rope_attention: bad shape");
  SBK_REQUIRE(!cosines || table_rows >= T, "rope_attention: the sinusoid tables hold %d rows, T = %d", table_rows, T);
  SBK_REQUIRE(sbk::aligned16(qkv) && sbk::aligned16(cosines) && sbk::aligned16(sines),
              "rope_attention: operands must be 16-byte aligned");
  SBK_REQUIRE(chunk_size >= 0, "rope_attention: negative chunk size");
  return sbk::rope_attention(qkv, cosines, sines, key_len, out, attn, B, T, H, Dh, scale, sbk::as_stream(stream),
                             chunk_size, left_chunks);
}

namespace {
int rope_attention_bf16_impl(const float* qkv, const float* cosines, const float* sines, const int32_t* key_len, void* out,
                             int out_bf16, int B, int T, int H, int Dh, int table_rows, float scale, int chunk_size,
                             int left_chunks, sbk_stream_t stream);
}
extern "C" int sbk_rope_attention_bf16(const float* qkv, const float* cosines, const float* sines,
                                       const int32_t* key_len, float* out, int B, int T, int H, int Dh, int table_rows,
                                       float scale, int chunk_size, int left_chunks, sbk_stream_t stream) {
  return rope_attention_bf16_impl(qkv, cosines, sines, key_len, out, 0, B, T, H, Dh, table_rows, scale, chunk_size,
                                  left_chunks, stream);
}
extern "C" int sbk_rope_attention_bf16o(const float* qkv, const float* cosines, const float* sines,
                                        const int32_t* key_len, uint16_t* out, int B, int T, int H, int Dh, int table_rows,
                                        float scale, int chunk_size, int left_chunks, sbk_stream_t stream) {
  return rope_attention_bf16_impl(qkv, cosines, sines, key_len, out, 1, B, T, H, Dh, table_rows, scale, chunk_size,
                                  left_chunks, stream);
}
namespace {
int rope_attention_bf16_impl(const float* qkv, const float* cosines, const float* sines, const int32_t* key_len, void* out,
                             int out_bf16, int B, int T, int H, int Dh, int table_rows, float scale, int chunk_size,
                             int left_chunks, sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(qkv && out && (cosines != nullptr) == (sines != nullptr), "rope_attention_bf16: null operand");
  SBK_REQUIRE(B >= 0 && T >= 0 && H > 0 && Dh == 64, "rope_attention_bf16: head_dim 64 only (got %d)", Dh);
  SBK_REQUIRE(!cosines || table_rows >= T, "rope_attention_bf16: the sinusoid tables hold %d rows, T = %d", table_rows, T);
  SBK_REQUIRE(sbk::aligned16(qkv) && sbk::aligned16(cosines) && sbk::aligned16(sines),
              "rope_attention_bf16: operands must be 16-byte aligned");
  SBK_REQUIRE(chunk_size >= 0, "rope_attention_bf16: negative chunk size");
  hipStream_t st = sbk::as_stream(stream);
  AttnArgs a{qkv, cosines, sines, nullptr, key_len, reinterpret_cast<float*>(out), nullptr, B, T, H, 0, scale, chunk_size, left_chunks, out_bf16};
  sbk::ProfScope prof("rope_attention_bf16", 4.0 * B * H * (double)T * T * Dh, 4.0 * B * T * (3.0 * H * Dh) + (out_bf16 ? 2.0 : 4.0) * B * T * H * Dh, st);
  SBK_LAUNCH(rope_flash_t_bf16_kernel, dim3((T + 127) / 128, H, B), dim3(256), 0, st, a);
  return sbk::launch_status("sbk_rope_attention_bf16");
}
}  // namespace

extern "C" size_t sbk_attention_bf16io_workspace_bytes(int B, int T, int H) {
  return (size_t)B * H * 64 * ((T + 63) / 64 * 64) * sizeof(uint16_t);
}
extern "C" int sbk_attention_bf16io(const uint16_t* qkv, const int32_t* key_len, uint16_t* out, uint16_t* workspace, int B,
                                    int T, int H, int Dh, float scale, sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;
  SBK_REQUIRE(qkv && out && workspace, "attention_bf16io: null operand");
  SBK_REQUIRE(B > 0 && T > 0 && H > 0 && Dh == 64, "attention_bf16io: head_dim 64 only (got %d)", Dh);
  SBK_REQUIRE(sbk::aligned16(qkv) && sbk::aligned16(workspace) && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
              "attention_bf16io: operands must be 16-byte aligned");
  hipStream_t st = sbk::as_stream(stream);
  const int Tp = (T + 63) / 64 * 64;
  AttnLdsArgs a{qkv, workspace, key_len, out, B, T, Tp, H, scale};
  sbk::ProfScope prof("attention_bf16io", 4.0 * B * H * (double)T * T * Dh, 2.0 * B * T * (5.0 * H * Dh) + 2.0 * B * H * 64.0 * Tp, st);
  SBK_LAUNCH(v_transpose_bf16_kernel, dim3(Tp / 64, H, B), dim3(256), 0, st, qkv, workspace, T, Tp, H);
  int rc = sbk::launch_status("sbk_attention_bf16io (transpose)");
  if (rc) return rc;
  const size_t lds = (size_t)2 * 2 * 64 * 32 * sizeof(float);
  SBK_LAUNCH(attn_lds_bf16_kernel, dim3((T + 127) / 128, H, B), dim3(256), lds, st, a);
  return sbk::launch_status("sbk_attention_bf16io");
}

extern "C" int sbk_relpos_attention_f32(const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                                        const int32_t* key_len, float* out, float* attn, int B, int T, int H, int Dh,
                                        float scale, int chunk_size, int left_chunks, sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(qkv && pos && bias_u && bias_v && out, "relpos_attention: null operand");
  SBK_REQUIRE(B >= 0 && T >= 0 && H > 0 && Dh > 0, "relpos_attention: bad shape");
  SBK_REQUIRE(sbk::aligned16(qkv) && sbk::aligned16(pos), "relpos_attention: operands must be 16-byte aligned");
  SBK_REQUIRE(chunk_size >= 0, "relpos_attention: negative chunk size");
  return sbk::relpos_attention(qkv, pos, bias_u, bias_v, key_len, out, attn, B, T, H, Dh, scale,
                               sbk::as_stream(stream), chunk_size, left_chunks);
}
