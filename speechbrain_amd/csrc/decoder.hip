// Kernels of the KV-cached Transformer decoder step (TransformerASR.decode replaced).
//
// The reference recomputes the whole prefix every step (TransformerASR.py:426-473,
// seq2seq.py:1929-1934).  Here one step touches one new token per hypothesis:
//   embed_pos        x = emb[tok]*sqrt(d) + pe[step]                        (HBM-bound gather)
//   self_attn_step   causal MHA of the new token over its own prefix; K/V of
//                    earlier positions live in a slot-addressed cache
//                    [slot][pos][d] and are reached through the per-hypothesis
//                    ancestry table kv_slot[hyp][pos] (beam reordering moves
//                    4-byte slot ids, never K/V rows)
//   cross_attn_step  MHA over the encoder memory; the K/V projections of the
//                    memory are computed ONCE per utterance and all beams of
//                    an utterance are served by the same workgroup, so each
//                    K/V row is read once per step instead of once per beam
//   log_softmax_row  log_softmax(logits / temperature) * attn_weight
// All are HBM/L2-bandwidth kernels (a few flops per byte); the dense projections
// around them are csrc/gemm.hip launches.
#include "common.h"
#include "internal.h"

namespace {

// ---------------------------------------------------------------- embedding
__global__ void __launch_bounds__(256) embed_pos_kernel(const int32_t* __restrict__ tok, const float* __restrict__ emb,
                                                        const float* __restrict__ pe_row, float* __restrict__ x,
                                                        int n, int d, float scale, const int32_t* __restrict__ step_ptr) {
  const int i = blockIdx.x;
  if (step_ptr) pe_row += (size_t)step_ptr[0] * d;  // pe_row = row 0 of the table in that mode
  const float* e = emb + (size_t)tok[i] * d;
  for (int c = threadIdx.x; c < d; c += 256) x[(size_t)i * d + c] = e[c] * scale + pe_row[c];
}

// ---------------------------------------------------------------- self attention, one new token
struct SelfAttnArgs {
  const float* qkv;        // [n,3d] stacked (q | k | v) for the new token
  float* kcache;           // [nslot][Lmax][d]: slot-major, so the prefix of a hypothesis (its ancestors are the
                           // beams of its own utterance) lives in a few MB instead of one row per 2.6 MB stride
  float* vcache;
  const int32_t* kv_slot;  // [n][Lmax]: slot holding position p of hypothesis i (p < step)
  float* out;              // [n,d]
  int n, d, H, Dh, step, nslot, Lmax;
  float scale;
  const int32_t* step_ptr;  // non-null: the position comes from the device-side step counter
  // optional key padding mask (TransformerLM.make_masks, TransformerLM.py:165-187): key position p is
  // masked when the token fed at p equals pad_idx.  That token is key_tok[i*key_stride + p - key_shift]
  // (key_first for p < key_shift).  NULL = no mask.
  const int32_t* key_tok;
  int key_stride, key_shift, key_first, pad_idx;
  int group;  // > 1: hypotheses come in groups of `group` beams of one utterance -- the waves of a workgroup then
              // take different BEAMS of the same (utterance, head): beams share most of their ancestry, so the
              // workgroup's waves read the same cache rows at the same time (one trip to HBM, the rest from L1/L2)
};

// One wave per (hypothesis, head).  The 64 lanes form 4 position groups x 16 lanes; a group's 16
// lanes read one K (or V) head row as 16-byte pieces (a 256-byte row per load instruction and
// group, fully coalesced).  The ancestry slots are fetched once into LDS, and rows are requested
// four position-quads at a time, so the walk over the prefix costs a few dependent round trips
// instead of two per position.
__global__ void __launch_bounds__(256) self_attn_step_kernel(SelfAttnArgs a) {
  SBK_DYN_LDS(float, lds);  // [4 waves][2][Lmax_pad]: probabilities, slots
  if (a.step_ptr) a.step = a.step_ptr[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  const bool live = item < a.n * a.H;
  int i = live ? item / a.H : 0, h = live ? item % a.H : 0;
  if (live && a.group > 1) {  // item = ((utterance * H + head) * group + beam)
    const int per = a.H * a.group, u = item / per, rem = item % per;
    h = rem / a.group;
    i = u * a.group + rem % a.group;
  }
  const int d = a.d, Dh = a.Dh, L = a.step + 1;
  const int lpad = ((a.Lmax + 63) / 64) * 64;
  float* prob = lds + wave * 2 * lpad;
  int* slot = reinterpret_cast<int*>(prob + lpad);
  const float* q = a.qkv + (size_t)i * 3 * d + h * Dh;
  const float* knew = q + d;
  const float* vnew = q + 2 * d;
  const int pg = lane >> 4, cq = lane & 15;  // position group, 16-byte piece of the head row
  const bool vec = (Dh % 4 == 0) && Dh <= 64;
  // append this token's K/V head slice to the cache (slot = hypothesis index)
  if (live) {
    for (int c = lane; c < Dh; c += 64) {
      const size_t o = ((size_t)i * a.Lmax + a.step) * d + h * Dh + c;
      a.kcache[o] = knew[c];
      a.vcache[o] = vnew[c];
    }
  }
  for (int p = lane; p < a.step; p += 64) slot[p] = a.kv_slot[(size_t)i * a.Lmax + p];
  sbk::wave_sync();
  const size_t head_off = (size_t)h * Dh;
  auto krow = [&](int p) { return (p == a.step) ? knew : a.kcache + ((size_t)slot[p] * a.Lmax + p) * d + head_off; };
  auto vrow = [&](int p) { return (p == a.step) ? vnew : a.vcache + ((size_t)slot[p] * a.Lmax + p) * d + head_off; };
  const bool piece = vec && cq * 4 < Dh;
  float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (piece) {
    q4 = *reinterpret_cast<const float4*>(q + cq * 4);
    q4.x *= a.scale; q4.y *= a.scale; q4.z *= a.scale; q4.w *= a.scale;
  }
  // scores
  for (int p0 = 0; p0 < L; p0 += 16) {
    float sc[4];
    if (vec) {
      float4 kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 4 * u + pg;
        kv[u] = (p < L && piece) ? *reinterpret_cast<const float4*>(krow(p) + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) sc[u] = (q4.x * kv[u].x + q4.y * kv[u].y) + (q4.z * kv[u].z + q4.w * kv[u].w);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 4 * u + pg;
        float s = 0.0f;
        if (p < L) {
          const float* kp = krow(p);
          for (int c = cq; c < Dh; c += 16) s = fmaf(q[c] * a.scale, kp[c], s);
        }
        sc[u] = s;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = sc[u];
      s += sbk::shfl_xor(s, 1);
      s += sbk::shfl_xor(s, 2);
      s += sbk::shfl_xor(s, 4);
      s += sbk::shfl_xor(s, 8);
      const int p = p0 + 4 * u + pg;
      if (p < L && cq == 0) {
        if (a.key_tok) {
          const int tk = p < a.key_shift ? a.key_first : a.key_tok[(size_t)i * a.key_stride + p - a.key_shift];
          if (tk == a.pad_idx) s = -INFINITY;
        }
        prob[p] = s;
      }
    }
  }
  sbk::wave_sync();
  float m = -INFINITY;
  for (int p = lane; p < L; p += 64) m = fmaxf(m, prob[p]);
  m = sbk::wave_max(m);
  float sum = 0.0f;
  for (int p = lane; p < L; p += 64) {
    const float e = expf(prob[p] - m);
    prob[p] = e;
    sum += e;
  }
  sum = sbk::wave_sum(sum);
  for (int p = lane; p < L; p += 64) prob[p] = prob[p] / sum;
  sbk::wave_sync();
  // context: each position group accumulates its positions, then the 4 groups are summed
  if (vec) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = 0; p0 < L; p0 += 16) {
      float4 vv[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 4 * u + pg;
        const bool ok = p < L && piece;
        vv[u] = ok ? *reinterpret_cast<const float4*>(vrow(p) + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        w[u] = ok ? prob[p] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x = fmaf(w[u], vv[u].x, acc.x);
        acc.y = fmaf(w[u], vv[u].y, acc.y);
        acc.z = fmaf(w[u], vv[u].z, acc.z);
        acc.w = fmaf(w[u], vv[u].w, acc.w);
      }
    }
    float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] += sbk::shfl_xor(o[e], 16);
      o[e] += sbk::shfl_xor(o[e], 32);
    }
    if (live && pg == 0 && piece) *reinterpret_cast<float4*>(a.out + (size_t)i * d + head_off + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    for (int c0 = 0; c0 < Dh; c0 += 16) {
      const int c = c0 + cq;
      float acc = 0.0f;
      for (int p = pg; p < L; p += 4)
        if (c < Dh) acc = fmaf(prob[p], vrow(p)[c], acc);
      acc += sbk::shfl_xor(acc, 16);
      acc += sbk::shfl_xor(acc, 32);
      if (live && pg == 0 && c < Dh) a.out[(size_t)i * d + head_off + c] = acc;
    }
  }
}

// ---------------------------------------------------------------- cross attention, all beams of an utterance
constexpr int kQT = 16;   // queries (beams) served per workgroup
constexpr int kFC = 128;  // memory frames per workgroup (flash-decoding style split of the memory)

struct CrossAttnArgs {
  const float* q;         // [n,d]   n = B*beam, hypothesis i belongs to utterance i / beam
  const float* kv;        // head_major = 0: [B,T,2d] per frame K (d) then V (d) (the projection GEMM's output);
                          // head_major = 1: [B,H,T,2*Dh] per (utterance, head) K|V rows back to back, so a
                          // workgroup's chunk of the memory is ONE contiguous run of HBM
  const int32_t* enc_len; // [B]
  float* out;             // [n,d]
  float* part;            // [B,H,NS,kQT,DH+2] partial (context, max, sum) when NS > 1
  int B, T, d, H, Dh, beam, NS;
  float scale;
  int head_major;
  int fc;  // memory frames per workgroup of the frame-per-thread kernel (64, 128 or 256)
  int32_t* cnt;  // [B] arrival tickets of the LDS-DMA kernel's runs (zero between launches), nullptr: cross_merge_kernel merges
};

// element offsets of (utterance b, head h): base of frame 0, frame stride, K -> V distance
struct KvView {
  size_t base;
  int row, voff;
};
__device__ __forceinline__ KvView kv_view(const CrossAttnArgs& a, int b, int h, int DH) {
  if (a.head_major) return {((size_t)b * a.H + h) * a.T * 2 * DH, 2 * DH, DH};
  return {(size_t)b * a.T * 2 * a.d + (size_t)h * DH, 2 * a.d, a.d};
}

// [B,T,2d] -> [B,H,T,2*Dh]: once per utterance batch and layer, after the K/V projection GEMM.
__global__ void __launch_bounds__(256) kv_head_major_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                            int T, int H, int Dh4, long total4) {
  // one float4 per thread; consecutive threads walk the DESTINATION (fully coalesced writes, 16*Dh-byte read runs)
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total4; o += (long)gridDim.x * 256) {
    const int c = (int)(o % Dh4);
    long r = o / Dh4;
    const int part = (int)(r & 1);
    r >>= 1;
    const int t = (int)(r % T);
    r /= T;
    const int h = (int)(r % H);
    const long b = r / H;
    dst[o] = src[((b * T + t) * 2 + part) * (long)H * Dh4 + (long)h * Dh4 + c];
  }
}

// grid (NS, H, B x query tiles).  A workgroup scores kFC memory frames against every beam of one
// (utterance, head): thread <-> (frame, beam parity), the K row is held in registers and reused by
// the beams; probabilities go through LDS; the context pass reads each V row once for all beams.
// With NS > 1 the workgroup emits (un-normalised context, running max, sum) and cross_merge_kernel
// combines the splits exactly like an online softmax.
template <int DH, int FC>  // FC = memory frames per workgroup (128 or 256)
__global__ void __launch_bounds__(256) cross_attn_step_kernel(CrossAttnArgs a) {
  __shared__ float qs[kQT][DH];
  __shared__ float S[kQT][FC + 1];
  __shared__ float red[4][kQT][DH];
  __shared__ float mx[kQT], sm[kQT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, h = blockIdx.y;
  const int qtiles = (a.beam + kQT - 1) / kQT;
  const int b = blockIdx.z / qtiles, q0 = (blockIdx.z % qtiles) * kQT;
  const int nq = min(kQT, a.beam - q0);
  const int T = a.T, d = a.d;
  const int klen = min(max(a.enc_len[b], 1), T);
  const int per = ((klen + a.NS - 1) / a.NS + 3) & ~3;
  const int t0 = split * per, t1 = min(klen, t0 + per);
  const int nf = max(0, t1 - t0);
  const KvView kvv = kv_view(a, b, h, DH);
  const float* kvb = a.kv + kvv.base;

  for (int idx = tid; idx < kQT * DH; idx += 256) {
    const int j = idx / DH, c = idx % DH;
    qs[j][c] = j < nq ? a.q[((size_t)b * a.beam + q0 + j) * d + h * DH + c] * a.scale : 0.0f;
  }
  __syncthreads();
  // scores: per needs <= FC (the launcher chooses NS accordingly)
  {
    const int f = tid & (FC - 1), par = tid / FC;  // 256 / FC threads per frame, beams split between them
    if (f < nf) {
      float kr[DH];
      const float* kp = kvb + (size_t)(t0 + f) * kvv.row;
#pragma unroll
      for (int c = 0; c < DH; ++c) kr[c] = kp[c];
      for (int j = par; j < nq; j += 256 / FC) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < DH; ++c) s = fmaf(qs[j][c], kr[c], s);
        S[j][f] = s;
      }
    }
  }
  __syncthreads();
  for (int j = wave; j < nq; j += 4) {  // one wave per query row
    float m = -INFINITY;
    for (int f = lane; f < nf; f += 64) m = fmaxf(m, S[j][f]);
    m = sbk::wave_max(m);
    float sum = 0.0f;
    for (int f = lane; f < nf; f += 64) {
      const float e = expf(S[j][f] - m);
      S[j][f] = e;
      sum += e;
    }
    sum = sbk::wave_sum(sum);
    if (a.NS == 1)
      for (int f = lane; f < nf; f += 64) S[j][f] = S[j][f] / sum;
    if (lane == 0) {
      mx[j] = m;
      sm[j] = sum;
    }
  }
  __syncthreads();
  {  // context: wave <-> quarter of the frames, lane <-> channel; 8 V rows in flight per lane
    float acc[kQT];
#pragma unroll
    for (int j = 0; j < kQT; ++j) acc[j] = 0.0f;
    const int c = lane;
    if (c < DH) {
      const float* vcol = kvb + (size_t)t0 * kvv.row + kvv.voff + c;
      for (int f0 = wave; f0 < nf; f0 += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + 4 * u;
          v[u] = f < nf ? vcol[(size_t)f * kvv.row] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + 4 * u;
          if (f < nf) {
#pragma unroll
            for (int j = 0; j < kQT; ++j) acc[j] = fmaf(S[j][f], v[u], acc[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kQT; ++j) red[wave][j][c] = acc[j];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < nq * DH; idx += 256) {
    const int j = idx / DH, c = idx % DH;
    const float v = (red[0][j][c] + red[1][j][c]) + (red[2][j][c] + red[3][j][c]);
    if (a.NS == 1) {
      a.out[((size_t)b * a.beam + q0 + j) * d + h * DH + c] = v;
    } else {
      float* pp = a.part + ((((size_t)b * a.H + h) * a.NS + split) * a.beam + q0 + j) * (DH + 2);
      pp[c] = v;
      if (c == 0) {
        pp[DH] = nf > 0 ? mx[j] : -INFINITY;
        pp[DH + 1] = nf > 0 ? sm[j] : 0.0f;
      }
    }
  }
}

// Same work, laid out for HBM streaming (head_dim 64 / 32 / 16): the K/V rows of a (utterance, head)
// are 4*DH-byte runs 8*d bytes apart, re-read from HBM every step (B*T'*2d*4 B per layer: 115 MB at
// B=64 -- far beyond L2).  Here LPR = DH/4 lanes read one row as 16-byte pieces, so a wave instruction
// covers 64/LPR whole rows, and a wave requests ALL K and V rows of its 32 frames before any arithmetic
// (16 KB in flight per wave, the workgroup's whole 64 KB chunk at once).  Scores are 4-wide partial
// dots reduced over the LPR lanes by shuffles; the context pass keeps a float4 per beam and folds the
// row groups by shuffles, then the 4 waves through LDS.
template <int DH>
__global__ void __launch_bounds__(256) cross_attn_rows_kernel(CrossAttnArgs a) {
  constexpr int LPR = DH / 4, RPI = 64 / LPR, U = 32 / RPI;
  static_assert(kFC == 128 && LPR * 4 == DH && RPI * LPR == 64 && U * RPI == 32 && (kQT * DH) % 256 == 0, "cross_attn_rows: layout");
  __shared__ __attribute__((aligned(16))) float qs[kQT][DH];
  __shared__ float S[kQT][kFC + 1];
  __shared__ float red[4][kQT][DH];
  __shared__ float mx[kQT], sm[kQT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, h = blockIdx.y;
  const int qtiles = (a.beam + kQT - 1) / kQT;
  const int b = blockIdx.z / qtiles, q0 = (blockIdx.z % qtiles) * kQT;
  const int nq = min(kQT, a.beam - q0);
  const int T = a.T, d = a.d;
  const int klen = min(max(a.enc_len[b], 1), T);
  const int per = ((klen + a.NS - 1) / a.NS + 3) & ~3;
  const int t0 = split * per, t1 = min(klen, t0 + per);
  const int nf = max(0, t1 - t0);
  const int rg = lane / LPR, cq = lane % LPR;
  const KvView kvv = kv_view(a, b, h, DH);
  const float* kvb = a.kv + kvv.base + cq * 4;

  // request order = completion order (vmcnt): queries, then K rows, then V rows, so the score pass can
  // start while the V rows are still in flight
  constexpr int QL = kQT * DH / 256;
  float qv[QL];
#pragma unroll
  for (int e = 0; e < QL; ++e) {
    const int idx = tid + e * 256, j = idx / DH, c = idx % DH;
    qv[e] = a.q[((size_t)b * a.beam + q0 + min(j, nq - 1)) * d + h * DH + c];  // rows >= nq: duplicates, never used
  }
  float4 k4[U], v4[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int f = wave * 32 + u * RPI + rg;
    k4[u] = *reinterpret_cast<const float4*>(kvb + (size_t)min(t0 + f, T - 1) * kvv.row);  // rows >= nf: loaded, never used
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int f = wave * 32 + u * RPI + rg;
    v4[u] = *reinterpret_cast<const float4*>(kvb + (size_t)min(t0 + f, T - 1) * kvv.row + kvv.voff);
  }
#pragma unroll
  for (int e = 0; e < QL; ++e) {
    const int idx = tid + e * 256;
    qs[idx / DH][idx % DH] = qv[e] * a.scale;
  }
  __syncthreads();
  {
    float4 q4[kQT];
#pragma unroll
    for (int j = 0; j < kQT; ++j) q4[j] = *reinterpret_cast<const float4*>(&qs[j][cq * 4]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = wave * 32 + u * RPI + rg;
#pragma unroll
      for (int j = 0; j < kQT; ++j) {
        if (j < nq) {
          float p = fmaf(q4[j].x, k4[u].x, fmaf(q4[j].y, k4[u].y, fmaf(q4[j].z, k4[u].z, q4[j].w * k4[u].w)));
          p = sbk::group_sum<LPR>(p);
          if (cq == 0 && f < nf) S[j][f] = p;
        }
      }
    }
  }
  __syncthreads();
  for (int j = wave; j < nq; j += 4) {  // one wave per query row
    float m = -INFINITY;
    for (int f = lane; f < nf; f += 64) m = fmaxf(m, S[j][f]);
    m = sbk::wave_max(m);
    float sum = 0.0f;
    for (int f = lane; f < nf; f += 64) {
      const float e = expf(S[j][f] - m);
      S[j][f] = e;
      sum += e;
    }
    sum = sbk::wave_sum(sum);
    if (a.NS == 1)
      for (int f = lane; f < nf; f += 64) S[j][f] = S[j][f] / sum;
    if (lane == 0) {
      mx[j] = m;
      sm[j] = sum;
    }
  }
  __syncthreads();
  {
    float4 acc[kQT];
#pragma unroll
    for (int j = 0; j < kQT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = wave * 32 + u * RPI + rg;
      if (f < nf) {
#pragma unroll
        for (int j = 0; j < kQT; ++j) {
          if (j < nq) {
            const float p = S[j][f];
            acc[j].x = fmaf(p, v4[u].x, acc[j].x);
            acc[j].y = fmaf(p, v4[u].y, acc[j].y);
            acc[j].z = fmaf(p, v4[u].z, acc[j].z);
            acc[j].w = fmaf(p, v4[u].w, acc[j].w);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kQT; ++j) {
      if (j < nq) {
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) {
          acc[j].x += sbk::shfl_xor(acc[j].x, m);
          acc[j].y += sbk::shfl_xor(acc[j].y, m);
          acc[j].z += sbk::shfl_xor(acc[j].z, m);
          acc[j].w += sbk::shfl_xor(acc[j].w, m);
        }
        if (rg == 0) {
          red[wave][j][cq * 4] = acc[j].x;
          red[wave][j][cq * 4 + 1] = acc[j].y;
          red[wave][j][cq * 4 + 2] = acc[j].z;
          red[wave][j][cq * 4 + 3] = acc[j].w;
        }
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < nq * DH; idx += 256) {
    const int j = idx / DH, c = idx % DH;
    const float v = (red[0][j][c] + red[1][j][c]) + (red[2][j][c] + red[3][j][c]);
    if (a.NS == 1) {
      a.out[((size_t)b * a.beam + q0 + j) * d + h * DH + c] = v;
    } else {
      float* pp = a.part + ((((size_t)b * a.H + h) * a.NS + split) * a.beam + q0 + j) * (DH + 2);
      pp[c] = v;
      if (c == 0) {
        pp[DH] = nf > 0 ? mx[j] : -INFINITY;
        pp[DH + 1] = nf > 0 ? sm[j] : 0.0f;
      }
    }
  }
}

// MFMA formulation (head_dim 64 / 32): the frame-per-thread kernel above spends its time in the LDS pipe
// (one broadcast read of q per FMA in the score pass, one read of every probability per beam and
// frame in the context pass).  Here the score tile S[32 frames x 32 beams] = K . q^T is one chain of
// v_mfma_f32_32x32x2 per wave with BOTH operands in registers (lane (r, half) holds the `half` side of
// frame row r / of beam row r -- the same contiguous-run trick as csrc/gemm.hip), and the context
// P . V feeds P from LDS once per MFMA step (16 frames x 32 beams per read) and V straight from L2 in
// 128-byte rows.  Up to 32 beams per workgroup.
constexpr int kQT2 = 32;

template <int DH>
__global__ void __launch_bounds__(256) cross_attn_mfma_kernel(CrossAttnArgs a) {
  constexpr int DH2 = DH / 2, NC = DH / 32, NPART = 4 / NC;
  static_assert(DH == 64 || DH == 32, "cross_attn_mfma: head_dim");
  __shared__ float S[kQT2][kFC + 1];
  __shared__ float red[3][kQT2][33];
  __shared__ float mx[kQT2], sm[kQT2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 31, half = lane >> 5;
  const int split = blockIdx.x, h = blockIdx.y;
  const int qtiles = (a.beam + kQT2 - 1) / kQT2;
  const int b = blockIdx.z / qtiles, q0 = (blockIdx.z % qtiles) * kQT2;
  const int nq = min(kQT2, a.beam - q0);
  const int T = a.T, d = a.d;
  const int klen = min(max(a.enc_len[b], 1), T);
  const int per = ((klen + a.NS - 1) / a.NS + 3) & ~3;
  const int t0 = split * per, t1 = min(klen, t0 + per);
  const int nf = max(0, t1 - t0);
  const KvView kvv = kv_view(a, b, h, DH);
  const float* kvb = a.kv + kvv.base;

  // context work of this wave: (32-column tile ct, frame range part).  Its V operands -- one value per
  // lane and k-step -- are requested NOW, together with the K and q runs of the score pass, so the whole
  // workgroup pays one memory round trip instead of one per phase (371 -> 315 ms at B=128; the same
  // hoisting made the frame-per-thread kernel slower, 303 -> 353 ms, and was not kept there).
  constexpr int SPAN = kFC / NPART, VSTEPS = SPAN / 2;
  const int ct = wave % NC, part = wave / NC;
  const int f_begin = part * SPAN, f_end = min(nf, f_begin + SPAN);
  float vall[VSTEPS];
  {
    const float* vbase = kvb + kvv.voff + ct * 32 + jl;
#pragma unroll
    for (int u = 0; u < VSTEPS; ++u) {
      const int f = f_begin + 2 * u + half;
      vall[u] = vbase[(size_t)min(t0 + f, T - 1) * kvv.row];  // rows >= f_end: loaded, multiplied by p = 0
    }
  }
  {  // scores of this wave's 32 frames against all beams
    float kreg[DH2], qreg[DH2];
    const int f = wave * 32 + jl;
    const float* kp = kvb + (size_t)min(t0 + f, T - 1) * kvv.row + half * DH2;
    const float* qp = a.q + ((size_t)b * a.beam + q0 + min(jl, nq - 1)) * d + h * DH + half * DH2;
#pragma unroll
    for (int s4 = 0; s4 < DH2; s4 += 4) {
      const float4 kv4 = *reinterpret_cast<const float4*>(kp + s4);
      const float4 qv4 = *reinterpret_cast<const float4*>(qp + s4);
      kreg[s4] = kv4.x; kreg[s4 + 1] = kv4.y; kreg[s4 + 2] = kv4.z; kreg[s4 + 3] = kv4.w;
      qreg[s4] = qv4.x; qreg[s4 + 1] = qv4.y; qreg[s4 + 2] = qv4.z; qreg[s4 + 3] = qv4.w;
    }
    const float qs = jl < nq ? a.scale : 0.0f;
    sbk::f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DH2; ++s) acc = sbk::mfma_32x32x2(kreg[s], qreg[s] * qs, acc);
    // acc[r] = S[frame wave*32 + i(r)][beam jl]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int fr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (jl < nq && fr < nf) S[jl][fr] = acc[r];
    }
  }
  __syncthreads();
  for (int j = wave; j < nq; j += 4) {  // one wave per query row
    float m = -INFINITY;
    for (int f = lane; f < nf; f += 64) m = fmaxf(m, S[j][f]);
    m = sbk::wave_max(m);
    float sum = 0.0f;
    for (int f = lane; f < nf; f += 64) {
      const float e = expf(S[j][f] - m);
      S[j][f] = e;
      sum += e;
    }
    sum = sbk::wave_sum(sum);
    if (a.NS == 1)
      for (int f = lane; f < nf; f += 64) S[j][f] = S[j][f] / sum;
    if (lane == 0) {
      mx[j] = m;
      sm[j] = sum;
    }
  }
  __syncthreads();
  {  // context: P (LDS) . V (registers since the top of the kernel)
    sbk::f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
    for (int u = 0; u < VSTEPS; ++u) {
      const int f = f_begin + 2 * u + half;
      const float pv = (f < f_end && jl < nq) ? S[jl][f] : 0.0f;
      const float vv = f < f_end ? vall[u] : 0.0f;
      o = sbk::mfma_32x32x2(pv, vv, o);
    }
    // o[r] = ctx[beam i(r)][column ct*32 + jl] over this wave's frames
    if (part > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(part - 1) * NC + ct][(r & 3) + 8 * (r >> 2) + 4 * half][jl] = o[r];
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = o[r];
#pragma unroll
        for (int p = 1; p < NPART; ++p) v += red[(p - 1) * NC + ct][j][jl];
        if (j < nq) {
          const int c = ct * 32 + jl;
          if (a.NS == 1) {
            a.out[((size_t)b * a.beam + q0 + j) * d + h * DH + c] = v;
          } else {
            float* pp = a.part + ((((size_t)b * a.H + h) * a.NS + split) * a.beam + q0 + j) * (DH + 2);
            pp[c] = v;
            if (c == 0) {
              pp[DH] = nf > 0 ? mx[j] : -INFINITY;
              pp[DH + 1] = nf > 0 ? sm[j] : 0.0f;
            }
          }
        }
      }
    }
  }
}

// Streaming formulation for d = 512 (Conformer-L: 8 heads x 64): ONE WAVE walks a run of memory frames by itself --
// no LDS, no barrier.  A frame's K row (all heads, 2 KB) is one fully coalesced wave load of 32 bytes per lane, so
// lane l owns channels 8l .. 8l+7 = head l/8; it keeps the matching 8 channels of every beam's query and context in
// registers (the queries sit in a per-wave LDS slab).  Per frame and beam: 8 FMAs, a 3-step DPP sum over the 8 lanes of
// the head, and -- once per 2 frames -- one flash-style rescale (running max / sum per beam, replicated in the head's
// lanes).  K and V of the next 2 frames (8 KB per wave) are requested before the current 2 are consumed.  Partials go through the same
// (context, max, sum) buffer as the other variants and are combined by cross_merge_kernel.
template <int NQ>
__global__ void __launch_bounds__(256) cross_attn_stream_kernel(CrossAttnArgs a, int fpw) {
  __shared__ __attribute__((aligned(16))) float qs[4][NQ][512];  // each wave's own queries (scaled), 20 KB per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  const int qtiles = (a.beam + NQ - 1) / NQ;
  const int item = gw / a.NS, split = gw % a.NS;
  const int b = item / qtiles, q0 = (item % qtiles) * NQ;
  if (b >= a.B) return;  // whole wave; nothing below synchronises across waves
  const int nq = min(NQ, a.beam - q0);
  const int d = a.d, T = a.T, DH = a.Dh;
  const int klen = min(max(a.enc_len[b], 1), T);
  const int t0 = split * fpw, t1 = min(klen, t0 + fpw);
  const int h = (lane * 8) / DH, c0 = (lane * 8) % DH;

  float acc[NQ][8], m[NQ], l[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const float* qp = a.q + ((size_t)b * a.beam + q0 + min(j, nq - 1)) * d + lane * 8;
    float4 x0 = *reinterpret_cast<const float4*>(qp), x1 = *reinterpret_cast<const float4*>(qp + 4);
    x0.x *= a.scale; x0.y *= a.scale; x0.z *= a.scale; x0.w *= a.scale;
    x1.x *= a.scale; x1.y *= a.scale; x1.z *= a.scale; x1.w *= a.scale;
    *reinterpret_cast<float4*>(&qs[wave][j][lane * 8]) = x0;
    *reinterpret_cast<float4*>(&qs[wave][j][lane * 8 + 4]) = x1;
    m[j] = -INFINITY;
    l[j] = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.0f;
  }
  sbk::wave_sync();
  const float* kvb = a.kv + (size_t)b * T * 2 * d + lane * 8;
  // two register buffers of two frames each (K and V, 8 channels per lane): one is consumed while the other lands
  float4 bufK[2][2][2], bufV[2][2][2];
  auto fetch = [&](int t, int which) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* row = kvb + (size_t)min(t + u, T - 1) * 2 * d;  // frames >= t1: loaded, weighted 0
      bufK[which][u][0] = *reinterpret_cast<const float4*>(row);
      bufK[which][u][1] = *reinterpret_cast<const float4*>(row + 4);
      bufV[which][u][0] = *reinterpret_cast<const float4*>(row + d);
      bufV[which][u][1] = *reinterpret_cast<const float4*>(row + d + 4);
    }
  };
  auto consume = [&](int t, int which) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const float4 qa = *reinterpret_cast<const float4*>(&qs[wave][j][lane * 8]);
      const float4 qb = *reinterpret_cast<const float4*>(&qs[wave][j][lane * 8 + 4]);
      float sc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 ka = bufK[which][u][0], kb = bufK[which][u][1];
        float p = qa.x * ka.x;
        p = fmaf(qa.y, ka.y, p); p = fmaf(qa.z, ka.z, p); p = fmaf(qa.w, ka.w, p);
        p = fmaf(qb.x, kb.x, p); p = fmaf(qb.y, kb.y, p); p = fmaf(qb.z, kb.z, p); p = fmaf(qb.w, kb.w, p);
        p = sbk::group_sum<8>(p);  // the 8 lanes of this head (DH = 64)
        sc[u] = (t + u < t1) ? p : -INFINITY;
      }
      const float mn = fmaxf(m[j], fmaxf(sc[0], sc[1]));  // finite: frame t is valid
      const float al = expf(m[j] - mn), w0 = expf(sc[0] - mn), w1 = expf(sc[1] - mn);
      l[j] = fmaf(l[j], al, w0 + w1);
      m[j] = mn;
      const float4 va0 = bufV[which][0][0], vb0 = bufV[which][0][1], va1 = bufV[which][1][0], vb1 = bufV[which][1][1];
      acc[j][0] = fmaf(w1, va1.x, fmaf(w0, va0.x, acc[j][0] * al));
      acc[j][1] = fmaf(w1, va1.y, fmaf(w0, va0.y, acc[j][1] * al));
      acc[j][2] = fmaf(w1, va1.z, fmaf(w0, va0.z, acc[j][2] * al));
      acc[j][3] = fmaf(w1, va1.w, fmaf(w0, va0.w, acc[j][3] * al));
      acc[j][4] = fmaf(w1, vb1.x, fmaf(w0, vb0.x, acc[j][4] * al));
      acc[j][5] = fmaf(w1, vb1.y, fmaf(w0, vb0.y, acc[j][5] * al));
      acc[j][6] = fmaf(w1, vb1.z, fmaf(w0, vb0.z, acc[j][6] * al));
      acc[j][7] = fmaf(w1, vb1.w, fmaf(w0, vb0.w, acc[j][7] * al));
    }
  };
  if (t0 < t1) fetch(t0, 0);
  for (int t = t0; t < t1; t += 4) {
    if (t + 2 < t1) fetch(t + 2, 1);
    consume(t, 0);
    if (t + 2 < t1) {
      if (t + 4 < t1) fetch(t + 4, 0);
      consume(t + 2, 1);
    }
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (j < nq) {
      float* pp = a.part + ((((size_t)b * a.H + h) * a.NS + split) * a.beam + q0 + j) * (DH + 2);
#pragma unroll
      for (int e = 0; e < 8; e += 2)  // rows of DH + 2 floats are 8-byte, not 16-byte, aligned
        *reinterpret_cast<float2*>(pp + c0 + e) = make_float2(acc[j][e], acc[j][e + 1]);
      if (c0 == 0) {
        pp[DH] = m[j];
        pp[DH + 1] = l[j];
      }
    }
  }
}

// out[i, h*DH + c] = sum_s e^{m_s - M} o_s[c] / sum_s e^{m_s - M} l_s
// ---- cross-attention step on LDS-DMA tiles and the matrix cores (head_dim 64, d <= 640, beam <= 16, row-major K/V).
// A workgroup = one utterance x one run of frames, ALL heads (wave w = head w): the K|V rows of 16 frames (16 x 8d
// bytes, contiguous in HBM: one fully sequential stream per workgroup instead of 256-byte pieces 8d bytes apart) go
// straight into a double-buffered LDS image by global_load_lds_dwordx4 while the previous tile is consumed -- no
// staging registers, 16 x 8d bytes in flight per CU at all times.  Per tile and head, transposed scores
// S^T[frame][beam] = K Q^T on v_mfma_f32_16x16x4_f32 (beams padded to 16): in the result layout a lane owns one beam
// and four frames, so the online-softmax statistics need two cross-lane steps, the probabilities ARE the B operand of
// O^T[channel][beam] += V^T P^T (k slot g of MFMA i <-> frame 4g+i, V^T read in that order), and the running rescale
// is one scalar per lane.  LDS image: the 16-byte slot s of frame row R lands in slot s ^ R ^ ((R & 4) << 1) within
// its 256-byte head segment (swizzle on the global SOURCE address, the image of an LDS-DMA being lane-linear): the
// ds_read_b128 of the K operand and the ds_read_b32 of the V operand are both conflict-free (checked exhaustively,
// DESIGN.md).  Partial (context, max, sum) per run of frames -> cross_merge_kernel, as for the other variants.
// FR = frames per LDS tile: 16, or 8 (half of the MFMA rows idle, half the LDS: two workgroups per CU whose load
// and compute phases interleave).
template <int FR>
__global__ void __launch_bounds__(1024) cross_attn_dma_kernel(CrossAttnArgs a, int chunk) {
  SBK_DYN_LDS(float, lds);  // [2 stages][FR frames][2d]
  const int ROW = 2 * a.d, TILE = FR * ROW, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, h = sbk::uniform(tid >> 6);  // wave = head
  const int split = blockIdx.x, b = blockIdx.y;
  const int nq = a.beam, col = lane & 15, g = lane >> 4;
  const int klen = min(max(a.enc_len[b], 1), a.T);
  const int t0 = split * chunk, t1 = min(klen, t0 + chunk);
  float* pp = a.part ? a.part + ((((size_t)b * H + h) * a.NS + split) * nq + col) * (64 + 2) : nullptr;
  // The partial results of an utterance's runs are merged by whichever run finishes LAST (a.cnt; round 3): publish,
  // ticket, and the last ticket combines the NS partials in run order with cross_merge_kernel's arithmetic (bit-identical
  // to the two-launch path), then re-arms the ticket.  One launch less per layer and step; nobody waits.
  auto merge_if_last = [&]() SBK_INLINE_LAMBDA {
    if (!a.cnt || a.NS == 1) return;
    int* ticket = reinterpret_cast<int*>(lds);  // (every wave is past its last tile)
    sbk::vm_drain();
    __syncthreads();
    if (tid == 0) {
      sbk::release_agent();
      *ticket = sbk::atomic_add_agent(a.cnt + b, 1);
    }
    __syncthreads();
    const bool last = sbk::uniform(*ticket) == a.NS - 1;
    if (!last) return;
    if (tid == 0) {
      sbk::acquire_agent();
      sbk::atomic_store_agent(a.cnt + b, 0);
    }
    __syncthreads();
    const int nthr = 64 * H;
    for (int e = tid; e < nq * H * 64; e += nthr) {
      const int j = e / (H * 64), hh = (e / 64) % H, c = e % 64;
      const float* qq = a.part + ((((size_t)b * H + hh) * a.NS) * nq + j) * (64 + 2);
      const size_t stride = (size_t)nq * (64 + 2);
      float Mx = -INFINITY;
      for (int s = 0; s < a.NS; ++s) Mx = fmaxf(Mx, qq[s * stride + 64]);
      float num = 0.0f, den = 0.0f;
      for (int s = 0; s < a.NS; ++s) {
        const float l = qq[s * stride + 65];
        if (l > 0.0f) {
          const float w = expf(qq[s * stride + 64] - Mx);
          num = fmaf(w, qq[s * stride + c], num);
          den = fmaf(w, l, den);
        }
      }
      a.out[((size_t)b * nq + j) * a.d + hh * 64 + c] = num / den;
    }
  };
  if (t0 >= t1) {  // (uniform per workgroup) nothing of this run is inside the utterance: an empty partial
    if (a.NS > 1 && g == 0 && col < nq) {
      pp[64] = -INFINITY;
      pp[65] = 0.0f;
    }
    merge_if_last();
    return;
  }
  const float* kvb = a.kv + (size_t)b * a.T * ROW;
  float qf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    qf[i] = col < nq ? a.q[((size_t)b * nq + col) * a.d + h * 64 + 16 * g + i] * a.scale : 0.0f;

  // loader: the H waves share the 16 frame rows of a tile; one wave-instruction moves 1 KB = 4 head segments of a row
  const int per_row = ROW / 256;             // wave-instructions per frame row
  const int pieces = FR * per_row;           // per tile
  const int seg = lane >> 4, slot = lane & 15;
  auto issue = [&](int tile, int stage) SBK_INLINE_LAMBDA {
    for (int pc = h; pc < pieces; pc += H) {
      const int R = pc / per_row, qd = pc - R * per_row;
      const int frame = min(t0 + tile * FR + R, klen - 1);  // rows past the run re-read a valid row (masked below)
      const int fsw = R ^ ((R & 4) << 1);
      sbk::glds16(kvb + (size_t)frame * ROW + (qd * 4 + seg) * 64 + ((slot ^ fsw) & 15) * 4,
                  lds + stage * TILE + R * ROW + qd * 256);
    }
  };
  sbk::f32x4 o[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[ct][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const int ntiles = (t1 - t0 + FR - 1) / FR;
  const int krow = col & (FR - 1);  // (FR = 8: MFMA rows 8-15 repeat rows 0-7 and are masked out of the softmax)
  const int ksw = krow ^ ((krow & 4) << 1);  // swizzle of this lane's K row
  issue(0, 0);
  sbk::vm_drain();
  __syncthreads();
  int stage = 0;
  for (int k = 0; k < ntiles; ++k) {
    if (k + 1 < ntiles) issue(k + 1, stage ^ 1);
    const float* Kt = lds + stage * TILE + h * 64;
    const float* Vt = Kt + a.d;
    float4 kq[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) kq[jq] = *reinterpret_cast<const float4*>(Kt + krow * ROW + (((4 * g + jq) ^ ksw) & 15) * 4);
    sbk::f32x4 sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[r] = 0.0f;
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      sc = sbk::mfma_16x16x4(kq[jq].x, qf[4 * jq], sc);
      sc = sbk::mfma_16x16x4(kq[jq].y, qf[4 * jq + 1], sc);
      sc = sbk::mfma_16x16x4(kq[jq].z, qf[4 * jq + 2], sc);
      sc = sbk::mfma_16x16x4(kq[jq].w, qf[4 * jq + 3], sc);
    }
    // sc[r] = score of (frame t0 + FR k + 4g + r, beam col)
    const int fb = t0 + k * FR + 4 * g;
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (fb + r >= t1 || 4 * g + r >= FR) sc[r] = -INFINITY;
      mt = fmaxf(mt, sc[r]);
    }
    mt = fmaxf(mt, sbk::shfl_xor(mt, 16));
    mt = fmaxf(mt, sbk::shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);  // finite: frame t0 + 16k is inside the run
    const float alpha = expf(m_run - m_new);
    float p[4], ps = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = expf(sc[r] - m_new);
      ps += p[r];
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[ct][r] *= alpha;
    // O^T[channel 16ct + 4g' + r][beam] += V^T P^T: k slot g of MFMA i <-> frame row 4g + i
    const int vsw = (g & 1) << 3;  // (R & 4) << 1 of R = 4g + i (mod FR); the R part of the swizzle is XORed below
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int R = (4 * g + i) & (FR - 1);
        const float va = Vt[R * ROW + ((((4 * ct + (col >> 2)) ^ R ^ vsw) & 15) << 2) + (col & 3)];
        o[ct] = sbk::mfma_16x16x4(va, p[i], o[ct]);
      }
    sbk::vm_drain();   // this wave's pieces of the next tile have landed ...
    __syncthreads();   // ... and everybody's; every wave is done with `stage`
    stage ^= 1;
  }
  float l_tot = l_run + sbk::shfl_xor(l_run, 16);
  l_tot += sbk::shfl_xor(l_tot, 32);
  if (col < nq) {
    if (a.NS == 1) {
      float* op = a.out + ((size_t)b * nq + col) * a.d + h * 64;
      const float inv = 1.0f / l_tot;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[16 * ct + 4 * g + r] = o[ct][r] * inv;
    } else {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[16 * ct + 4 * g + r] = o[ct][r];
      if (g == 0) {
        pp[64] = m_run;
        pp[65] = l_tot;
      }
    }
  }
  merge_if_last();
}

__global__ void __launch_bounds__(256) cross_merge_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int H, int NS, int beam, int DH, int d) {
  const int i = blockIdx.x;  // hypothesis
  const int b = i / beam, j = i % beam;
  for (int e = threadIdx.x; e < H * DH; e += 256) {
    const int h = e / DH, c = e % DH;
    const float* pp = part + ((((size_t)b * H + h) * NS) * beam + j) * (DH + 2);
    const size_t stride = (size_t)beam * (DH + 2);
    float M = -INFINITY;
    for (int s = 0; s < NS; ++s) M = fmaxf(M, pp[s * stride + DH]);
    float num = 0.0f, den = 0.0f;
    for (int s = 0; s < NS; ++s) {
      const float l = pp[s * stride + DH + 1];
      if (l > 0.0f) {
        const float w = expf(pp[s * stride + DH] - M);
        num = fmaf(w, pp[s * stride + c], num);
        den = fmaf(w, l, den);
      }
    }
    out[(size_t)i * d + h * DH + c] = num / den;
  }
}

template <int DH>
void launch_frames(const CrossAttnArgs& a, int qtiles, hipStream_t st) {
  if (a.fc == 64) {
    SBK_LAUNCH((cross_attn_step_kernel<DH, 64>), dim3(a.NS, a.H, a.B * qtiles), dim3(256), 0, st, a);
  } else if (a.fc == 256) {
    SBK_LAUNCH((cross_attn_step_kernel<DH, 256>), dim3(a.NS, a.H, a.B * qtiles), dim3(256), 0, st, a);
  } else {
    SBK_LAUNCH((cross_attn_step_kernel<DH, kFC>), dim3(a.NS, a.H, a.B * qtiles), dim3(256), 0, st, a);
  }
}

template <int DH>
int launch_cross(const CrossAttnArgs& a, hipStream_t st) {
  const int qtiles = (a.beam + kQT - 1) / kQT;
  sbk::ProfScope prof("cross_attn_step", 4.0 * a.B * a.beam * (double)a.T * a.d, 8.0 * a.B * (double)a.T * a.d, st);
  if constexpr (DH == 64) {
    // default: LDS-DMA tiles + matrix cores (all heads of an utterance per workgroup, sequential HBM stream)
    // Measured (tools/decode_probe.py, profiles/r03_cross_attention_dma_sweep.log): with four recipe-sized batches per
    // search (128 utterances x 430 frames) one workgroup per CU streaming ~14 tiles runs the step in 64 us against 92
    // for the frame-per-thread kernel (3.5 vs 2.45 TB/s incl. the merge); more, shorter workgroups or 8-frame tiles
    // lose (73 / 92 / 78-95 us), and for a single 32-utterance batch the frame-per-thread kernel wins (38 vs 52 us):
    // the default (7) takes the LDS-DMA kernel from ~40 K memory frames per search on.
    const bool dma_auto = sbk::g_cross_rows == 7 && (long)a.B * a.T >= 40000;
    if ((sbk::g_cross_rows == 5 || sbk::g_cross_rows == 6 || dma_auto) && !a.head_major && a.d <= 640 && a.H * 64 == a.d &&
        a.beam <= 16 && sbk::aligned16(a.kv) && (a.part || a.T <= 8)) {
      CrossAttnArgs c = a;
      if (!sbk::g_cross_fused_merge) c.cnt = nullptr;
      const int FR = sbk::g_cross_rows == 6 ? 8 : 16;
      // ONE run per utterance from ~100 utterances per search on (round 5 default; knob 8 = 3 forces it, 4 = round 4's 256
      // workgroups): no partials, no merge launch, B workgroups walk their whole memory.  On one stream it is slower (2.20 vs
      // 2.03 ms per step: half of the CUs idle), under the eight workers it is +1.8 % on the headline in three paired runs
      // (11 925 / 11 870 / 11 901 against 11 725 / 11 648 / 11 681, profiles/r05_b_*): the co-resident streams use the other CUs
      // and six merge launches per step are gone.
      const bool one_run = sbk::g_cross_fc256 == 3 || (dma_auto && a.B >= 96 && sbk::g_cross_fc256 != 4);
      const int target = one_run ? a.B : dma_auto ? 256 : (sbk::g_cross_fc256 == 1 ? 256 : (sbk::g_cross_fc256 == 2 ? 1024 : 512));
      int ns = sbk::cdiv(target, a.B);                          // workgroups over the batch ...
      if (ns > sbk::cdiv(a.T, FR)) ns = sbk::cdiv(a.T, FR);    // ... of at least one tile
      if (ns > sbk::cdiv(a.T, 16)) ns = sbk::cdiv(a.T, 16);    // (the partial buffer is sized for 16-frame runs)
      if (ns < 1) ns = 1;
      const int chunk = sbk::cdiv(sbk::cdiv(a.T, ns), 16) * 16;
      c.NS = sbk::cdiv(a.T, chunk);
      const size_t lds = (size_t)2 * FR * 2 * a.d * sizeof(float);
      static bool once = false;
      if (!once) {
        (void)SBK_ALLOW_DYN_LDS(cross_attn_dma_kernel<16>, 160 * 1024);
        (void)SBK_ALLOW_DYN_LDS(cross_attn_dma_kernel<8>, 160 * 1024);
        once = true;
      }
      if (FR == 16) {
        SBK_LAUNCH(cross_attn_dma_kernel<16>, dim3(c.NS, a.B), dim3(64 * a.H), lds, st, c, chunk);
      } else {
        SBK_LAUNCH(cross_attn_dma_kernel<8>, dim3(c.NS, a.B), dim3(64 * a.H), lds, st, c, chunk);
      }
      int rc5 = sbk::launch_status("cross_attn_step");
      if (rc5 || c.NS == 1 || c.cnt) return rc5;  // (c.cnt: merged by the last-arriving run of every utterance)
      SBK_LAUNCH(cross_merge_kernel, dim3(a.B * a.beam), dim3(256), 0, st, (const float*)c.part, c.out, c.H, c.NS,
                 c.beam, DH, c.d);
      return sbk::launch_status("cross_merge");
    }
    // streaming variant (one wave per run of frames, no LDS): d = 512 with 16-byte aligned rows, row-major K/V
    if ((sbk::g_cross_rows == 3 || sbk::g_cross_rows == 4) && a.d == 512 && !a.head_major && a.part &&
        sbk::aligned16(a.kv) && sbk::aligned16(a.q) && (reinterpret_cast<uintptr_t>(a.part) & 7) == 0) {
      // knob 4 = 3: all (<= 10) beams in one wave (288 registers, one wave per SIMD); 4: five beams per wave, two
      // waves share a run of frames through L1/L2 (half the registers, three waves per SIMD)
      const int NQ = sbk::g_cross_rows == 3 ? 10 : 5;
      const int qt = (a.beam + NQ - 1) / NQ;
      // frames per wave: >= ~4 waves per CU where the batch allows it, 16 .. 64 frames, multiple of 4
      long total = (long)a.B * qt * a.T;
      int fpw = (int)(total / 1024);
      fpw = fpw < 16 ? 16 : (fpw > 64 ? 64 : fpw);
      fpw = (fpw + 3) & ~3;
      CrossAttnArgs b = a;
      b.NS = sbk::cdiv(a.T, fpw);
      const int waves = a.B * qt * b.NS;
      if (NQ == 10) {
        SBK_LAUNCH((cross_attn_stream_kernel<10>), dim3(sbk::cdiv(waves, 4)), dim3(256), 0, st, b, fpw);
      } else {
        SBK_LAUNCH((cross_attn_stream_kernel<5>), dim3(sbk::cdiv(waves, 4)), dim3(256), 0, st, b, fpw);
      }
      int rc3 = sbk::launch_status("cross_attn_step");
      if (rc3) return rc3;
      SBK_LAUNCH(cross_merge_kernel, dim3(a.B * a.beam), dim3(256), 0, st, (const float*)b.part, b.out, b.H, b.NS,
                 b.beam, DH, b.d);
      return sbk::launch_status("cross_merge");
    }
  }
  if constexpr (DH == 64 || DH == 32) {
    if (sbk::g_cross_rows == 2 && (a.d % 4) == 0 && sbk::aligned16(a.kv) && sbk::aligned16(a.q)) {
      const int qt2 = (a.beam + kQT2 - 1) / kQT2;
      SBK_LAUNCH((cross_attn_mfma_kernel<DH>), dim3(a.NS, a.H, a.B * qt2), dim3(256), 0, st, a);
      int rc2 = sbk::launch_status("cross_attn_step");
      if (rc2 || a.NS == 1) return rc2;
      SBK_LAUNCH(cross_merge_kernel, dim3(a.B * a.beam), dim3(256), 0, st, (const float*)a.part, a.out, a.H, a.NS,
                 a.beam, DH, a.d);
      return sbk::launch_status("cross_merge");
    }
  }
  if constexpr (DH == 64 || DH == 32 || DH == 16) {
    if (sbk::g_cross_rows == 1 && (a.d % 4) == 0 && sbk::aligned16(a.kv) && sbk::aligned16(a.q)) {
      SBK_LAUNCH((cross_attn_rows_kernel<DH>), dim3(a.NS, a.H, a.B * qtiles), dim3(256), 0, st, a);
    } else {
      launch_frames<DH>(a, qtiles, st);
    }
  } else {
    launch_frames<DH>(a, qtiles, st);
  }
  int rc = sbk::launch_status("cross_attn_step");
  if (rc || a.NS == 1) return rc;
  SBK_LAUNCH(cross_merge_kernel, dim3(a.B * a.beam), dim3(256), 0, st, (const float*)a.part, a.out, a.H, a.NS, a.beam,
             DH, a.d);
  return sbk::launch_status("cross_merge");
}

// ---- head-averaged cross-attention probabilities of ONE decoder layer for the current position: what
// nn.MultiheadAttention hands back as attention weights (average over heads) and the CTC scorer's attention window
// reads (decoders/ctc.py:189-200 through scorer.py:183-187).  One workgroup per hypothesis; only run when
// ctc_window_size > 0.  out [n,T]: frames past the utterance end get exactly 0 (the key padding mask).
__global__ void __launch_bounds__(256) cross_attn_avg_probs_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                   const int32_t* __restrict__ enc_len,
                                                                   float* __restrict__ out, int T, int d, int H, int beam,
                                                                   float scale) {
  SBK_DYN_LDS(float, lds);  // acc[T] | sc[T] | red[8]
  float* acc = lds;
  float* sc = lds + T;
  float* red = sc + T;
  const int n = blockIdx.x, b = n / beam, tid = threadIdx.x, Dh = d / H;
  const int klen = min(max(enc_len[b], 1), T);
  const float* kvb = kv + (size_t)b * T * 2 * d;
  for (int t = tid; t < T; t += 256) acc[t] = 0.0f;
  for (int h = 0; h < H; ++h) {
    const float* qh = q + (size_t)n * d + h * Dh;
    float m = -INFINITY;
    for (int t = tid; t < klen; t += 256) {
      const float* kr = kvb + (size_t)t * 2 * d + h * Dh;
      float s = 0.0f;
      for (int c = 0; c < Dh; ++c) s = fmaf(qh[c] * scale, kr[c], s);
      sc[t] = s;
      m = fmaxf(m, s);
    }
    m = sbk::wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int t = tid; t < klen; t += 256) {
      const float e = expf(sc[t] - m);
      sc[t] = e;
      sum += e;
    }
    sum = sbk::wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    for (int t = tid; t < klen; t += 256) acc[t] += sc[t] / sum;
    __syncthreads();
  }
  for (int t = tid; t < T; t += 256) out[(size_t)n * T + t] = t < klen ? acc[t] / (float)H : 0.0f;
}

// ---------------------------------------------------------------- log-softmax over the vocabulary
// out[i,c] = w * (x[i,c]/temp - logsumexp(x[i,:]/temp));  one workgroup per row.
// bias / bias2 (optional, [V]): additive masks on the logits (0 or -inf: suppressed tokens come out as -inf)
__global__ void __launch_bounds__(256) log_softmax_row_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              int V, float inv_temp, float w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ bias2) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const float* xr = x + (size_t)blockIdx.x * V;
  float* orow = out + (size_t)blockIdx.x * V;
  float m = -INFINITY;
  auto at = [&](int c) SBK_INLINE_LAMBDA { return sbk::ls_logit(xr[c], bias ? bias[c] : 0.0f, bias2 ? bias2[c] : 0.0f, inv_temp); };
  for (int c = tid; c < V; c += 256) m = fmaxf(m, at(c));
  m = sbk::wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
  for (int c = tid; c < V; c += 256) s = sbk::add_rn(s, expf(sbk::sub_rn(at(c), m)));
  s = sbk::wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float lse = sbk::add_rn(m, logf((red[0] + red[1]) + (red[2] + red[3])));
  for (int c = tid; c < V; c += 256) orow[c] = sbk::ls_out(at(c), lse, w);
}

}  // namespace

namespace sbk {
thread_local const int32_t* g_step_ptr = nullptr;
thread_local int g_step_min_steps = 0;

// Measured alternatives kept behind sbk_prof_set_knob (Conformer-L, B=64, MI355X; cross_attn_step total per
// 8 batches): frame-per-thread kernel 247 ms with either layout; row-coalesced kernel 336 ms on [B,T,2d],
// 306 ms on head-major [B,H,T,2*Dh]; at B=128 the MFMA formulation takes 315 ms vs 303 ms.  The defaults stay 0.
int g_cross_rows = 7;     // key 4: 7 (default) = LDS-DMA / MFMA kernel for large grouped searches, else frame-per-thread;
                          // 0 frame-per-thread kernel, 1 row-coalesced kernel, 2 MFMA kernel (head_dim 64 / 32), 5 / 6 = the
                          // LDS-DMA kernel always (16- / 8-frame tiles)
int g_kv_head_major = 0;  // key 5: cross K/V stored [B,H,T,2*Dh] instead of [B,T,2d]

int embed_pos(const int32_t* tok, const float* emb, const float* pe_row, float* x, int n, int d, float scale,
              hipStream_t st) {
  if (n == 0) return 0;
  ProfScope prof("embed_pos", 2.0 * n * d, 8.0 * n * d, st);
  const int32_t* sp = g_step_ptr;  // a local: launch arguments must not name the thread_local itself
  SBK_LAUNCH(embed_pos_kernel, dim3(n), dim3(256), 0, st, tok, emb, pe_row, x, n, d, scale, sp);
  return launch_status("embed_pos");
}

int self_attn_step(const float* qkv, float* kcache, float* vcache, const int32_t* kv_slot, float* out, int n, int d,
                   int H, int step, int nslot, int Lmax, hipStream_t st, const int32_t* key_tok, int key_stride,
                   int key_shift, int key_first, int pad_idx, int group) {
  if (n == 0) return 0;
  if (group < 1 || n % group != 0 || g_self_group_off) group = 1;
  SelfAttnArgs a{qkv, kcache, vcache, kv_slot, out, n, d, H, d / H, step, nslot, Lmax, 1.0f / sqrtf((float)(d / H)),
                 g_step_ptr, key_tok, key_stride, key_shift, key_first, pad_idx, group};
  const size_t lds = (size_t)8 * (((Lmax + 63) / 64) * 64) * sizeof(float);
  if (lds > 64 * 1024) return fail(SBK_EINVAL, "self_attn_step: Lmax=%d too long for the LDS window", Lmax);
  ProfScope prof("self_attn_step", 4.0 * n * d * (step + 1), 8.0 * n * d * (step + 1), st);
  SBK_LAUNCH(self_attn_step_kernel, dim3(cdiv(n * H, 4)), dim3(256), lds, st, a);
  return launch_status("self_attn_step");
}

// Number of memory splits used for T frames and floats of partial storage they need.
int g_self_group_off = 0;  // tuning knob (key 13): 1 = self-attention waves of a workgroup take 4 heads of one hypothesis
                           // (the round-1 mapping) instead of 4 beams of one (utterance, head)
int g_cross_fc256 = 0;  // tuning knob (key 8): memory frames per workgroup of the frame-per-thread kernel: 0 = 128, 1 = 256, 2 = 64
int cross_attn_splits(int T) { return cdiv(T, 16); }  // sizes the partial buffer for the finest split (16-frame runs)
size_t cross_attn_partial_floats(int B, int T, int H, int Dh, int beam) {
  const int ns = cross_attn_splits(T);
  return ns > 1 ? (size_t)B * H * ns * beam * (Dh + 2) : 0;
}

// [B,T,2d] (projection output) -> [B,H,T,2*Dh] (what cross_attn_step streams when head_major = 1)
int kv_head_major(const float* src, float* dst, int B, int T, int d, int H, hipStream_t st) {
  const int Dh = d / H;
  if (B == 0 || T == 0) return 0;
  if (Dh % 4 != 0 || !aligned16(src) || !aligned16(dst)) return fail(SBK_EINVAL, "kv_head_major: head_dim %d / alignment", Dh);
  const long total4 = (long)B * T * 2 * d / 4;
  ProfScope prof("kv_head_major", 0.0, 8.0 * B * (double)T * 2 * d, st);
  const long blocks = (total4 + 255) / 256;
  SBK_LAUNCH(kv_head_major_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
             reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), T, H, Dh / 4, total4);
  return launch_status("kv_head_major");
}

// tuning knob (key 37): 1 = the LDS-DMA cross-attention merges in its last-arriving workgroup.  OFF (see knob 36: the
// per-workgroup agent-scope release costs more than the cross_merge launch it saves)
int g_cross_fused_merge = 0;
int cross_attn_step(const float* q, const float* kv, const int32_t* enc_len, float* out, float* part, int B, int T,
                    int d, int H, int beam, hipStream_t st, int head_major, int32_t* cnt) {
  if (B == 0) return 0;
  const int Dh = d / H;
  const int fc = g_cross_rows != 0 ? kFC : (g_cross_fc256 == 1 ? 256 : (g_cross_fc256 == 2 ? 64 : kFC));
  const int NS = cdiv(T, fc);
  if (NS > 1 && !part) return fail(SBK_EINVAL, "cross_attn_step: T=%d needs a partial buffer", T);
  CrossAttnArgs a{q, kv, enc_len, out, part, B, T, d, H, Dh, beam, NS, 1.0f / sqrtf((float)Dh), head_major, fc, cnt};
  switch (Dh) {
    case 64: return launch_cross<64>(a, st);
    case 36: return launch_cross<36>(a, st);
    case 32: return launch_cross<32>(a, st);
    case 16: return launch_cross<16>(a, st);
    case 8: return launch_cross<8>(a, st);
    default: return fail(SBK_EINVAL, "cross_attn_step: head_dim %d not instantiated (8,16,32,36,64)", Dh);
  }
}

int cross_attn_avg_probs(const float* q, const float* kv, const int32_t* enc_len, float* out, int n, int T, int d, int H,
                         int beam, hipStream_t st) {
  if (n == 0) return 0;
  const size_t lds = ((size_t)2 * T + 8) * sizeof(float);
  if (lds > 64 * 1024) return fail(SBK_EINVAL, "cross_attn_avg_probs: T=%d too long for the LDS window", T);
  ProfScope prof("cross_attn_probs", 2.0 * n * (double)T * d, 4.0 * (n / beam) * (double)T * d, st);
  SBK_LAUNCH(cross_attn_avg_probs_kernel, dim3(n), dim3(256), lds, st, q, kv, enc_len, out, T, d, H, beam,
             1.0f / sqrtf((float)(d / H)));
  return launch_status("cross_attn_avg_probs");
}

int log_softmax_rows(const float* x, float* out, int rows, int V, float temperature, float weight, hipStream_t st,
                     const float* bias, const float* bias2) {
  if (rows == 0) return 0;
  ProfScope prof("log_softmax", 4.0 * rows * V, 8.0 * rows * V, st);
  SBK_LAUNCH(log_softmax_row_kernel, dim3(rows), dim3(256), 0, st, x, out, V, 1.0f / temperature, weight, bias, bias2);
  return launch_status("log_softmax_rows");
}

}  // namespace sbk

extern "C" int sbk_log_softmax_f32(const float* x, float* out, int rows, int V, float temperature, float weight,
                                   sbk_stream_t stream) {
  if (rows == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(x && out && rows >= 0 && V > 0 && temperature > 0.0f, "log_softmax: bad arguments");
  return sbk::log_softmax_rows(x, out, rows, V, temperature, weight, sbk::as_stream(stream), nullptr, nullptr);
}
