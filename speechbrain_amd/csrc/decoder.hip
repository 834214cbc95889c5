// Kernels of the KV-cached Transformer decoder step (TransformerASR.decode replaced).
//
// The reference recomputes the whole prefix every step (TransformerASR.py:426-473,
// seq2seq.py:1929-1934).  Here one step touches one new token per hypothesis:
//   embed_pos        x = emb[tok]*sqrt(d) + pe[step]                        (HBM-bound gather)
//   self_attn_step   causal MHA of the new token over its own prefix; K/V of
//                    earlier positions live in a slot-addressed cache
//                    [pos][slot][d] and are reached through the per-hypothesis
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
                                                        int n, int d, float scale) {
  const int i = blockIdx.x;
  const float* e = emb + (size_t)tok[i] * d;
  for (int c = threadIdx.x; c < d; c += 256) x[(size_t)i * d + c] = e[c] * scale + pe_row[c];
}

// ---------------------------------------------------------------- self attention, one new token
struct SelfAttnArgs {
  const float* qkv;        // [n,3d] stacked (q | k | v) for the new token
  float* kcache;           // [Lmax][nslot][d]
  float* vcache;
  const int32_t* kv_slot;  // [n][Lmax]: slot holding position p of hypothesis i (p < step)
  float* out;              // [n,d]
  int n, d, H, Dh, step, nslot, Lmax;
  float scale;
};

// One wave per (hypothesis, head).  The 64 lanes form 4 position groups x 16 lanes; a group's 16
// lanes read one K (or V) head row as 16-byte pieces (a 256-byte row per load instruction and
// group, fully coalesced) and several rows are in flight at once, so the walk over the prefix is
// a handful of dependent round trips instead of one per position.
__global__ void __launch_bounds__(256) self_attn_step_kernel(SelfAttnArgs a) {
  SBK_DYN_LDS(float, lds);  // [4 waves][Lmax_pad] scores -> probabilities
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  const bool live = item < a.n * a.H;
  const int i = live ? item / a.H : 0, h = live ? item % a.H : 0;
  const int d = a.d, Dh = a.Dh, L = a.step + 1;
  const int lpad = ((a.Lmax + 63) / 64) * 64;
  float* prob = lds + wave * lpad;
  const float* q = a.qkv + (size_t)i * 3 * d + h * Dh;
  const float* knew = q + d;
  const float* vnew = q + 2 * d;
  const int pg = lane >> 4, cq = lane & 15;  // position group, 16-byte piece of the head row
  const int nchunk = (Dh + 63) / 64;         // 64-float chunks per head row (1 for Dh <= 64)
  // append this token's K/V head slice to the cache (slot = hypothesis index)
  if (live) {
    for (int c = lane; c < Dh; c += 64) {
      const size_t o = ((size_t)a.step * a.nslot + i) * d + h * Dh + c;
      a.kcache[o] = knew[c];
      a.vcache[o] = vnew[c];
    }
  }
  // scores
  for (int p0 = 0; p0 < L; p0 += 4) {
    const int p = p0 + pg;
    float s = 0.0f;
    if (p < L) {
      const float* kp = (p == a.step) ? knew : a.kcache + ((size_t)p * a.nslot + a.kv_slot[(size_t)i * a.Lmax + p]) * d + h * Dh;
      for (int ch = 0; ch < nchunk; ++ch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = ch * 64 + cq * 4 + e;
          if (c < Dh) s = fmaf(q[c] * a.scale, kp[c], s);
        }
      }
    }
    s += sbk::shfl_xor(s, 1);
    s += sbk::shfl_xor(s, 2);
    s += sbk::shfl_xor(s, 4);
    s += sbk::shfl_xor(s, 8);
    if (p < L && cq == 0) prob[p] = s;
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
  for (int ch = 0; ch < nchunk; ++ch) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int p0 = 0; p0 < L; p0 += 4) {
      const int p = p0 + pg;
      if (p < L) {
        const float* vp = (p == a.step) ? vnew : a.vcache + ((size_t)p * a.nslot + a.kv_slot[(size_t)i * a.Lmax + p]) * d + h * Dh;
        const float w = prob[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = ch * 64 + cq * 4 + e;
          if (c < Dh) acc[e] = fmaf(w, vp[c], acc[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] += sbk::shfl_xor(acc[e], 16);
      acc[e] += sbk::shfl_xor(acc[e], 32);
      const int c = ch * 64 + cq * 4 + e;
      if (live && pg == 0 && c < Dh) a.out[(size_t)i * d + h * Dh + c] = acc[e];
    }
  }
}

// ---------------------------------------------------------------- cross attention, all beams of an utterance
constexpr int kQT = 16;  // queries (beams) served per workgroup

struct CrossAttnArgs {
  const float* q;         // [n,d]   n = B*beam, hypothesis i belongs to utterance i / beam
  const float* kv;        // [B,T,2d] per frame: K (d) then V (d), projected once per utterance
  const int32_t* enc_len; // [B]
  float* out;             // [n,d]
  int B, T, d, H, Dh, beam, SP;
  float scale;
};

template <int DH>
__global__ void __launch_bounds__(256) cross_attn_step_kernel(CrossAttnArgs a) {
  SBK_DYN_LDS(float, lds);
  float* qs = lds;                    // [kQT][DH]   scaled queries
  float* S = qs + kQT * DH;           // [kQT][SP]   scores -> probabilities
  float* red = S + kQT * a.SP;        // [4][kQT][DH] partial contexts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * kQT, h = blockIdx.y, b = blockIdx.z;
  const int nq = min(kQT, a.beam - q0);
  const int T = a.T, d = a.d, SP = a.SP;
  const int klen = min(max(a.enc_len[b], 1), T);
  const float* kvb = a.kv + (size_t)b * T * 2 * d + h * DH;

  for (int idx = tid; idx < kQT * DH; idx += 256) {
    const int j = idx / DH, c = idx % DH;
    qs[idx] = j < nq ? a.q[((size_t)b * a.beam + q0 + j) * d + h * DH + c] * a.scale : 0.0f;
  }
  __syncthreads();
  // scores: thread <-> memory frame, K row held in registers and reused by every beam
  for (int t = tid; t < klen; t += 256) {
    float kr[DH];
    const float* kp = kvb + (size_t)t * 2 * d;
#pragma unroll
    for (int c = 0; c < DH; ++c) kr[c] = kp[c];
    for (int j = 0; j < nq; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < DH; ++c) s = fmaf(qs[j * DH + c], kr[c], s);
      S[j * SP + t] = s;
    }
  }
  __syncthreads();
  // softmax over valid frames, one wave per query row
  for (int j = wave; j < nq; j += 4) {
    float* Sr = S + j * SP;
    float m = -INFINITY;
    for (int t = lane; t < klen; t += 64) m = fmaxf(m, Sr[t]);
    m = sbk::wave_max(m);
    float sum = 0.0f;
    for (int t = lane; t < klen; t += 64) {
      const float e = expf(Sr[t] - m);
      Sr[t] = e;
      sum += e;
    }
    sum = sbk::wave_sum(sum);
    for (int t = lane; t < klen; t += 64) Sr[t] = Sr[t] / sum;
  }
  __syncthreads();
  // context: wave <-> quarter of the frames, lane <-> channel; V row read once for all beams.
  // Rows are fetched 8 at a time so that 8 loads are in flight per lane.
  {
    float acc[kQT];
#pragma unroll
    for (int j = 0; j < kQT; ++j) acc[j] = 0.0f;
    const int c = lane;
    if (c < DH) {
      const float* vcol = kvb + d + c;
      for (int t0 = wave; t0 < klen; t0 += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = t0 + 4 * u;
          v[u] = t < klen ? vcol[(size_t)t * 2 * d] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = t0 + 4 * u;
          if (t < klen) {
#pragma unroll
            for (int j = 0; j < kQT; ++j) acc[j] = fmaf(S[j * SP + t], v[u], acc[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kQT; ++j) red[(wave * kQT + j) * DH + c] = acc[j];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < nq * DH; idx += 256) {
    const int j = idx / DH, c = idx % DH;
    const float v = (red[(0 * kQT + j) * DH + c] + red[(1 * kQT + j) * DH + c]) +
                    (red[(2 * kQT + j) * DH + c] + red[(3 * kQT + j) * DH + c]);
    a.out[((size_t)b * a.beam + q0 + j) * d + h * DH + c] = v;
  }
}

template <int DH>
int launch_cross(const CrossAttnArgs& a, hipStream_t st) {
  const size_t lds = ((size_t)kQT * DH + (size_t)kQT * a.SP + (size_t)4 * kQT * DH) * sizeof(float);
  if (lds > 160 * 1024) return sbk::fail(SBK_EINVAL, "cross_attn: T=%d needs %zu B of LDS", a.T, lds);
  if (lds > 64 * 1024) {
    hipError_t e = SBK_ALLOW_DYN_LDS((cross_attn_step_kernel<DH>), lds);
    if (e != hipSuccess) return sbk::fail((int)e, "cross_attn: cannot raise the LDS window");
  }
  sbk::ProfScope prof("cross_attn_step", 4.0 * a.B * a.beam * (double)a.T * a.d, 8.0 * a.B * (double)a.T * a.d, st);
  SBK_LAUNCH((cross_attn_step_kernel<DH>), dim3(sbk::cdiv(a.beam, kQT), a.H, a.B), dim3(256), lds, st, a);
  return sbk::launch_status("cross_attn_step");
}

// ---------------------------------------------------------------- log-softmax over the vocabulary
// out[i,c] = w * (x[i,c]/temp - logsumexp(x[i,:]/temp));  one workgroup per row.
__global__ void __launch_bounds__(256) log_softmax_row_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              int V, float inv_temp, float w) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const float* xr = x + (size_t)blockIdx.x * V;
  float* orow = out + (size_t)blockIdx.x * V;
  float m = -INFINITY;
  for (int c = tid; c < V; c += 256) m = fmaxf(m, xr[c] * inv_temp);
  m = sbk::wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
  for (int c = tid; c < V; c += 256) s += expf(xr[c] * inv_temp - m);
  s = sbk::wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float lse = m + logf((red[0] + red[1]) + (red[2] + red[3]));
  for (int c = tid; c < V; c += 256) orow[c] = w * (xr[c] * inv_temp - lse);
}

}  // namespace

namespace sbk {

int embed_pos(const int32_t* tok, const float* emb, const float* pe_row, float* x, int n, int d, float scale,
              hipStream_t st) {
  if (n == 0) return 0;
  ProfScope prof("embed_pos", 2.0 * n * d, 8.0 * n * d, st);
  SBK_LAUNCH(embed_pos_kernel, dim3(n), dim3(256), 0, st, tok, emb, pe_row, x, n, d, scale);
  return launch_status("embed_pos");
}

int self_attn_step(const float* qkv, float* kcache, float* vcache, const int32_t* kv_slot, float* out, int n, int d,
                   int H, int step, int nslot, int Lmax, hipStream_t st) {
  if (n == 0) return 0;
  SelfAttnArgs a{qkv, kcache, vcache, kv_slot, out, n, d, H, d / H, step, nslot, Lmax, 1.0f / sqrtf((float)(d / H))};
  const size_t lds = (size_t)4 * (((Lmax + 63) / 64) * 64) * sizeof(float);
  if (lds > 64 * 1024) return fail(SBK_EINVAL, "self_attn_step: Lmax=%d too long for the LDS window", Lmax);
  ProfScope prof("self_attn_step", 4.0 * n * d * (step + 1), 8.0 * n * d * (step + 1), st);
  SBK_LAUNCH(self_attn_step_kernel, dim3(cdiv(n * H, 4)), dim3(256), lds, st, a);
  return launch_status("self_attn_step");
}

int cross_attn_step(const float* q, const float* kv, const int32_t* enc_len, float* out, int B, int T, int d, int H,
                    int beam, hipStream_t st) {
  if (B == 0) return 0;
  const int Dh = d / H;
  CrossAttnArgs a{q, kv, enc_len, out, B, T, d, H, Dh, beam, T + 1, 1.0f / sqrtf((float)Dh)};
  switch (Dh) {
    case 64: return launch_cross<64>(a, st);
    case 36: return launch_cross<36>(a, st);
    case 32: return launch_cross<32>(a, st);
    case 16: return launch_cross<16>(a, st);
    case 8: return launch_cross<8>(a, st);
    default: return fail(SBK_EINVAL, "cross_attn_step: head_dim %d not instantiated (8,16,32,36,64)", Dh);
  }
}

int log_softmax_rows(const float* x, float* out, int rows, int V, float temperature, float weight, hipStream_t st) {
  if (rows == 0) return 0;
  ProfScope prof("log_softmax", 4.0 * rows * V, 8.0 * rows * V, st);
  SBK_LAUNCH(log_softmax_row_kernel, dim3(rows), dim3(256), 0, st, x, out, V, 1.0f / temperature, weight);
  return launch_status("log_softmax_rows");
}

}  // namespace sbk

extern "C" int sbk_log_softmax_f32(const float* x, float* out, int rows, int V, float temperature, float weight,
                                   sbk_stream_t stream) {
  SBK_REQUIRE(x && out && rows >= 0 && V > 0 && temperature > 0.0f, "log_softmax: bad arguments");
  return sbk::log_softmax_rows(x, out, rows, V, temperature, weight, sbk::as_stream(stream));
}
