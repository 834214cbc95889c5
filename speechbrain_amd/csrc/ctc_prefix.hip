// CTC prefix scorer (Watanabe et al. 2017, Alg. 2), full-vocabulary mode.
// Replaces decoders/ctc.py:26-295 (CTCPrefixScore) as driven by scorer.py:108-255.
//
// The reference materialises r[T,2,n_bh,V] every step (10 MB per hypothesis) and walks
// the frames in a Python loop.  Here the state kept per hypothesis is only its own
// forward variables r[T][2] (plus phi inputs derived from them); one step is
//   ctc_score_step : thread <-> vocabulary entry c of one utterance, ALL beams of
//                    the utterance in registers, frames walked sequentially, so the
//                    emission row x[b,t,:] is read ONCE per step and utterance
//                    (coalesced along V) -- algorithmic traffic T*V*4 B per utterance
//                    per step; everything else is transcendental math (about 5
//                    exp/log per (hypothesis, token, frame)): VALU-bound, not HBM.
//   ctc_advance    : after the beam top-k picked (parent, token) for each new
//                    hypothesis, re-run the same recurrence for that single pair to
//                    obtain the new r[T][2] (the reference gathers it out of the
//                    materialised tensor, ctc.py:243-295).
// Numerics follow the reference: finite -1e20 "minus infinity" (ctc.py:53), two-term
// logsumexp as max + log(exp(a-max)+exp(b-max)).
#include "common.h"
#include "internal.h"

namespace {

constexpr float kNeg = -1e20f;
constexpr int kBT = 16;  // beams held in registers per thread

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

// x[b,t,c]: frames >= enc_len are -1e20 except column 0 which is 0 (ctc.py:57-61; the
// reference hard-codes column 0, which is the blank in every recipe).
__global__ void __launch_bounds__(256) ctc_mask_kernel(float* __restrict__ x, const int32_t* __restrict__ enc_len,
                                                       int T, int V) {
  const int b = blockIdx.y, t = blockIdx.x;
  if (t < enc_len[b]) return;
  float* row = x + ((size_t)b * T + t) * V;
  for (int c = threadIdx.x; c < V; c += 256) row[c] = c == 0 ? 0.0f : kNeg;
}

// Initial state (ctc.py:103-116): r[t][nb] = -1e20, r[t][b] = cumsum_t x[b,t,blank], for every beam.
// phi[n][t] = (logsumexp(r[t]), r[t][b]).
__global__ void ctc_init_kernel(const float* __restrict__ x, float* __restrict__ phi, float* __restrict__ psi_prev,
                                int B, int T, int V, int beam, int blank) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float cum = 0.0f;
  for (int t = 0; t < T; ++t) {
    cum += x[((size_t)b * T + t) * V + blank];
    const float rs = lse2(kNeg, cum);
    for (int j = 0; j < beam; ++j) {
      const size_t o = (((size_t)b * beam + j) * T + t) * 2;
      phi[o] = rs;
      phi[o + 1] = cum;
    }
  }
  for (int j = 0; j < beam; ++j) psi_prev[b * beam + j] = 0.0f;
}

struct CtcStepArgs {
  const float* x;         // [B,T,V] masked log-posteriors
  const float* phi;       // [n_bh,T,2] (r_sum, r_blank) of each hypothesis' prefix
  const float* psi_prev;  // [n_bh]
  const int32_t* last_tok;  // [n_bh]
  const int32_t* enc_len;   // [B]
  const float* am;        // [n_bh,V] acoustic-model log-probs (already * attn_weight), or null
  float* comb;            // [n_bh,V] out: am' + w * (psi - psi_prev)
  float* psi;             // [n_bh,V] out
  int B, T, V, beam, prefix_len, blank, eos;
  float weight;
  // modifications of the AM scores applied before the scorer (seq2seq.py:995-1017, scorer.py:1250)
  int eos_floor;          // 1: step < min_decode_steps -> am[eos] = minus_inf
  int use_eos_threshold;
  float eos_threshold, minus_inf;
  const float* am_max;    // [n_bh] max over V of am (only when use_eos_threshold)
};

__global__ void __launch_bounds__(256) ctc_score_step_kernel(CtcStepArgs a) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const bool c_ok = c < a.V;
  const int cc = c_ok ? c : a.V - 1;
  const int T = a.T, V = a.V;
  const int start = a.prefix_len > 1 ? a.prefix_len : 1;
  const float* xb = a.x + (size_t)b * T * V;
  const int last_frame = a.enc_len[b] - 1;

  for (int j0 = 0; j0 < a.beam; j0 += kBT) {
    const int nb = min(kBT, a.beam - j0);
    float r_nb[kBT], r_b[kBT], pm[kBT], ps[kBT];
    bool same[kBT];
#pragma unroll
    for (int j = 0; j < kBT; ++j) {
      r_nb[j] = (a.prefix_len == 0) ? xb[cc] : kNeg;  // r[start-1][nb]: x[0] at the first step, else untouched
      r_b[j] = kNeg;
      pm[j] = r_nb[j];  // running logsumexp of {psi_init, phix[start..]} as (max, sum)
      ps[j] = 1.0f;
      same[j] = j < nb && a.last_tok[b * a.beam + j0 + j] == cc;
    }
    for (int t = start; t < T; ++t) {
      const float x_nb = xb[(size_t)t * V + cc];
      const float x_b = xb[(size_t)t * V + a.blank];
#pragma unroll
      for (int j = 0; j < kBT; ++j) {
        if (j < nb) {
          const float* ph = a.phi + (((size_t)b * a.beam + j0 + j) * T + (t - 1)) * 2;
          const float phi_prev = same[j] ? ph[1] : ph[0];
          const float n_nb = lse2(r_nb[j], phi_prev) + x_nb;
          const float n_b = lse2(r_nb[j], r_b[j]) + x_b;
          r_nb[j] = n_nb;
          r_b[j] = n_b;
          const float v = phi_prev + x_nb;
          if (v > pm[j]) {
            ps[j] = ps[j] * expf(pm[j] - v) + 1.0f;
            pm[j] = v;
          } else {
            ps[j] += expf(v - pm[j]);
          }
        }
      }
    }
    if (!c_ok) continue;
#pragma unroll
    for (int j = 0; j < kBT; ++j) {
      if (j < nb) {
        const int n = b * a.beam + j0 + j;
        float psi = pm[j] + logf(ps[j]);
        if (c == a.eos) psi = a.phi[((size_t)n * T + last_frame) * 2];
        if (c == a.blank && a.eos != a.blank) psi = kNeg;
        a.psi[(size_t)n * V + c] = psi;
        float am = 0.0f;
        if (a.am) {
          am = a.am[(size_t)n * V + c];
          if (c == a.eos) {
            if (a.eos_floor) am = a.minus_inf;
            if (a.use_eos_threshold && !(am > a.eos_threshold * a.am_max[n])) am = a.minus_inf;
          }
          if (c == a.blank) am = kNeg;
        }
        a.comb[(size_t)n * V + c] = am + (psi - a.psi_prev[n]) * a.weight;
      }
    }
  }
}

// New forward variables of hypothesis n = (parent hyp, token) chosen by the beam search.
__global__ void ctc_advance_kernel(const float* __restrict__ x, const float* __restrict__ phi_old,
                                   const float* __restrict__ psi, const int32_t* __restrict__ parent,
                                   const int32_t* __restrict__ token, const int32_t* __restrict__ parent_last_tok,
                                   float* __restrict__ phi_new, float* __restrict__ psi_prev_new, int n_bh, int T, int V,
                                   int beam, int prefix_len, int blank) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_bh) return;
  const int b = n / beam, p = parent[n], c = token[n];
  const float* xb = x + (size_t)b * T * V;
  const bool same = parent_last_tok[p] == c;
  const int start = prefix_len > 1 ? prefix_len : 1;
  float* out = phi_new + (size_t)n * T * 2;
  float r_nb = (prefix_len == 0) ? xb[c] : kNeg, r_b = kNeg;
  for (int t = 0; t < start; ++t) {  // frames before `start` keep r = -1e20 (except r[0][nb] at the first step)
    const float a_nb = (t == start - 1) ? r_nb : kNeg;
    out[2 * t] = lse2(a_nb, kNeg);
    out[2 * t + 1] = kNeg;
  }
  for (int t = start; t < T; ++t) {
    const float* ph = phi_old + ((size_t)p * T + (t - 1)) * 2;
    const float phi_prev = same ? ph[1] : ph[0];
    const float n_nb = lse2(r_nb, phi_prev) + xb[(size_t)t * V + c];
    const float n_b = lse2(r_nb, r_b) + xb[(size_t)t * V + blank];
    r_nb = n_nb;
    r_b = n_b;
    out[2 * t] = lse2(r_nb, r_b);
    out[2 * t + 1] = r_b;
  }
  psi_prev_new[n] = psi[(size_t)p * V + c];
}

// Without a CTC scorer: comb = am with the eos modifications only.
__global__ void __launch_bounds__(256) am_only_kernel(const float* __restrict__ am, float* __restrict__ comb, int V,
                                                      int eos, int eos_floor, int use_thr, float thr, float minus_inf,
                                                      const float* __restrict__ am_max) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= V) return;
  float v = am[(size_t)n * V + c];
  if (c == eos) {
    if (eos_floor) v = minus_inf;
    if (use_thr && !(v > thr * am_max[n])) v = minus_inf;
  }
  comb[(size_t)n * V + c] = v;
}

__global__ void __launch_bounds__(256) row_max_kernel(const float* __restrict__ x, float* __restrict__ out, int V) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * V;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, xr[c]);
  m = sbk::wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace

namespace sbk {

int ctc_prepare(float* x, const int32_t* enc_len, float* phi, float* psi_prev, int B, int T, int V, int beam, int blank,
                hipStream_t st) {
  SBK_LAUNCH(ctc_mask_kernel, dim3(T, B), dim3(256), 0, st, x, enc_len, T, V);
  int rc = launch_status("ctc_mask");
  if (rc) return rc;
  SBK_LAUNCH(ctc_init_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, (const float*)x, phi, psi_prev, B, T, V, beam,
             blank);
  return launch_status("ctc_init");
}

int ctc_score_step(const float* x, const float* phi, const float* psi_prev, const int32_t* last_tok,
                   const int32_t* enc_len, const float* am, float* comb, float* psi, int B, int T, int V, int beam,
                   int prefix_len, int blank, int eos, float weight, int eos_floor, int use_thr, float thr,
                   float minus_inf, const float* am_max, hipStream_t st) {
  CtcStepArgs a{x, phi, psi_prev, last_tok, enc_len, am, comb, psi, B, T, V, beam, prefix_len, blank, eos, weight,
                eos_floor, use_thr, thr, minus_inf, am_max};
  ProfScope prof("ctc_score_step", 10.0 * B * beam * (double)T * V, 4.0 * B * (double)T * V + 12.0 * B * beam * V, st);
  SBK_LAUNCH(ctc_score_step_kernel, dim3(cdiv(V, 256), B), dim3(256), 0, st, a);
  return launch_status("ctc_score_step");
}

int ctc_advance(const float* x, const float* phi_old, const float* psi, const int32_t* parent, const int32_t* token,
                const int32_t* parent_last_tok, float* phi_new, float* psi_prev_new, int n_bh, int T, int V, int beam,
                int prefix_len, int blank, hipStream_t st) {
  ProfScope prof("ctc_advance", 10.0 * n_bh * T, 16.0 * n_bh * T, st);
  SBK_LAUNCH(ctc_advance_kernel, dim3(cdiv(n_bh, 64)), dim3(64), 0, st, x, phi_old, psi, parent, token,
             parent_last_tok, phi_new, psi_prev_new, n_bh, T, V, beam, prefix_len, blank);
  return launch_status("ctc_advance");
}

int am_only(const float* am, float* comb, int n_bh, int V, int eos, int eos_floor, int use_thr, float thr,
            float minus_inf, const float* am_max, hipStream_t st) {
  SBK_LAUNCH(am_only_kernel, dim3(cdiv(V, 256), n_bh), dim3(256), 0, st, am, comb, V, eos, eos_floor, use_thr, thr,
             minus_inf, am_max);
  return launch_status("am_only");
}

int row_max(const float* x, float* out, int rows, int V, hipStream_t st) {
  SBK_LAUNCH(row_max_kernel, dim3(rows), dim3(256), 0, st, x, out, V);
  return launch_status("row_max");
}

}  // namespace sbk
