// CTC prefix scorer (Watanabe et al. 2017, Alg. 2), full-vocabulary mode.
// Replaces decoders/ctc.py:26-295 (CTCPrefixScore) as driven by scorer.py:108-255.
//
// The reference materialises r[T,2,n_bh,V] every step (10 MB per hypothesis), walks the
// frames in a Python loop and evaluates every recurrence in the log domain (5 exp/log per
// (hypothesis, token, frame)).  Here
//  * the state kept per hypothesis is only its own forward variables per frame, stored as
//    block-floating-point pairs  exp(gamma[t]) = mg * 2^eg  (gamma = log-sum of the
//    non-blank/blank variables) and  exp(beta[t]) = mb * 2^eb  (beta = blank variable);
//  * emissions are kept as linear probabilities P[b,t,c] (frames past the utterance end:
//    P = 0 except column 0 = 1, the linear image of ctc.py:57-61);
//  * ctc_score_step: the score of extending prefix g by token c needs only g's state,
//        psi(g.c) = log( r_init + sum_t phi_g[t-1] * P_c[t] ),   phi = gamma (beta if c repeats g's last token)
//    i.e. a [beams x T] . [T x V] product per utterance with an enormous dynamic range along T.
//    The state tables therefore also carry the phi values pre-scaled per 32-frame segment
//    (value * 2^-segment_exponent, exact power-of-two scaling); a thread (<-> token c of one
//    utterance, ALL beams in registers) does ONE fma per (beam, frame) and folds each segment's
//    partial sum into a block-float accumulator.  The repeated-token entries (one per
//    hypothesis) are recomputed from the beta table by ctc_same_token_kernel.
//    The emission row P[b,t,:] is read ONCE per step and utterance (coalesced along V):
//    algorithmic traffic T*V*4 B per utterance per step -- the kernel is HBM/L2-bound.
//  * ctc_advance: after the beam top-k picked (parent, token) for each new hypothesis, the
//    forward recurrences
//        Rnb[t] = (Rnb[t-1] + phi[t-1]) * P_c[t],   Rb[t] = (Rnb[t-1] + Rb[t-1]) * P_blank[t]
//    are evaluated for that single pair as a wave-parallel prefix scan over per-frame affine maps
//    (block-float) to obtain the survivor's per-frame state (the reference gathers it out of the
//    materialised tensor, ctc.py:243-295).
// Values the reference represents with its finite -1e20 sentinel (ctc.py:53) are exact
// zeros here and map back to -1e20 whenever a log-domain number leaves the kernels.
#include "common.h"
#include "internal.h"

namespace {

constexpr float kNeg = sbk::kCtcNeg;
constexpr int kNegE = -(1 << 20);  // exponent of an exact zero
constexpr int kHead = 24;          // head-room (bits) kept above the incoming phi term

struct BF {  // exp(gamma) = mg * 2^eg, exp(beta) = mb * 2^eb   (16 bytes per (hypothesis, frame))
  float mg, mb;
  int eg, eb;
};

// natural log of m * 2^e (m >= 0) in f64, rounded once
__device__ __forceinline__ float bf_log(float m, int e) {
  if (!(m > 0.0f)) return kNeg;
  const double v = ((double)e + (double)log2f(m)) * 0.69314718055994530942;
  return v < (double)kNeg ? kNeg : (float)v;
}
// block-float image of exp(x), x a natural-log value (<= ~0)
__device__ __forceinline__ void bf_exp(float x, float* m, int* e) {
  if (!(x > -1e19f)) {
    *m = 0.0f;
    *e = kNegE;
    return;
  }
  const double y = (double)x * 1.44269504088896340736;
  const double ip = floor(y);
  *m = exp2f((float)(y - ip));
  *e = (int)ip;
}

// Segment-scaled phi tables used by ctc_score_step: for hypothesis h and frame t,
//   sg[h][t] = exp(gamma_h[t]) * 2^-seg_e[h][t/32],  sb[h][t] = exp(beta_h[t]) * 2^-seg_e[h][t/32]
// with seg_e = the largest exponent inside the segment (so every entry is < 2).
constexpr int kSeg = 32;
__device__ __host__ __forceinline__ int nseg_of(int T) { return (T + kSeg - 1) / kSeg; }
__device__ __host__ __forceinline__ int beam_pitch(int beam) { return 16 * ((beam + 15) / 16); }  // table row pitch: beams in tiles of 16 (4 float4 per frame and tile)

// One wave converts the BF rows of one hypothesis (already in global memory / LDS as `row`) into the
// scaled tables.  `row(t)` returns the BF of frame t.
// sg / se are written with stride `bp` (the gamma table is stored [utterance][frame][beam], beam
// fastest and padded to a multiple of 4, so that the score kernel fetches all beams of a frame with
// a few wide scalar loads); sb is a plain per-hypothesis row.
template <typename Row>
__device__ __forceinline__ void build_segment_tables(Row row, int T, int lane, float* __restrict__ sg,
                                                     float* __restrict__ sb, int* __restrict__ se, int bp) {
  const int nseg = nseg_of(T);
  for (int s0 = 0; s0 < nseg; s0 += 2) {  // two 32-frame segments per pass: lanes 0-31 and 32-63
    const int seg = s0 + (lane >> 5);
    const int t = seg * kSeg + (lane & 31);
    BF v{0.0f, 0.0f, kNegE, kNegE};
    if (seg < nseg && t < T) v = row(t);
    int e = max(v.mg > 0.0f ? v.eg : kNegE, v.mb > 0.0f ? v.eb : kNegE);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) e = max(e, sbk::shfl_xor(e, m));  // max within the 32-lane half
    if (seg < nseg && t < T) {
      sg[(size_t)t * bp] = sbk::fast_ldexp(v.mg, v.eg - e);
      sb[t] = sbk::fast_ldexp(v.mb, v.eb - e);
    }
    if (seg < nseg && (lane & 31) == 0) se[(size_t)seg * bp] = e;
  }
}

// ---- emissions: log_softmax rows -> linear probabilities with the reference's frame mask
__global__ void __launch_bounds__(256) ctc_emissions_kernel(float* __restrict__ x, float* __restrict__ xb_log,
                                                            const int32_t* __restrict__ enc_len, int T, int V, int blank, int ldp) {
  const int b = blockIdx.y, t = blockIdx.x;
  float* row = x + ((size_t)b * T + t) * ldp;
  const bool pad = t >= enc_len[b];
  if (threadIdx.x == 0) xb_log[b * T + t] = pad ? (blank == 0 ? 0.0f : kNeg) : row[blank];
  __syncthreads();
  for (int c = threadIdx.x; c < V; c += 256) row[c] = pad ? (c == 0 ? 1.0f : 0.0f) : expf(row[c]);
}

// ---- initial state (ctc.py:103-116): r[t][nb] = -1e20, r[t][b] = cumsum_t x[b,t,blank], for every beam.
// One workgroup per utterance: the cumulative sum is serial (same order as torch.cumsum), the
// block-float conversion and the per-beam replication are parallel over frames.
__global__ void __launch_bounds__(256) ctc_init_kernel(const float* __restrict__ xb_log, BF* __restrict__ st,
                                                       float* __restrict__ sg, float* __restrict__ sb,
                                                       int* __restrict__ se, float* __restrict__ psi_prev, int T,
                                                       int beam) {
  SBK_DYN_LDS(float, cum);
  const int b = blockIdx.x;
  const int nseg = nseg_of(T);
  if (threadIdx.x == 0) {
    float c = 0.0f;
    for (int t = 0; t < T; ++t) {
      c += xb_log[b * T + t];
      cum[t] = c;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    BF v;
    bf_exp(cum[t], &v.mg, &v.eg);
    v.mb = v.mg;
    v.eb = v.eg;
    for (int j = 0; j < beam; ++j) st[((size_t)b * beam + j) * T + t] = v;
  }
  if (threadIdx.x < beam) psi_prev[b * beam + threadIdx.x] = 0.0f;
  __syncthreads();
  // scaled tables: wave w builds beams w, w+4, ... from the rows just written by this workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < beam; j += 4) {
    const size_t n = (size_t)b * beam + j;
    const BF* rowp = st + n * T;
    const int bp = beam_pitch(beam);
    build_segment_tables([&](int t) { return rowp[t]; }, T, lane, sg + (size_t)b * T * bp + j, sb + n * T,
                         se + (size_t)b * nseg * bp + j, bp);
  }
}

struct CtcStepArgs {
  const int32_t* last_tok;  // [n_bh]
  const int32_t* enc_len;   // [B]
  int B, T, V, beam, prefix_len, blank, eos;
  float weight;
  // modifications of the AM scores applied before the scorer (seq2seq.py:995-1017, scorer.py:1250)
  int eos_floor;            // 1: step < min_decode_steps -> am[eos] = minus_inf
  int use_eos_threshold;
  float eos_threshold, minus_inf;
  const int32_t* step_ptr;  // non-null: prefix_len / eos_floor come from the device-side step counter
  int min_steps;
  // grouped search: per-utterance min_decode_steps (overrides eos_floor / min_steps); row n belongs to utterance n / beam
  const int32_t* utt_min;
  int utt_beam, step;
  // attention window (ctc.py:189-200, ctc_window_size > 0): win = {min, max} of the attention peaks of this step
  // (device memory), frames [max(start, win[0] - window), min(T, win[1] + window)) are scored; NULL = every frame
  const int32_t* win;
  int window;
  int ldp;  // row pitch of P in floats (>= V; the search pads it to whole 128-byte lines: a workgroup's 1 KB segment of a frame row
            // then is 8 lines instead of 9 -- 20 000-byte rows at V = 5 000 are not line-aligned)
};
// scored frame range [start, end) of a step (ctc.py:187-200)
__device__ __forceinline__ void ctc_frame_range(int prefix_len, int T, const int32_t* win, int window, int* start, int* end) {
  *start = prefix_len > 1 ? prefix_len : 1;
  *end = T;
  if (win) {
    *start = max(*start, win[0] - window);
    *end = min(T, win[1] + window);
  }
}
// The frames of an utterance past its own length hold P = 1 for token 0 and EXACTLY 0 for every other token (ctc_emissions_kernel:
// the reference's frame mask), so they add exact zeros to the score of every token but 0 -- and when token 0 is the blank, its
// score is overwritten in any case (the blank column / the <eos> rule).  The frame loops of the score kernels therefore stop at
// the utterance's own length: bit-identical results, fewer bytes for the shorter utterances of a padded batch.
__device__ __forceinline__ int ctc_clip_to_length(int end, int blank, int len) { return blank == 0 ? min(end, max(len, 1)) : end; }

// P [B,T,V] masked linear posteriors; sg [n_bh,T] / se [n_bh,nseg] segment-scaled gamma tables
// (uniform per workgroup: fetched through the scalar cache); am [n_bh,V] acoustic log-probs
// (already * attn weight); outputs comb = am' + w * (psi - psi_prev) and psi, both [n_bh,V].
// TPT = 4 takes FOUR CONSECUTIVE vocabulary entries per thread and reads the emissions as float4 (needs V % 4 == 0);
// TPT = 1 / 2 take entries 256 apart with scalar loads.
template <int NB, int TPT, bool NT>  // beams per tile in registers; tokens per thread (each table read feeds NB*TPT FMAs); NT: non-temporal loads of P
__global__ void __launch_bounds__(256) ctc_score_step_kernel(CtcStepArgs a, const float* __restrict__ P,
                                                             const BF* __restrict__ st, const float* __restrict__ sg,
                                                             const int* __restrict__ se,
                                                             float* __restrict__ psi_out) {
  constexpr int BP = 16;               // LDS row pitch = one tile of 16 beams (blockIdx.z selects the tile)
  constexpr int NV = (NB + 3) / 4;     // float4 per frame actually consumed
  SBK_DYN_LDS(float, lds);
  if (a.step_ptr) a.prefix_len = a.step_ptr[0];
  const int b = blockIdx.y;
  const int j0 = blockIdx.z * BP, bp = beam_pitch(a.beam);
  const int T = a.T, V = a.V;
  int c[TPT], cc[TPT];
  bool c_ok[TPT];
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    c[k] = TPT == 4 ? (blockIdx.x * 256 + threadIdx.x) * 4 + k : (blockIdx.x * TPT + k) * 256 + threadIdx.x;
    c_ok[k] = c[k] < V;
    cc[k] = c_ok[k] ? c[k] : V - 1;
  }
  const int c4 = c_ok[0] ? c[0] : V - 4;  // TPT = 4: base of this thread's float4 (V % 4 == 0)
  const int nseg = nseg_of(T);
  int start, end;
  ctc_frame_range(a.prefix_len, T, a.win, a.window, &start, &end);
  end = ctc_clip_to_length(end, a.blank, a.enc_len[b]);
  const int ldp = a.ldp;
  const float* Pb = P + (size_t)b * T * ldp;
  const int last_frame = a.enc_len[b] - 1;
  // the utterance's scaled gamma table [T][16] and segment exponents [nseg][16] -> LDS (coalesced copy);
  // every lane then reads the same address per frame (broadcast, conflict-free)
  float* tab = lds;
  int* segs = reinterpret_cast<int*>(lds + (size_t)T * BP);
  {
    float4* dst = reinterpret_cast<float4*>(tab);
    for (int i = threadIdx.x; i < T * (BP / 4); i += 256)
      dst[i] = *reinterpret_cast<const float4*>(sg + ((size_t)b * T + i / (BP / 4)) * bp + j0 + 4 * (i % (BP / 4)));
    const int* es = se + (size_t)b * nseg * bp + j0;
    for (int i = threadIdx.x; i < nseg * BP; i += 256) segs[i] = es[(size_t)(i / BP) * bp + i % BP];
  }
  __syncthreads();

  // The prefix score of h = g.c needs only g's state:  psi = log( r_init + sum_t phi_g[t-1] * P_c[t] )
  // (ctc.py:212-229).  Block-float accumulator per (token, beam): value = mps * 2^Eps.  (beam <= NB; table
  // columns past `beam` are zero, so the surplus accumulators idle harmlessly.)
  float mps[TPT][NB];
  int Eps[TPT][NB];
  float part[TPT][NB];
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const float p0 = Pb[cc[k]];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      // psi_init = r[start-1][nb]: x[0] (non-blank) at the very first step, nothing otherwise (ctc.py:168-172,212)
      const bool first = a.prefix_len == 0 && start == 1 && p0 > 0.0f;
      mps[k][j] = first ? p0 : 0.0f;
      Eps[k][j] = first ? 0 : kNegE;
      part[k][j] = 0.0f;
    }
  }
  // phi[t-1] * P[t]: frame t uses table entry u = t-1.  Chunks of CH table entries; the emission values of
  // the NEXT chunk are requested before the current chunk is accumulated (two chunks in flight per lane).
  constexpr int CH = 16 / TPT < 8 ? 8 : 16 / TPT;
  static_assert(kSeg % CH == 0, "ctc_score_step: chunks must tile a scale segment");
  const int u_begin = start - 1, u_end = end - 1;
  auto fetch = [&](float (&buf)[TPT][CH], int cb) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int u = cb + q;
      const bool ok = u >= u_begin && u < u_end;
      if constexpr (TPT == 4) {
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v4 = sbk::ld16<NT>(Pb + (size_t)(u + 1) * ldp + c4);
        buf[0][q] = v4.x; buf[1][q] = v4.y; buf[2][q] = v4.z; buf[3][q] = v4.w;
      } else {
#pragma unroll
        for (int k = 0; k < TPT; ++k) buf[k][q] = ok ? sbk::ld4<NT>(Pb + (size_t)(u + 1) * ldp + cc[k]) : 0.0f;
      }
    }
  };
  float nxt[TPT][CH];
  int cb = (u_begin / CH) * CH;
  if (cb < u_end) fetch(nxt, cb);
  for (; cb < u_end; cb += CH) {
    float cur[TPT][CH];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
#pragma unroll
      for (int q = 0; q < CH; ++q) cur[k][q] = nxt[k][q];
    if (cb + CH < u_end) fetch(nxt, cb + CH);
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const float4* row = reinterpret_cast<const float4*>(tab + (size_t)min(cb + q, u_end - 1) * BP);
      float g[NV * 4];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 t4 = row[v];
        g[4 * v] = t4.x; g[4 * v + 1] = t4.y; g[4 * v + 2] = t4.z; g[4 * v + 3] = t4.w;
      }
#pragma unroll
      for (int k = 0; k < TPT; ++k)
#pragma unroll
        for (int j = 0; j < NB; ++j) part[k][j] = fmaf(g[j], cur[k][q], part[k][j]);
    }
    if (((cb + CH) % kSeg) == 0 || cb + CH >= u_end) {  // end of a scale segment: fold into the block-float sum
      const int* es = segs + (cb / kSeg) * BP;
#pragma unroll
      for (int k = 0; k < TPT; ++k)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int kx = sbk::frexp_exp(part[k][j]);
          const int ep = part[k][j] > 0.0f ? es[j] + kx : kNegE;
          const float mp = sbk::fast_ldexp(part[k][j], -kx);
          const int P2 = max(Eps[k][j], ep);
          mps[k][j] = sbk::fast_ldexp(mps[k][j], Eps[k][j] - P2) + sbk::fast_ldexp(mp, ep - P2);
          Eps[k][j] = P2;
          part[k][j] = 0.0f;
        }
    }
  }
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    if (!c_ok[k]) continue;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j0 + j < a.beam) {
        const int n = b * a.beam + j0 + j;
        float psi = bf_log(mps[k][j], Eps[k][j]);
        if (c[k] == a.eos) {  // psi[eos] = log-sum of the prefix' own variables at the last frame (ctc.py:232-235)
          const BF s = st[(size_t)n * T + last_frame];
          psi = bf_log(s.mg, s.eg);
        }
        if (c[k] == a.blank && a.eos != a.blank) psi = kNeg;
        psi_out[(size_t)n * V + c[k]] = psi;
      }
    }
  }
}

// (The same score on v_mfma_f32_16x16x4_f32 -- knob 7 = 8 of round 4 -- was bit-identical and no faster: 255 vs 252-269 us, the
// kernel is bound by its HBM stream at 3.7-4 TB/s, not by the LDS broadcast reads the counters had pointed at; two and four
// tokens per thread -- knob 7 = 2 / 4 -- were within noise: profiles/r04_i_*.  Both removed in round 5.)
// The one token per hypothesis that repeats the prefix' last token uses phi = beta instead of
// gamma (ctc.py:175-186): recompute that entry.  One wave per hypothesis, lanes over frames.
__global__ void __launch_bounds__(64) ctc_same_token_kernel(CtcStepArgs a, const float* __restrict__ P,
                                                            const float* __restrict__ sb, const int* __restrict__ se,
                                                            float* __restrict__ psi_out) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (a.step_ptr) a.prefix_len = a.step_ptr[0];
  const int b = n / a.beam, c = a.last_tok[n];
  const int T = a.T, V = a.V, nseg = nseg_of(T);
  if (c == a.eos || c == a.blank || c < 0 || c >= V) return;  // those entries are overridden anyway
  int start, end;
  ctc_frame_range(a.prefix_len, T, a.win, a.window, &start, &end);
  end = ctc_clip_to_length(end, a.blank, a.enc_len[b]);
  const float* Pb = P + (size_t)b * T * a.ldp;
  // per-lane block-float partial over its frames, then a wave reduction on a common exponent
  float m = 0.0f;
  int E = kNegE;
  if (lane == 0 && a.prefix_len == 0 && start == 1 && Pb[c] > 0.0f) {
    m = Pb[c];
    E = 0;
  }
  for (int u = start - 1 + lane; u < end - 1; u += 64) {
    const float term = sb[(size_t)n * T + u] * Pb[(size_t)(u + 1) * a.ldp + c];
    const int k = sbk::frexp_exp(term);
    const int et = term > 0.0f ? se[((size_t)b * nseg + u / kSeg) * beam_pitch(a.beam) + n % a.beam] + k : kNegE;
    const float mt = sbk::fast_ldexp(term, -k);
    const int E2 = max(E, et);
    m = sbk::fast_ldexp(m, E - E2) + sbk::fast_ldexp(mt, et - E2);
    E = E2;
  }
  int Emax = E;
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) Emax = max(Emax, sbk::shfl_xor(Emax, k));
  float v = sbk::fast_ldexp(m, E - Emax);
  v = sbk::wave_sum(v);
  if (lane == 0) psi_out[(size_t)n * V + c] = bf_log(v, Emax);
}

// comb = am' + w * (psi - psi_prev): the scorer combination of scorer.py:1248-1253 with the AM
// modifications applied first (eos floor / eos threshold, seq2seq.py:995-1017; blank column, scorer.py:1250).
__global__ void __launch_bounds__(256) ctc_combine_kernel(CtcStepArgs a, const float* __restrict__ am,
                                                          const float* __restrict__ am_max,
                                                          const float* __restrict__ psi,
                                                          const float* __restrict__ psi_prev, float* __restrict__ comb,
                                                          const float* __restrict__ extra) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.V) return;
  float v = am[(size_t)n * a.V + c];
  if (c == a.eos) {
    const int step = a.step_ptr ? a.step_ptr[0] : a.step;
    const bool floor = a.utt_min ? step < a.utt_min[n / a.utt_beam] : (a.step_ptr ? step < a.min_steps : a.eos_floor);
    if (floor) v = a.minus_inf;
    if (a.use_eos_threshold && !(v > a.eos_threshold * am_max[n])) v = a.minus_inf;
  }
  if (extra) v = sbk::add_rn(v, extra[(size_t)n * a.V + c]);  // full scorers listed before "ctc" (already weighted)
  if (c == a.blank) v = kNeg;
  comb[(size_t)n * a.V + c] = sbk::score_ctc(v, psi[(size_t)n * a.V + c], psi_prev[n], a.weight);
}

// ---- the forward recurrence as a prefix scan ---------------------------------------------------
// One frame maps the state s = (Rnb, Rb) to  s' = A s + v  with  A = [[pc,0],[pb,pb]],  v = (pc*phi, 0).
// Such affine maps compose associatively (lower-triangular A stays lower-triangular), so the T-step
// recurrence becomes: each lane composes the maps of its own run of frames, a 6-step wave scan
// composes the lane maps, and each lane replays its run from the scanned prefix.  Maps and states
// are block-float: (a, c, d) * 2^ea, (v0, v1) * 2^ev, (nb, bl) * 2^e -- exact power-of-two scaling.
struct AMap {
  float a, c, d, v0, v1;
  int ea, ev;
};
struct AState {
  float nb, bl;
  int e;
};
__device__ __forceinline__ void norm2(float& x, float& y, int& e) {
  const float m = fmaxf(x, y);
  const int k = sbk::frexp_exp(m);
  x = sbk::fast_ldexp(x, -k);
  y = sbk::fast_ldexp(y, -k);
  e = m > 0.0f ? e + k : kNegE;
}
__device__ __forceinline__ void norm3(float& x, float& y, float& z, int& e) {
  const float m = fmaxf(fmaxf(x, y), z);
  const int k = sbk::frexp_exp(m);
  x = sbk::fast_ldexp(x, -k);
  y = sbk::fast_ldexp(y, -k);
  z = sbk::fast_ldexp(z, -k);
  e = m > 0.0f ? e + k : kNegE;
}
__device__ __forceinline__ int clamp_e(int e) { return max(e, kNegE); }
// s' = M s
__device__ __forceinline__ AState apply_map(const AMap& m, const AState& s) {
  // A s has exponent ea + e; v has exponent ev
  const int e1 = clamp_e(m.ea + s.e), e2 = m.ev;
  const int E = max(e1, e2);
  const float x = sbk::fast_ldexp(m.a * s.nb, e1 - E) + sbk::fast_ldexp(m.v0, e2 - E);
  const float y = sbk::fast_ldexp(m.c * s.nb + m.d * s.bl, e1 - E) + sbk::fast_ldexp(m.v1, e2 - E);
  AState r{x, y, E};
  norm2(r.nb, r.bl, r.e);
  return r;
}
// (m2 after m1)
__device__ __forceinline__ AMap compose(const AMap& m2, const AMap& m1) {
  AMap r;
  r.a = m2.a * m1.a;
  r.d = m2.d * m1.d;
  r.c = m2.c * m1.a + m2.d * m1.c;
  r.ea = clamp_e(m2.ea + m1.ea);
  norm3(r.a, r.c, r.d, r.ea);
  const int e1 = clamp_e(m2.ea + m1.ev), e2 = m2.ev;
  const int E = max(e1, e2);
  r.v0 = sbk::fast_ldexp(m2.a * m1.v0, e1 - E) + sbk::fast_ldexp(m2.v0, e2 - E);
  r.v1 = sbk::fast_ldexp(m2.c * m1.v0 + m2.d * m1.v1, e1 - E) + sbk::fast_ldexp(m2.v1, e2 - E);
  r.ev = E;
  norm2(r.v0, r.v1, r.ev);
  return r;
}
__device__ __forceinline__ AMap frame_map(float pc, float pb, float mphi, int ephi) {
  AMap m;
  m.a = pc;
  m.c = pb;
  m.d = pb;
  m.ea = 0;
  norm3(m.a, m.c, m.d, m.ea);
  m.v0 = pc * mphi;
  m.v1 = 0.0f;
  m.ev = ephi;
  norm2(m.v0, m.v1, m.ev);
  return m;
}
__device__ __forceinline__ AMap identity_map() { return AMap{1.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0, kNegE}; }
__device__ __forceinline__ AMap shfl_up_map(const AMap& m, int delta, int lane) {
  const int src = max(lane - delta, 0);
  AMap r;
  r.a = sbk::shfl(m.a, src);
  r.c = sbk::shfl(m.c, src);
  r.d = sbk::shfl(m.d, src);
  r.v0 = sbk::shfl(m.v0, src);
  r.v1 = sbk::shfl(m.v1, src);
  r.ea = sbk::shfl(m.ea, src);
  r.ev = sbk::shfl(m.ev, src);
  return r;
}

// New per-frame state of hypothesis n = (parent hyp, token) chosen by the beam search: one wave per
// hypothesis; emissions and parent state are prefetched into LDS, lane 0 runs the serial recurrence,
// all lanes normalise and store.
struct CtcAdvArgs {
  const float* P;
  const BF* st_old;
  const float* psi;
  const int32_t* parent;
  const int32_t* token;
  const int32_t* parent_last_tok;
  BF* st_new;
  float* sg_new;
  float* sb_new;
  int* se_new;
  float* psi_prev_new;
  int n_bh, T, V, beam, prefix_len, blank;
  const int32_t* step_ptr;
  const int32_t* win;  // attention window of this step (see CtcStepArgs)
  int window;
  int ldp;             // row pitch of P in floats
};

__global__ void __launch_bounds__(64) ctc_advance_kernel(CtcAdvArgs a) {
  SBK_DYN_LDS(float, lds);
  if (a.step_ptr) a.prefix_len = a.step_ptr[0];
  const int T = a.T;
  float* pc = lds;              // [T] P[t][token]
  float* pb = pc + T;           // [T] P[t][blank]
  float* mph = pb + T;          // [T] phi mantissa
  int* eph = reinterpret_cast<int*>(mph + T);  // [T] phi exponent
  float* onb = reinterpret_cast<float*>(eph + T);  // [T] results
  float* obl = onb + T;
  int* oe = reinterpret_cast<int*>(obl + T);
  const int n = blockIdx.x, lane = threadIdx.x;
  const int b = n / a.beam, p = a.parent[n], c = a.token[n];
  const bool same = a.parent_last_tok[p] == c;
  const float* Pb = a.P + (size_t)b * T * a.ldp;
  for (int tb = 0; tb < T; tb += 256) {  // 4 frames per lane per round: 12 loads in flight per lane
    float vpc[4], vpb[4];
    BF vs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(tb + u * 64 + lane, T - 1);
      vpc[u] = Pb[(size_t)t * a.ldp + c];
      vpb[u] = Pb[(size_t)t * a.ldp + a.blank];
      vs[u] = a.st_old[(size_t)p * T + t];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tb + u * 64 + lane;
      if (t < T) {
        pc[t] = vpc[u];
        pb[t] = vpb[u];
        mph[t] = same ? vs[u].mb : vs[u].mg;
        eph[t] = same ? vs[u].eb : vs[u].eg;
      }
    }
  }
  __syncthreads();
  {
    int start, end;
    ctc_frame_range(a.prefix_len, T, a.win, a.window, &start, &end);
    const bool first = a.prefix_len == 0 && pc[0] > 0.0f;
    AState r0{first ? pc[0] : 0.0f, 0.0f, first ? 0 : kNegE};  // r[0]: x[0][c] at the first step (ctc.py:168-172)
    norm2(r0.nb, r0.bl, r0.e);
    const AState zero{0.0f, 0.0f, kNegE};
    const AState s0 = start == 1 ? r0 : zero;  // r[start-1]: frames outside the recurrence stay "minus infinity"
    for (int t = lane; t < T; t += 64)
      if (t < start || t >= end) {  // (an attention window may leave frames on either side untouched)
        onb[t] = t == 0 ? r0.nb : 0.0f;
        obl[t] = 0.0f;
        oe[t] = t == 0 ? r0.e : kNegE;
      }
    // lane l owns frames [f0, f1)
    const int nfr = max(end - start, 0);
    const int fpl = (nfr + 63) / 64;
    const int f0 = min(start + lane * fpl, end), f1 = min(f0 + fpl, end);
    AMap mine = identity_map();
    for (int t = f0; t < f1; ++t) mine = compose(frame_map(pc[t], pb[t], mph[t - 1], eph[t - 1]), mine);
    // inclusive scan of the lane maps (Hillis-Steele)
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const AMap other = shfl_up_map(mine, dlt, lane);
      if (lane >= dlt) mine = compose(mine, other);
    }
    // exclusive prefix = the scanned map of lane-1; replay this lane's frames from it
    AMap pre = shfl_up_map(mine, 1, lane);
    if (lane == 0) pre = identity_map();
    AState st = apply_map(pre, s0);
    for (int t = f0; t < f1; ++t) {
      st = apply_map(frame_map(pc[t], pb[t], mph[t - 1], eph[t - 1]), st);
      onb[t] = st.nb;
      obl[t] = st.bl;
      oe[t] = st.e;
    }
  }
  __syncthreads();
  BF* bfrow = reinterpret_cast<BF*>(pc);  // reuse the (now dead) input area: 4 floats per frame
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    BF s{0.0f, 0.0f, kNegE, kNegE};
    if (t < T) {
      const float g = onb[t] + obl[t];
      const int kg = sbk::frexp_exp(g), kb = sbk::frexp_exp(obl[t]);
      s.mg = g > 0.0f ? sbk::fast_ldexp(g, 1 - kg) : 0.0f;  // mantissa in [1,2)
      s.eg = g > 0.0f ? oe[t] + kg - 1 : kNegE;
      s.mb = obl[t] > 0.0f ? sbk::fast_ldexp(obl[t], 1 - kb) : 0.0f;
      s.eb = obl[t] > 0.0f ? oe[t] + kb - 1 : kNegE;
      a.st_new[(size_t)n * T + t] = s;
    }
    __syncthreads();  // everyone has read onb/obl/oe of this round before pc.. is overwritten
    if (t < T) bfrow[t] = s;
  }
  __syncthreads();
  const int nseg = nseg_of(T);
  const int bp = beam_pitch(a.beam);
  build_segment_tables([&](int t) { return bfrow[t]; }, T, lane, a.sg_new + (size_t)b * T * bp + n % a.beam,
                       a.sb_new + (size_t)n * T, a.se_new + (size_t)b * nseg * bp + n % a.beam, bp);
  if (lane == 0) a.psi_prev_new[n] = a.psi[(size_t)p * a.V + c];
}

// Without a CTC scorer: comb = am with the eos modifications only.
__global__ void __launch_bounds__(256) am_only_kernel(const float* __restrict__ am, float* __restrict__ comb, int V,
                                                      int eos, int eos_floor, int use_thr, float thr, float minus_inf,
                                                      const float* __restrict__ am_max,
                                                      const float* __restrict__ extra,
                                                      const int32_t* __restrict__ step_ptr, int min_steps,
                                                      const int32_t* __restrict__ utt_min, int beam, int step) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= V) return;
  float v = am[(size_t)n * V + c];
  if (c == eos) {
    if (step_ptr) step = step_ptr[0];
    if (utt_min ? step < utt_min[n / beam] : (step_ptr ? step < min_steps : eos_floor)) v = minus_inf;
    if (use_thr && !(v > thr * am_max[n])) v = minus_inf;
  }
  if (extra) v = sbk::add_rn(v, extra[(size_t)n * V + c]);
  comb[(size_t)n * V + c] = v;
}

// ---- CTC as a PARTIAL scorer (scorer.py:1280-1300, ctc.py:168-262 with `candidates`) -----------------------------------
// The reference scores only the top `k` tokens of each hypothesis (after the attention log-probs, the eos rules and
// the full scorers) and gives every other token minus_inf; <eos> always gets its score, blank never does.  Here the
// dense psi of the full scorer is computed anyway (it is one matrix product), so the partial scorer is a mask:
// thr[n] = k-th largest entry of row n, found by bisection over the order-preserving integer image of the floats.
__device__ __forceinline__ unsigned order_key(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256) row_kth_largest_kernel(const float* __restrict__ x, float* __restrict__ thr, int V,
                                                              int k) {
  __shared__ int cnt[4];
  __shared__ unsigned prefix_s;
  const float* xr = x + (size_t)blockIdx.x * V;
  unsigned prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned t = prefix | (1u << bit);
    int c = 0;
    for (int i = threadIdx.x; i < V; i += 256) c += order_key(xr[i]) >= t ? 1 : 0;
    c = (int)sbk::wave_sum((float)c);  // counts <= V < 2^24: exact in fp32
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) prefix_s = ((cnt[0] + cnt[1]) + (cnt[2] + cnt[3])) >= k ? t : prefix;
    __syncthreads();
    prefix = prefix_s;
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // back from the key to the float
    const unsigned u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
    thr[blockIdx.x] = __uint_as_float(u);
  }
}

// comb (= the attention log-probs after the eos rules and the full scorers) += w * (masked psi - psi_prev)
__global__ void __launch_bounds__(256) ctc_partial_combine_kernel(float* __restrict__ comb, const float* __restrict__ thr,
                                                                  const float* __restrict__ psi,
                                                                  const float* __restrict__ psi_prev, int V, int blank,
                                                                  int eos, float weight, float minus_inf) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= V) return;
  const float base = comb[(size_t)n * V + c];
  float p = (base >= thr[n] || c == eos) ? psi[(size_t)n * V + c] : minus_inf;
  if (c == blank && eos != blank) p = minus_inf;
  comb[(size_t)n * V + c] = base + (p - psi_prev[n]) * weight;
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

namespace {
struct StateView {
  BF* st;
  float* sg;
  float* sb;
  int* se;
};
StateView view(float* base, int B, int beam, int T) {
  const size_t n_bh = (size_t)B * beam, bp = beam_pitch(beam);
  StateView v;
  v.st = reinterpret_cast<BF*>(base);
  v.sg = base + 4 * n_bh * T;
  v.sb = v.sg + (size_t)B * T * bp;
  v.se = reinterpret_cast<int*>(v.sb + n_bh * T);
  return v;
}
}  // namespace

// floats of one CTC state buffer: BF rows + the two segment-scaled tables + segment exponents
int ctc_psi_step(const float* P, const float* state, const int32_t* last_tok, const int32_t* enc_len, float* psi, int B,
                 int T, int V, int beam, int prefix_len, int blank, int eos, hipStream_t st, const int32_t* win, int window, int ldp);

size_t ctc_state_floats(int B, int beam, int T) {
  const size_t n_bh = (size_t)B * beam, bp = beam_pitch(beam);
  return 4 * n_bh * T + (size_t)B * T * bp + n_bh * T + (size_t)B * nseg_of(T) * bp + 16;
}

// x: [B,T,V] log_softmax(ctc_lin(enc)) on entry, linear masked posteriors on exit.
int ctc_prepare(float* x, float* xb_log, const int32_t* enc_len, float* state, float* psi_prev, int B, int T, int V,
                int beam, int blank, hipStream_t st, int ldp) {
  if (ldp < V) ldp = V;
  SBK_LAUNCH(ctc_emissions_kernel, dim3(T, B), dim3(256), 0, st, x, xb_log, enc_len, T, V, blank, ldp);
  int rc = launch_status("ctc_emissions");
  if (rc) return rc;
  StateView v = view(state, B, beam, T);
  // columns past `beam` of the [frame][16] tables are never written afterwards and must read as zero
  if (hipMemsetAsync(v.sg, 0, (size_t)B * T * beam_pitch(beam) * sizeof(float), st) != hipSuccess) return fail(1, "ctc: memset");
  SBK_LAUNCH(ctc_init_kernel, dim3(B), dim3(256), (size_t)T * sizeof(float), st, (const float*)xb_log, v.st, v.sg, v.sb,
             v.se, psi_prev, T, beam);
  return launch_status("ctc_init");
}

// psi[n,c] for every hypothesis / token (needs only the CTC state: can run beside the decoder step)
int ctc_psi_step(const float* P, const float* state, const int32_t* last_tok, const int32_t* enc_len, float* psi, int B,
                 int T, int V, int beam, int prefix_len, int blank, int eos, hipStream_t st, const int32_t* win, int window, int ldp) {
  CtcStepArgs a{last_tok, enc_len, B, T, V, beam, prefix_len, blank, eos, 0.0f, 0, 0, 0.0f, 0.0f, g_step_ptr, 0};
  a.win = win;
  a.window = window;
  a.ldp = ldp < V ? V : ldp;
  const StateView v = view(const_cast<float*>(state), B, beam, T);
  ProfScope prof("ctc_score_step", 2.0 * B * beam * (double)T * V, 4.0 * B * (double)T * V + 4.0 * B * beam * V, st);
  constexpr int tpt = 1;
  dim3 grid(cdiv(V, 256 * tpt), B, beam_pitch(beam) / 16), block(256);  // z: tiles of 16 beams (P is re-read per tile)
  const size_t lds = ((size_t)T * 16 + (size_t)((T + kSeg - 1) / kSeg) * 16) * sizeof(float);
  if (lds > 160 * 1024) return fail(SBK_EINVAL, "ctc_psi_step: T=%d frames need %zu B of LDS (max 160 KiB)", T, lds);
  // utterances beyond ~39 s (T' > 990) need more than the default 64 KiB dynamic-LDS window
#define SBK_CTC_LAUNCH_NT(NB, TP, NT)                                                                                   \
  do {                                                                                                                  \
    if (lds > 64 * 1024 && SBK_ALLOW_DYN_LDS((ctc_score_step_kernel<NB, TP, NT>), lds) != hipSuccess)                    \
      return fail(SBK_EINVAL, "ctc_psi_step: cannot raise the LDS window to %zu B", lds);                               \
    SBK_LAUNCH((ctc_score_step_kernel<NB, TP, NT>), grid, block, lds, st, a, P, (const BF*)v.st, (const float*)v.sg,    \
               (const int*)v.se, psi);                                                                                  \
  } while (0)
#define SBK_CTC_LAUNCH_ONE(NB, TP)                                                                                      \
  do {                                                                                                                  \
    if (g_nt_mask & 4)                                                                                                  \
      SBK_CTC_LAUNCH_NT(NB, TP, true);                                                                                  \
    else                                                                                                                \
      SBK_CTC_LAUNCH_NT(NB, TP, false);                                                                                 \
  } while (0)
#define SBK_CTC_LAUNCH(NB) SBK_CTC_LAUNCH_ONE(NB, 1)
  if (beam == 1) {
    SBK_CTC_LAUNCH(1);
  } else if (beam <= 4) {
    SBK_CTC_LAUNCH(4);
  } else if (beam <= 10) {
    SBK_CTC_LAUNCH(10);
  } else {
    SBK_CTC_LAUNCH(16);
  }
#undef SBK_CTC_LAUNCH
#undef SBK_CTC_LAUNCH_ONE
#undef SBK_CTC_LAUNCH_NT
  int rc = launch_status("ctc_score_step");
  if (rc) return rc;
  SBK_LAUNCH(ctc_same_token_kernel, dim3(B * beam), dim3(64), 0, st, a, P, (const float*)v.sb, (const int*)v.se, psi);
  return launch_status("ctc_same_token");
}

int ctc_combine(const float* am, const float* am_max, const float* psi, const float* psi_prev, float* comb, int n_bh,
                int V, int blank, int eos, float weight, int eos_floor, int use_thr, float thr, float minus_inf,
                const float* extra, hipStream_t st, const int32_t* utt_min, int beam, int step) {
  CtcStepArgs a{nullptr, nullptr, 0, 0, V, 0, 0, blank, eos, weight, eos_floor, use_thr, thr, minus_inf, g_step_ptr,
                g_step_min_steps, utt_min, beam > 0 ? beam : 1, step};
  SBK_LAUNCH(ctc_combine_kernel, dim3(cdiv(V, 256), n_bh), dim3(256), 0, st, a, am, am_max, psi, psi_prev, comb, extra);
  return launch_status("ctc_combine");
}

int ctc_advance(const float* P, const float* state_old, const float* psi, const int32_t* parent, const int32_t* token,
                const int32_t* parent_last_tok, float* state_new, float* psi_prev_new, int n_bh, int T, int V, int beam,
                int prefix_len, int blank, hipStream_t st, const int32_t* win, int window, int ldp) {
  const StateView vo = view(const_cast<float*>(state_old), n_bh / beam, beam, T);
  const StateView vn = view(state_new, n_bh / beam, beam, T);
  CtcAdvArgs a{P, vo.st, psi, parent, token, parent_last_tok, vn.st, vn.sg, vn.sb, vn.se, psi_prev_new, n_bh, T, V, beam,
               prefix_len, blank, g_step_ptr, win, window, ldp < V ? V : ldp};
  const size_t lds = (size_t)7 * T * sizeof(float);
  if (lds > 64 * 1024) return fail(SBK_EINVAL, "ctc_advance: T=%d too long for the LDS window", T);
  ProfScope prof("ctc_advance", 14.0 * n_bh * T, 64.0 * n_bh * T, st);
  SBK_LAUNCH(ctc_advance_kernel, dim3(n_bh), dim3(64), lds, st, a);
  return launch_status("ctc_advance");
}

int am_only(const float* am, float* comb, int n_bh, int V, int eos, int eos_floor, int use_thr, float thr,
            float minus_inf, const float* am_max, const float* extra, hipStream_t st, const int32_t* utt_min, int beam,
            int step) {
  const int32_t* sp = g_step_ptr;  // locals: launch arguments must not name the thread_locals themselves
  const int min_steps = g_step_min_steps;
  SBK_LAUNCH(am_only_kernel, dim3(cdiv(V, 256), n_bh), dim3(256), 0, st, am, comb, V, eos, eos_floor, use_thr, thr,
             minus_inf, am_max, extra, sp, min_steps, utt_min, beam > 0 ? beam : 1, step);
  return launch_status("am_only");
}

int ctc_partial_combine(float* comb, float* thr, const float* psi, const float* psi_prev, int n_bh, int V, int k, int blank,
                        int eos, float weight, float minus_inf, hipStream_t st) {
  const int kk = k < 1 ? 1 : (k > V ? V : k);  // scorer.py:1290-1291
  SBK_LAUNCH(row_kth_largest_kernel, dim3(n_bh), dim3(256), 0, st, (const float*)comb, thr, V, kk);
  SBK_LAUNCH(ctc_partial_combine_kernel, dim3(cdiv(V, 256), n_bh), dim3(256), 0, st, comb, (const float*)thr, psi, psi_prev,
             V, blank, eos, weight, minus_inf);
  return launch_status("ctc_partial_combine");
}

int row_max(const float* x, float* out, int rows, int V, hipStream_t st) {
  SBK_LAUNCH(row_max_kernel, dim3(rows), dim3(256), 0, st, x, out, V);
  return launch_status("row_max");
}

}  // namespace sbk


// Measurement helper (tools/microbench.py): `iters` back-to-back ctc_psi_step launches on a freshly
// initialised state; mean microseconds per launch.  `work` >= ctc_state_floats + B*T + n_bh floats.
// ---- the CTC prefix scorer as a per-step API (decoders/scorer.py:108-255 CTCScorer.reset_mem / score / permute_mem, i.e.
// decoders/ctc.py:79-295 CTCPrefixScore.forward_step / permute_mem): what a searcher written against the reference's
// ScorerBuilder calls once per decoding step.  The fused search (sbk_beam_search_f32) drives the same kernels itself.
namespace {
__global__ void __launch_bounds__(256) ctc_score_delta_kernel(const float* __restrict__ psi, const float* __restrict__ psi_prev,
                                                              float* __restrict__ out, int V) {
  const int n = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c < V) out[(size_t)n * V + c] = sbk::sub_rn(psi[(size_t)n * V + c], psi_prev[n]);  // ctc.py:262 (psi - psi_prev)
}
struct ScorerWs {
  float* phi[2];
  float* psi;
  float* psi_prev[2];
  float* xb;
};
size_t scorer_ws_floats(int B, int T, int V, int beam) {
  auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
  return 2 * up(sbk::ctc_state_floats(B, beam, T)) + up((size_t)B * beam * V) + 2 * up((size_t)B * beam) + up((size_t)B * T);
}
ScorerWs carve_scorer(void* ws, int B, int T, int V, int beam) {
  auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
  float* p = static_cast<float*>(ws);
  ScorerWs w;
  w.phi[0] = p, p += up(sbk::ctc_state_floats(B, beam, T));
  w.phi[1] = p, p += up(sbk::ctc_state_floats(B, beam, T));
  w.psi = p, p += up((size_t)B * beam * V);
  w.psi_prev[0] = p, p += up((size_t)B * beam);
  w.psi_prev[1] = p, p += up((size_t)B * beam);
  w.xb = p;
  return w;
}
}  // namespace

extern "C" size_t sbk_ctc_scorer_workspace_bytes(int B, int T, int V, int beam) {
  if (B <= 0 || T <= 0 || V <= 0 || beam <= 0) return 0;
  return scorer_ws_floats(B, T, V, beam) * sizeof(float);
}

extern "C" int sbk_ctc_scorer_reset_f32(float* x, const int32_t* enc_len, void* ws, size_t ws_bytes, int B, int T, int V, int beam,
                                        int blank, sbk_stream_t stream) {
  if (B == 0) return 0;
  SBK_REQUIRE(x && enc_len && ws && B > 0 && T > 0 && V > 0 && beam > 0 && blank >= 0 && blank < V, "ctc_scorer_reset: bad arguments");
  SBK_REQUIRE(ws_bytes >= sbk_ctc_scorer_workspace_bytes(B, T, V, beam) && sbk::aligned16(ws) && sbk::aligned16(x),
              "ctc_scorer_reset: workspace too small (%zu B) or unaligned", ws_bytes);
  hipStream_t st = sbk::as_stream(stream);
  const ScorerWs w = carve_scorer(ws, B, T, V, beam);
  if (hipMemsetAsync(w.phi[1], 0, sbk::ctc_state_floats(B, beam, T) * sizeof(float), st) != hipSuccess)  // (table padding reads as zero)
    return sbk::fail(1, "ctc_scorer_reset: memset");
  return sbk::ctc_prepare(x, w.xb, enc_len, w.phi[0], w.psi_prev[0], B, T, V, beam, blank, st, V);
}

extern "C" int sbk_ctc_scorer_score_f32(const float* x, const int32_t* enc_len, void* ws, size_t ws_bytes, const int32_t* inp_tokens,
                                        int step, const int32_t* attn_window, int ctc_window_size, float* scores, int B, int T, int V,
                                        int beam, int blank, int eos, sbk_stream_t stream) {
  if (B == 0) return 0;
  SBK_REQUIRE(x && enc_len && ws && inp_tokens && scores && B > 0 && T > 0 && V > 0 && beam > 0 && step >= 0, "ctc_scorer_score: bad arguments");
  SBK_REQUIRE(ws_bytes >= sbk_ctc_scorer_workspace_bytes(B, T, V, beam), "ctc_scorer_score: workspace too small");
  SBK_REQUIRE(ctc_window_size >= 0 && (ctc_window_size == 0 || attn_window), "ctc_scorer_score: a window needs the {min, max} attention peaks");
  hipStream_t st = sbk::as_stream(stream);
  const ScorerWs w = carve_scorer(ws, B, T, V, beam);
  const int cur = step & 1;
  const int32_t* win = ctc_window_size > 0 ? attn_window : nullptr;
  int rc = sbk::ctc_psi_step(x, w.phi[cur], inp_tokens, enc_len, w.psi, B, T, V, beam, step, blank, eos, st, win, ctc_window_size, V);
  if (rc) return rc;
  SBK_LAUNCH(ctc_score_delta_kernel, dim3(sbk::cdiv(V, 256), B * beam), dim3(256), 0, st, (const float*)w.psi,
             (const float*)w.psi_prev[cur], scores, V);
  return sbk::launch_status("ctc_score_delta");
}

extern "C" int sbk_ctc_scorer_permute_f32(const float* x, void* ws, size_t ws_bytes, const int32_t* parent, const int32_t* token,
                                          const int32_t* parent_last_tok, int step, const int32_t* attn_window, int ctc_window_size,
                                          int B, int T, int V, int beam, int blank, sbk_stream_t stream) {
  if (B == 0) return 0;
  SBK_REQUIRE(x && ws && parent && token && parent_last_tok && B > 0 && T > 0 && V > 0 && beam > 0 && step >= 0, "ctc_scorer_permute: bad arguments");
  SBK_REQUIRE(ws_bytes >= sbk_ctc_scorer_workspace_bytes(B, T, V, beam), "ctc_scorer_permute: workspace too small");
  SBK_REQUIRE(ctc_window_size >= 0 && (ctc_window_size == 0 || attn_window), "ctc_scorer_permute: a window needs the {min, max} attention peaks");
  const ScorerWs w = carve_scorer(ws, B, T, V, beam);
  const int cur = step & 1;
  return sbk::ctc_advance(x, w.phi[cur], w.psi, parent, token, parent_last_tok, w.phi[cur ^ 1], w.psi_prev[cur ^ 1], B * beam, T, V,
                          beam, step, blank, sbk::as_stream(stream), ctc_window_size > 0 ? attn_window : nullptr, ctc_window_size, V);
}

extern "C" int sbk_prof_ctc_psi_repeat_f32(float* P_logsoftmax, const int32_t* enc_len, const int32_t* last_tok,
                                           float* psi, float* work, int B, int T, int V, int beam, int prefix_len,
                                           int iters, float* us_per_launch, sbk_stream_t stream) {
  SBK_REQUIRE(P_logsoftmax && enc_len && last_tok && psi && work && us_per_launch && iters > 0, "psi_repeat: bad arguments");
  hipStream_t st = sbk::as_stream(stream);
  float* state = work;
  float* xb = work + sbk::ctc_state_floats(B, beam, T);
  float* psi_prev = xb + (size_t)B * T;
  int rc = sbk::ctc_prepare(P_logsoftmax, xb, enc_len, state, psi_prev, B, T, V, beam, 0, st, V);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return sbk::fail(1, "event create");
  for (int i = 0; i < 2 && !rc; ++i)
    rc = sbk::ctc_psi_step(P_logsoftmax, state, last_tok, enc_len, psi, B, T, V, beam, prefix_len, 0, 2, st, nullptr, 0, V);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters && !rc; ++i)
    rc = sbk::ctc_psi_step(P_logsoftmax, state, last_tok, enc_len, psi, B, T, V, beam, prefix_len, 0, 2, st, nullptr, 0, V);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = ms * 1000.0f / iters;
  return rc;
}
