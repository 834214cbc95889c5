// S2STransformerBeamSearcher / S2STransformerGreedySearcher on the device.
//
// Replaces decoders/seq2seq.py:711-1749 (S2SBeamSearcher.forward and helpers),
// :1853-1934 (S2STransformerBeamSearcher), :176-367 (greedy), scorer.py:1221-1315
// (ScorerBuilder with the full CTC and / or TransformerLM scorers, scorer.py:413-577) and
// TransformerASR.decode (TransformerASR.py:426-473).  The whole search runs inside ONE C-ABI call: the
// host loop only enqueues kernels; beam bookkeeping (length-normalised top-k over
// beam*V, predecessor gathers, EOS harvesting, finished-hypothesis lists) lives in
// device memory, so there is no per-step host synchronisation.  The reference's
// stop rule ("every utterance has beam_size finished hypotheses") is polled every
// `check_every` steps through a 4-byte asynchronous copy that is read back two polls
// later (AsyncPoll below: the stream is never drained); running past the stop point
// cannot change the result because full lists accept no further hypotheses.
#include <limits.h>

#include "common.h"
#include "internal.h"

namespace {
// Stop-rule polling without a pipeline bubble.  Every `check_every` steps the search enqueues a 4-byte device->host
// copy of its "finished" counter plus an event and goes on enqueuing steps; it reads a copy back once its event has
// completed.  The host never waits for the NEWEST copy (that would drain the stream: a bubble of one launch chain per
// poll, round 2's behaviour) -- only for the one issued two polls earlier, which bounds how far the host runs ahead of
// the device (otherwise it would enqueue every step before the first copy lands and an early stop would save
// nothing).  A search therefore runs at most ~3 poll intervals past its stop point; running past it cannot change the
// result (finished lists accept nothing more / outputs are EOS-latched).  One ring per host thread.
struct AsyncPoll {
  static constexpr int kRing = 4;
  int32_t* host = nullptr;  // pinned
  hipEvent_t ev[kRing] = {};
  int issued = 0;
  bool ok = false;
  bool begin() {
    if (!ok) {
      if (hipHostMalloc(reinterpret_cast<void**>(&host), kRing * sizeof(int32_t), 0) != hipSuccess) return false;
      for (int i = 0; i < kRing; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
      ok = true;
    }
    for (int j = issued > kRing ? issued - kRing : 0; j < issued; ++j) (void)hipEventSynchronize(ev[j % kRing]);  // a previous search's copies
    issued = 0;
    return true;
  }
  // 1: the counter has reached `target`; 0: go on; -1: error
  int poll(const int32_t* counter_dev, int target, hipStream_t st) {
    const int k = issued % kRing;
    if (hipMemcpyAsync(host + k, counter_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipEventRecord(ev[k], st) != hipSuccess)
      return -1;
    ++issued;
    if (issued >= 3) {  // bounded run-ahead
      const int j = (issued - 3) % kRing;
      if (hipEventSynchronize(ev[j]) != hipSuccess) return -1;
      if (host[j] >= target) return 1;
    }
    for (int q = issued >= 2 ? issued - 2 : 0; q < issued; ++q)
      if (hipEventQuery(ev[q % kRing]) == hipSuccess && host[q % kRing] >= target) return 1;
    return 0;
  }
};
thread_local AsyncPoll g_poll;
}  // namespace

namespace sbk {
// decoder.hip / ctc_prefix.hip
int embed_pos(const int32_t* tok, const float* emb, const float* pe_row, float* x, int n, int d, float scale,
              hipStream_t st);
int self_attn_step(const float* qkv, float* kcache, float* vcache, const int32_t* kv_slot, float* out, int n, int d,
                   int H, int step, int nslot, int Lmax, hipStream_t st, const int32_t* key_tok = nullptr,
                   int key_stride = 0, int key_shift = 0, int key_first = 0, int pad_idx = 0, int group = 1);
int cross_attn_step(const float* q, const float* kv, const int32_t* enc_len, float* out, float* part, int B, int T,
                    int d, int H, int beam, hipStream_t st);
size_t cross_attn_partial_floats(int B, int T, int H, int Dh, int beam);
int log_softmax_rows(const float* x, float* out, int rows, int V, float temperature, float weight, hipStream_t st,
                     const float* bias = nullptr, const float* bias2 = nullptr, int ld = 0);
int ctc_prepare(float* x, float* xb_log, const int32_t* enc_len, float* state, float* psi_prev, int B, int T, int V,
                int beam, int blank, hipStream_t st, int ldp = 0);
size_t ctc_state_floats(int B, int beam, int T);
int ctc_psi_step(const float* P, const float* state, const int32_t* last_tok, const int32_t* enc_len, float* psi, int B,
                 int T, int V, int beam, int prefix_len, int blank, int eos, hipStream_t st, const int32_t* win = nullptr,
                 int window = 0, int ldp = 0);
int cross_attn_avg_probs(const float* q, const float* kv, const int32_t* enc_len, float* out, int n, int T, int d, int H,
                         int beam, hipStream_t st);
int ctc_combine(const float* am, const float* am_max, const float* psi, const float* psi_prev, float* comb, int n_bh,
                int V, int blank, int eos, float weight, int eos_floor, int use_thr, float thr, float minus_inf,
                const float* extra, hipStream_t st, const int32_t* utt_min = nullptr, int beam = 1, int step = 0);
int ctc_advance(const float* x, const float* phi_old, const float* psi, const int32_t* parent, const int32_t* token,
                const int32_t* parent_last_tok, float* phi_new, float* psi_prev_new, int n_bh, int T, int V, int beam,
                int prefix_len, int blank, hipStream_t st, const int32_t* win = nullptr, int window = 0, int ldp = 0);
int am_only(const float* am, float* comb, int n_bh, int V, int eos, int eos_floor, int use_thr, float thr,
            float minus_inf, const float* am_max, const float* extra, hipStream_t st, const int32_t* utt_min = nullptr,
            int beam = 1, int step = 0);
int row_max(const float* x, float* out, int rows, int V, hipStream_t st);
int ctc_partial_combine(float* comb, float* thr, const float* psi, const float* psi_prev, int n_bh, int V, int k, int blank,
                        int eos, float weight, float minus_inf, hipStream_t st);
int g_score_fused = 1;  // tuning knob (key 40): 0 = log_softmax / row_max / combine / top-k stage 1 as separate launches
}  // namespace sbk

namespace {

// The CTC scorer's work of a step (psi of every candidate, then the survivors' new state) depends
// only on the beam bookkeeping, not on the decoder step: it runs on a helper stream beside the
// decoder GEMMs and is joined just before the scores are combined.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
SideStream* side_stream() {
  static thread_local SideStream ss;  // one per host thread (= per concurrent search)
  if (!ss.s) {
    if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) != hipSuccess)
      return nullptr;
  }
  return &ss;
}

constexpr int kMaxBeam = 16;        // per-thread list length of the two-stage top-k
constexpr int kMaxBeamLarge = 128;  // radix-select top-k (beam_topk_large_kernel) above that

// ---------------------------------------------------------------- top-k over beam*V per utterance
struct Cand {
  float v;
  int i;
};
__device__ __forceinline__ bool better(float v, int i, float v2, int i2) { return v > v2 || (v == v2 && i < i2); }

// Block-wide selection of the `k` best (value, index) pairs among the candidates a loader yields.
// Every thread keeps a sorted top-16 list of its strided share (registers); the lists go to LDS, every
// wave picks the k best of its 64 lists with k shuffle-only arg-max rounds, and the first wave merges
// the 4 * k wave winners the same way: one barrier in all, winners in descending order
// (ties: lower index first).
template <typename Load>
__device__ __forceinline__ void block_topk(int total, int k, float* out_val, int32_t* out_idx, Load load) {
  // per-thread sorted lists, transposed so that a wave's accesses to one list position are bank-conflict free
  __shared__ float lv[kMaxBeam][256];
  __shared__ int li[kMaxBeam][256];
  __shared__ float wv[4][kMaxBeam];  // each wave's k best, in order
  __shared__ int wi[4][kMaxBeam];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float vals[kMaxBeam];
  int ids[kMaxBeam];
#pragma unroll
  for (int s = 0; s < kMaxBeam; ++s) {
    vals[s] = -INFINITY;
    ids[s] = INT_MAX;
  }
  for (int e = tid; e < total; e += 256) {
    float vn;
    int id;
    load(e, vn, id);
    if (vn != vn) continue;
    if (better(vn, id, vals[kMaxBeam - 1], ids[kMaxBeam - 1])) {
      float cv = vn;
      int ci = id;
#pragma unroll
      for (int s = 0; s < kMaxBeam; ++s) {
        if (better(cv, ci, vals[s], ids[s])) {
          const float tv = vals[s];
          const int ti = ids[s];
          vals[s] = cv;
          ids[s] = ci;
          cv = tv;
          ci = ti;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < kMaxBeam; ++s) {
    lv[s][tid] = vals[s];
    li[s][tid] = ids[s];
  }
  // every wave selects the k best of its own 64 lists: k wave-wide arg-max rounds, no barrier (a lane reads only
  // what it wrote); the top k of the workgroup are among the 4 * k wave winners
  int hp = 0;
  for (int r = 0; r < k; ++r) {
    float v = hp < kMaxBeam ? lv[hp][tid] : -INFINITY;
    int i = hp < kMaxBeam ? li[hp][tid] : INT_MAX;
    const int mine = i;
    sbk::wave_argmax(v, i);
    if (lane == 0) {
      wv[wave][r] = v;
      wi[wave][r] = i;
    }
    if (hp < kMaxBeam && mine == i && i != INT_MAX) ++hp;  // candidate ids are unique: exactly one lane advances
  }
  __syncthreads();
  if (wave == 0) {  // merge the 4 * k <= 64 wave winners: one per lane, k arg-max rounds with removal
    float v = -INFINITY;
    int i = INT_MAX;
    if (lane < 4 * k) {
      v = wv[lane / k][lane % k];
      i = wi[lane / k][lane % k];
    }
    for (int r = 0; r < k; ++r) {
      float bv = v;
      int bi = i;
      sbk::wave_argmax(bv, bi);
      if (lane == 0) {
        out_val[r] = bv;
        out_idx[r] = bi;
      }
      if (i == bi && bi != INT_MAX) {  // the winner leaves the pool
        v = -INFINITY;
        i = INT_MAX;
      }
    }
  }
}

constexpr int kTopkChunks = 16;  // stage-1 workgroups per utterance

// stage 1: grid (chunks, B); each workgroup reduces one contiguous slice of the beam*V candidates
__global__ void __launch_bounds__(256) beam_topk_stage1_kernel(const float* __restrict__ comb,
                                                               const float* __restrict__ seq, float* __restrict__ pval,
                                                               int32_t* __restrict__ pidx, int V, int beam, float norm,
                                                               const int32_t* __restrict__ step_ptr, int step,
                                                               const int32_t* __restrict__ utt_max) {
  if (step_ptr && norm > 0.0f) norm = (float)(step_ptr[0] + 1);  // length normalisation by the device-side step
  if (step_ptr) step = step_ptr[0];
  const int b = blockIdx.y, ch = blockIdx.x;
  if (utt_max && step >= utt_max[b]) return;  // this utterance's search has ended: its candidates stay frozen
  const int total = beam * V;
  const int len = (total + kTopkChunks - 1) / kTopkChunks;
  const int e0 = ch * len, n = max(0, min(len, total - e0));
  const float* cb = comb + (size_t)b * total;
  const float* sq = seq + b * beam;
  block_topk(n, beam, pval + ((size_t)b * kTopkChunks + ch) * kMaxBeam, pidx + ((size_t)b * kTopkChunks + ch) * kMaxBeam,
             [&](int e, float& v, int& id) {
               id = e0 + e;
               v = sbk::score_cand(sq[id / V], cb[id], norm);
             });
}

// stage 2: grid (B); merge the winners of the first `nchunks` lists (stage 1: kTopkChunks slices of beam*V; the fused
// scoring kernel below: one list per hypothesis row)
__global__ void __launch_bounds__(256) beam_topk_stage2_kernel(const float* __restrict__ pval,
                                                               const int32_t* __restrict__ pidx,
                                                               float* __restrict__ out_val,
                                                               int32_t* __restrict__ out_idx, int beam,
                                                               const int32_t* __restrict__ step_ptr, int step,
                                                               const int32_t* __restrict__ utt_max, int nchunks) {
  const int b = blockIdx.x;
  if (step_ptr) step = step_ptr[0];
  if (utt_max && step >= utt_max[b]) return;
  const float* pv = pval + (size_t)b * kTopkChunks * kMaxBeam;
  const int32_t* pi = pidx + (size_t)b * kTopkChunks * kMaxBeam;
  block_topk(nchunks * kMaxBeam, beam, out_val + b * beam, out_idx + b * beam, [&](int e, float& v, int& id) {
    const bool ok = (e % kMaxBeam) < beam;
    v = ok ? pv[e] : -INFINITY;
    id = ok ? pi[e] : INT_MAX;
  });
  __syncthreads();
  if (threadIdx.x < beam && out_idx[b * beam + threadIdx.x] == INT_MAX) out_idx[b * beam + threadIdx.x] = 0;
}

// ---------------------------------------------------------------- one pass over a hypothesis' vocabulary row
// The scoring of a step as ONE kernel per hypothesis row instead of four launches and five sweeps over [n_bh, V]
// (log_softmax_row -> [row_max] -> ctc_combine / am_only -> beam_topk_stage1):  the logits row lives in registers
// (V <= 256 * NPT), so the log-softmax, the eos rules (seq2seq.py:995-1017), the scorer combination
// (scorer.py:1248-1253) and the candidate value seq_score + combined [/ length] (seq2seq.py:1225-1240) are applied in
// place and the row's `beam` best candidates are selected from the registers: the top `beam` of beam * V candidates are
// among the per-row top `beam` lists, which stage 2 merges (one list per row instead of one per slice).  Every value is
// computed by the expressions of the kernels it replaces, in the same order (the thread <-> column mapping and the
// reduction trees of log_softmax_row_kernel included), so candidates and scores are bit-identical to the four-launch
// path (tests/test_model_parity.py::test_fused_scoring_equals_separate_kernels); only `am` (the pre-scorer log-probs
// beam_update records per token) is still written out.  HBM traffic per step: logits + psi read, am written (12 B per
// candidate) against 24 B + the top-k sweep.
struct ScoreArgs {
  const float* logits;   // [n_bh, V] seq_lin output
  const float* bias;     // optional additive masks on the logits ([V]; Whisper's suppress lists)
  const float* bias2;
  float* am;             // [n_bh, V] out: w * log_softmax(logits / temp)
  const float* psi;      // [n_bh, V] CTC prefix scores of the step, NULL = no CTC scorer
  const float* psi_prev; // [n_bh]
  const float* extra;    // [n_bh, V] weighted log-probs of the full scorers listed before "ctc" (LM), or NULL
  const float* seq;      // [n_bh] running hypothesis scores
  float* pval;           // [B][kTopkChunks][kMaxBeam] per-row winners for stage 2
  int32_t* pidx;
  const int32_t* step_ptr;
  const int32_t* utt_min;
  const int32_t* utt_max;
  int V, beam, step, min_steps, eos_floor, use_thr, eos, blank;
  float inv_temp, w, ctc_weight, thr, minus_inf, norm;
};

template <int NPT>
__global__ void __launch_bounds__(256) score_topk_row_kernel(ScoreArgs a) {
  __shared__ float red[4];
  __shared__ float wv[4][kMaxBeam];
  __shared__ int wi[4][kMaxBeam];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = a.V, b = n / a.beam, j = n % a.beam;
  const int step = a.step_ptr ? a.step_ptr[0] : a.step;
  if (a.utt_max && step >= a.utt_max[b]) return;  // this utterance's search has ended: nothing of the row is read again
  float norm = a.norm;
  if (a.step_ptr && norm > 0.0f) norm = (float)(step + 1);
  const size_t ro = (size_t)n * V;
  // --- log-softmax (log_softmax_row_kernel: same column mapping, same reduction order)
  // every global operand of the row is requested up front, unconditionally, on a clamped column (a load under a lane mask is
  // a branch and a full wait each: NPT round trips in a row); columns past V are discarded below
  float x[NPT], ps[NPT], ex[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) x[i] = a.logits[ro + min(tid + 256 * i, V - 1)];
  if (a.psi) {  // (uniform)
#pragma unroll
    for (int i = 0; i < NPT; ++i) ps[i] = a.psi[ro + min(tid + 256 * i, V - 1)];
  }
  if (a.extra) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) ex[i] = a.extra[ro + min(tid + 256 * i, V - 1)];
  }
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = tid + 256 * i, cl = min(c, V - 1);
    const float v = sbk::ls_logit(x[i], a.bias ? a.bias[cl] : 0.0f, a.bias2 ? a.bias2[cl] : 0.0f, a.inv_temp);
    x[i] = c < V ? v : -INFINITY;
    if (c < V) m = fmaxf(m, v);
  }
  m = sbk::wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NPT; ++i)
    if (tid + 256 * i < V) s = sbk::add_rn(s, expf(sbk::sub_rn(x[i], m)));
  s = sbk::wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float lse = sbk::add_rn(m, logf((red[0] + red[1]) + (red[2] + red[3])));
  __syncthreads();
  float am_max = -INFINITY;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = tid + 256 * i;
    if (c < V) {
      x[i] = sbk::ls_out(x[i], lse, a.w);
      a.am[ro + c] = x[i];
      am_max = fmaxf(am_max, x[i]);
    }
  }
  if (a.use_thr) {  // row_max_kernel
    am_max = sbk::wave_max(am_max);
    if (lane == 0) red[wave] = am_max;
    __syncthreads();
    am_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  // --- eos rules, scorers, candidate value (am_only_kernel / ctc_combine_kernel, the loader of beam_topk_stage1_kernel)
  const float sq = a.seq[n];
  const float pp = a.psi ? a.psi_prev[n] : 0.0f;
  unsigned dead = 0;  // bit i: entry i is no candidate (past V, NaN, or already selected)
  float bv = -INFINITY;
  int bi = INT_MAX;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = tid + 256 * i;
    if (c >= V) {
      dead |= 1u << i;
      continue;
    }
    float v = x[i];
    if (c == a.eos) {
      const bool floor = a.utt_min ? step < a.utt_min[b] : (a.step_ptr ? step < a.min_steps : a.eos_floor != 0);
      if (floor) v = a.minus_inf;
      if (a.use_thr && !(v > a.thr * am_max)) v = a.minus_inf;
    }
    if (a.extra) v = sbk::add_rn(v, ex[i]);
    if (a.psi) {
      if (c == a.blank) v = sbk::kCtcNeg;
      v = sbk::score_ctc(v, ps[i], pp, a.ctc_weight);
    }
    v = sbk::score_cand(sq, v, norm);
    x[i] = v;
    if (v != v) {
      dead |= 1u << i;
      continue;
    }
    const int id = j * V + c;
    if (better(v, id, bv, bi)) {
      bv = v;
      bi = id;
    }
  }
  // --- the row's `beam` best: every wave selects the beam best of its lanes (beam arg-max rounds over the lanes' current
  // best; only the winning lane rescans its registers), the first wave merges the 4 * beam wave winners
  const int k = a.beam;
  for (int r = 0; r < k; ++r) {
    float v = bv;
    int i2 = bi;
    sbk::wave_argmax(v, i2);
    if (lane == 0) {
      wv[wave][r] = v;
      wi[wave][r] = i2;
    }
    if (bi == i2 && i2 != INT_MAX) {  // candidate ids are unique: exactly one lane of the wave
      dead |= 1u << ((bi - j * V - tid) >> 8);
      bv = -INFINITY;
      bi = INT_MAX;
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int id = j * V + tid + 256 * i;
        if (!((dead >> i) & 1u) && better(x[i], id, bv, bi)) {
          bv = x[i];
          bi = id;
        }
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float v = -INFINITY;
    int i2 = INT_MAX;
    if (lane < 4 * k) {
      v = wv[lane / k][lane % k];
      i2 = wi[lane / k][lane % k];
    }
    float* ov_ = a.pval + ((size_t)b * kTopkChunks + j) * kMaxBeam;
    int32_t* oi_ = a.pidx + ((size_t)b * kTopkChunks + j) * kMaxBeam;
    for (int r = 0; r < k; ++r) {
      float wbv = v;
      int wbi = i2;
      sbk::wave_argmax(wbv, wbi);
      if (lane == 0) {
        ov_[r] = wbv;
        oi_[r] = wbi;
      }
      if (i2 == wbi && wbi != INT_MAX) {
        v = -INFINITY;
        i2 = INT_MAX;
      }
    }
  }
}

// Beams wider than the per-thread lists (the recipe's test_search uses beam 66): exact radix select.
// One workgroup of 1024 threads per utterance: four 8-bit passes over the order-preserving integer
// image of the candidate scores find the beam-th largest key, a fifth pass collects the winners
// (ties on the threshold: lowest candidate index first, in index order), and a bitonic sort in LDS
// emits them in descending order.
__device__ __forceinline__ unsigned order_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(1024) beam_topk_large_kernel(const float* __restrict__ comb,
                                                               const float* __restrict__ seq,
                                                               float* __restrict__ out_val,
                                                               int32_t* __restrict__ out_idx, int V, int beam,
                                                               float norm, const int32_t* __restrict__ step_ptr,
                                                               int step, const int32_t* __restrict__ utt_max) {
  if (step_ptr && norm > 0.0f) norm = (float)(step_ptr[0] + 1);
  if (step_ptr) step = step_ptr[0];
  if (utt_max && step >= utt_max[blockIdx.x]) return;  // uniform per workgroup
  __shared__ int hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_remaining, s_count, s_eq_base;
  __shared__ int scan[1024];
  __shared__ float wv[kMaxBeamLarge];
  __shared__ int wi[kMaxBeamLarge];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int total = beam * V;
  const float* cb = comb + (size_t)b * total;
  const float* sq = seq + b * beam;
  auto value = [&](int e) {
    const float x = sq[e / V] + cb[e];
    const float v = norm > 0.0f ? x / norm : x;
    return v != v ? -INFINITY : v;  // NaN never wins (the small-beam path skips them as well)
  };
  if (tid == 0) {
    s_prefix = 0u;
    s_remaining = beam;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int e = tid; e < total; e += 1024) {
      const unsigned k = order_key(value(e));
      if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = s_remaining, d = 255;
      for (; d > 0; --d) {
        if (hist[d] >= rem) break;
        rem -= hist[d];
      }
      s_remaining = rem;  // how many keys with this digit (and prefix) are still needed
      s_prefix = (prefix << 8) | (unsigned)d;
    }
    __syncthreads();
  }
  const unsigned thr = s_prefix;  // key of the beam-th largest candidate
  const int need_eq = s_remaining;
  if (tid == 0) {
    s_count = 0;
    s_eq_base = 0;
  }
  for (int i = tid; i < kMaxBeamLarge; i += 1024) {
    wv[i] = -INFINITY;
    wi[i] = INT_MAX;
  }
  __syncthreads();
  // winners above the threshold in any order; threshold ties listed (normally exactly one element)
  __shared__ int s_ties;
  if (tid == 0) s_ties = 0;
  __syncthreads();
  for (int e = tid; e < total; e += 1024) {
    const float v = value(e);
    const unsigned k = order_key(v);
    if (k > thr) {
      const int slot = atomicAdd(&s_count, 1);
      if (slot < kMaxBeamLarge) {
        wv[slot] = v;
        wi[slot] = e;
      }
    } else if (k == thr) {
      const int t = atomicAdd(&s_ties, 1);
      if (t < 1024) scan[t] = e;
    }
  }
  __syncthreads();
  const int n_ties = s_ties;
  if (n_ties <= 1024) {  // ties resolve to the lowest candidate indices: rank each tie among the ties
    if (tid < n_ties) {
      const int e = scan[tid];
      int rank = 0;
      for (int q = 0; q < n_ties; ++q) rank += scan[q] < e;
      if (rank < need_eq) {
        const int slot = atomicAdd(&s_count, 1);
        if (slot < kMaxBeamLarge) {
          wv[slot] = value(e);
          wi[slot] = e;
        }
      }
    }
    __syncthreads();
  } else {  // a sea of equal keys (e.g. -inf candidates of dead beams): ordered sweep with a block scan
    __syncthreads();
    for (int e0 = 0; e0 < total; e0 += 1024) {
      const int e = e0 + tid;
      const int is_eq = (e < total && order_key(value(e)) == thr) ? 1 : 0;
      scan[tid] = is_eq;
      __syncthreads();
      for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan of the tie flags
        const int add = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
      }
      const int eq_rank = s_eq_base + scan[tid] - is_eq;
      if (is_eq && eq_rank < need_eq) {
        const int slot = atomicAdd(&s_count, 1);
        if (slot < kMaxBeamLarge) {
          wv[slot] = value(e);
          wi[slot] = e;
        }
      }
      __syncthreads();
      if (tid == 0) s_eq_base += scan[1023];
      __syncthreads();
      if (s_eq_base >= need_eq) break;
    }
    __syncthreads();
  }
  // bitonic sort of the kMaxBeamLarge slots, descending by (value, then lower index)
  for (int size = 2; size <= kMaxBeamLarge; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < kMaxBeamLarge) {
        const int partner = tid ^ stride;
        if (partner > tid) {
          const bool desc = (tid & size) == 0;
          const bool first_better = better(wv[tid], wi[tid], wv[partner], wi[partner]);
          if (desc != first_better) {
            const float tv = wv[tid];
            const int ti = wi[tid];
            wv[tid] = wv[partner];
            wi[tid] = wi[partner];
            wv[partner] = tv;
            wi[partner] = ti;
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid < beam) {
    out_val[b * beam + tid] = wv[tid];
    out_idx[b * beam + tid] = wi[tid] == INT_MAX ? 0 : wi[tid];
  }
}

// ---------------------------------------------------------------- beam bookkeeping
struct BeamState {
  // double-buffered per-hypothesis tables, [n_bh][Lmax]
  int32_t* seq[2];
  float* lp[2];
  int32_t* kv_slot[2];
  int32_t* tokens[2];   // [n_bh] input token of the next / current step
  float* seq_scores;    // [n_bh]
  float* cand_val;      // [n_bh] normalised top-k scores of the current step
  int32_t* cand_idx;    // [n_bh] index into beam*V
  int32_t* parent;      // [n_bh]
  // finished hypotheses, [B][beam]...
  int32_t* fin_count;   // [B]
  int32_t* fin_seq;     // [B][beam][Lmax]
  float* fin_lp;        // [B][beam][Lmax]
  int32_t* fin_len;     // [B][beam]
  float* fin_score;     // [B][beam]
  int32_t* n_full;      // [1] utterances whose finished list is full
  int pos_off;          // decoder position of search step 0 (prompt length - 1; 0 without a prompt): kv_slot is indexed by position
};

__global__ void step_inc_kernel(int32_t* step) { step[0] += 1; }

__global__ void beam_init_kernel(BeamState s, int B, int beam, int bos) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) {
    s.n_full[0] = 0;
    s.n_full[1] = 0;  // longest finished hypothesis (pad width of the reference's top_hyps tensor)
  }
  if (n < B) s.fin_count[n] = 0;
  if (n >= B * beam) return;
  s.tokens[0][n] = bos;
  s.seq_scores[n] = (n % beam == 0) ? 0.0f : -INFINITY;  // seq2seq.py:880-884
}

// prompt positions of the ancestry tables: every hypothesis primed its OWN cache rows with the (per-utterance) prompt
__global__ void kv_prompt_init_kernel(BeamState s, int n, int Lmax, int pos_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * pos_off) return;
  const int h = i / pos_off, p = i % pos_off;
  s.kv_slot[0][(size_t)h * Lmax + p] = h;
  s.kv_slot[1][(size_t)h * Lmax + p] = h;
}
// col[h] = prompt[h / beam][p]
__global__ void prompt_col_kernel(const int32_t* __restrict__ prompt, int32_t* __restrict__ col, int n, int beam, int P, int p) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h < n) col[h] = prompt[(size_t)(h / beam) * P + p];
}
// out[b] = probe[b * beam]  (the beams of an utterance share the prompt)
__global__ void probe_pick_kernel(const float* __restrict__ probe, float* __restrict__ out, int B, int beam) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = probe[(size_t)b * beam];
}

__device__ __forceinline__ void beam_update_body(const BeamState& s, const float* __restrict__ am, int cur, int step, int V,
                                                 int beam, int Lmax, int eos, int length_norm,
                                                 const int32_t* __restrict__ utt_max) {
  __shared__ int h_src[kMaxBeamLarge];
  __shared__ int h_dst[kMaxBeamLarge];
  __shared__ int h_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nxt = cur ^ 1;
  if (utt_max && step >= utt_max[b]) {
    // grouped search: this utterance ran its own number of steps already.  Its hypotheses, scores and finished
    // list stay as they were; only the double-buffered tables are carried over to the other buffer.
    // (The rows keep going through the decoder step with the others -- their results are ignored -- so the
    // ancestry table must stay addressable: position `step` points at the row's own slot.)
    for (int idx = tid; idx < beam * (step + 1); idx += 256) {
      const int j = idx / (step + 1), p = idx % (step + 1);
      const size_t o = ((size_t)b * beam + j) * Lmax + p;
      s.seq[nxt][o] = p < step ? s.seq[cur][o] : 0;
      s.lp[nxt][o] = p < step ? s.lp[cur][o] : 0.0f;
      s.kv_slot[nxt][o + s.pos_off] = p < step ? s.kv_slot[cur][o + s.pos_off] : b * beam + j;
    }
    if (tid < beam) {
      s.tokens[nxt][b * beam + tid] = s.tokens[cur][b * beam + tid];
      s.parent[b * beam + tid] = b * beam + tid;
    }
    return;
  }
  for (int idx = tid; idx < beam * (step + 1); idx += 256) {
    const int j = idx / (step + 1), p = idx % (step + 1);
    const int n = b * beam + j;
    const int cand = s.cand_idx[n];
    const int pred = b * beam + cand / V, tok = cand % V;
    const size_t o = (size_t)n * Lmax + p, po = (size_t)pred * Lmax + p;
    if (p < step) {
      s.seq[nxt][o] = s.seq[cur][po];
      s.lp[nxt][o] = s.lp[cur][po];
      s.kv_slot[nxt][o + s.pos_off] = s.kv_slot[cur][po + s.pos_off];
    } else {
      s.seq[nxt][o] = tok;
      s.lp[nxt][o] = am[(size_t)pred * V + tok];  // pre-scorer AM log-prob (seq2seq.py:1547,1187-1189)
      s.kv_slot[nxt][o + s.pos_off] = pred;       // this step's K/V were written at slot = parent index
      s.tokens[nxt][n] = tok;
      s.parent[n] = pred;
    }
  }
  if (tid == 0) {  // EOS harvest, first come in beam order, at most `beam` per utterance (seq2seq.py:1371-1416)
    int cnt = s.fin_count[b], nh = 0;
    const int before = cnt;
    for (int j = 0; j < beam; ++j) {
      const int n = b * beam + j;
      const int tok = s.cand_idx[n] % V;
      const float score = s.cand_val[n];
      if (tok == eos && cnt < beam) {
        h_src[nh] = j;
        h_dst[nh] = cnt;
        ++nh;
        s.fin_len[b * beam + cnt] = step + 1;
        s.fin_score[b * beam + cnt] = score;
        ++cnt;
      }
      s.seq_scores[n] = tok == eos ? -INFINITY : (length_norm ? score * (float)(step + 1) : score);
    }
    s.fin_count[b] = cnt;
    h_n = nh;
    if (cnt == beam && before < beam) atomicAdd(s.n_full, 1);
  }
  __syncthreads();
  for (int k = 0; k < h_n; ++k) {
    const int n = b * beam + h_src[k];
    const int cand = s.cand_idx[n];
    const int pred = b * beam + cand / V, tok = cand % V;
    const size_t fo = ((size_t)b * beam + h_dst[k]) * Lmax;
    for (int p = tid; p <= step; p += 256) {
      s.fin_seq[fo + p] = p < step ? s.seq[cur][(size_t)pred * Lmax + p] : tok;
      s.fin_lp[fo + p] = p < step ? s.lp[cur][(size_t)pred * Lmax + p] : am[(size_t)pred * V + tok];
    }
  }
}

__global__ void __launch_bounds__(256) beam_update_kernel(BeamState s, const float* __restrict__ am, int cur, int step,
                                                          int V, int beam, int Lmax, int eos, int length_norm,
                                                          const int32_t* __restrict__ step_ptr,
                                                          const int32_t* __restrict__ utt_max) {
  if (step_ptr) step = step_ptr[0];
  beam_update_body(s, am, cur, step, V, beam, Lmax, eos, length_norm, utt_max);
}

// The merge of the per-row winners (beam_topk_stage2_kernel) and the beam bookkeeping (beam_update_kernel) of an
// utterance in one launch: both are one workgroup per utterance, and the bookkeeping reads nothing but its own
// utterance's winners.
__global__ void __launch_bounds__(256) beam_merge_update_kernel(BeamState s, const float* __restrict__ pval,
                                                                const int32_t* __restrict__ pidx,
                                                                const float* __restrict__ am, int cur, int step, int V,
                                                                int beam, int Lmax, int eos, int length_norm,
                                                                const int32_t* __restrict__ step_ptr,
                                                                const int32_t* __restrict__ utt_max, int nchunks) {
  const int b = blockIdx.x;
  if (step_ptr) step = step_ptr[0];
  if (!(utt_max && step >= utt_max[b])) {  // (uniform per workgroup)
    const float* pv = pval + (size_t)b * kTopkChunks * kMaxBeam;
    const int32_t* pi = pidx + (size_t)b * kTopkChunks * kMaxBeam;
    block_topk(nchunks * kMaxBeam, beam, s.cand_val + b * beam, s.cand_idx + b * beam, [&](int e, float& v, int& id) {
      const bool ok = (e % kMaxBeam) < beam;
      v = ok ? pv[e] : -INFINITY;
      id = ok ? pi[e] : INT_MAX;
    });
    __syncthreads();
    if (threadIdx.x < beam && s.cand_idx[b * beam + threadIdx.x] == INT_MAX) s.cand_idx[b * beam + threadIdx.x] = 0;
    __syncthreads();  // the winners (plain global stores of this workgroup) are read back by all of its threads
  }
  beam_update_body(s, am, cur, step, V, beam, Lmax, eos, length_norm, utt_max);
}

// Fill unfinished lists with the alive hypotheses (seq2seq.py:1600-1630) and emit the `topk` best
// finished hypotheses per utterance in descending score order (:1418-1476).  topk = 1: the last token
// is stripped (:1461 + undo_padding) and the row zero padded; topk > 1 (return_topk): rows keep their
// last token like the reference's padded topk_hyps tensor.  out_len = token count - 1 either way.
__global__ void __launch_bounds__(256) beam_finalize_kernel(BeamState s, int cur, int steps_done, int beam, int Lmax,
                                                            int Lout, int topk, int32_t* __restrict__ out_tok,
                                                            int32_t* __restrict__ out_len, float* __restrict__ out_score,
                                                            float* __restrict__ out_lp,
                                                            const int32_t* __restrict__ utt_max,
                                                            int32_t* __restrict__ out_longest) {
  if (utt_max) steps_done = min(steps_done, utt_max[blockIdx.x]);  // grouped search: this utterance's own step count
  __shared__ float e_score[kMaxBeamLarge];
  __shared__ int e_src[kMaxBeamLarge];   // >= 0: finished slot f;  < 0: alive hypothesis -(j+1)
  __shared__ int sel[kMaxBeamLarge];
  __shared__ int n_entries;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int cnt = s.fin_count[b];
    for (int f = 0; f < cnt; ++f) {
      e_score[f] = s.fin_score[b * beam + f];
      e_src[f] = f;
    }
    int longest = 0;  // over ALL final entries of this utterance, filled ones included
    for (int f = 0; f < cnt; ++f) longest = max(longest, s.fin_len[b * beam + f]);
    if (steps_done > 0) {
      if (cnt < beam) longest = max(longest, steps_done);
      for (int j = 0; j < beam && cnt < beam; ++j, ++cnt) {
        e_score[cnt] = s.cand_val[b * beam + j];
        e_src[cnt] = -(j + 1);
      }
    }
    atomicMax(&s.n_full[1], longest);
    if (out_longest) out_longest[b] = longest;
    n_entries = cnt;
    for (int r = 0; r < topk; ++r) {  // selection in descending score order; ties: first entry first
      int bi = -1;
      for (int e = 0; e < cnt; ++e) {
        bool used = false;
        for (int q = 0; q < r; ++q) used |= sel[q] == e;
        if (!used && (bi < 0 || e_score[e] > e_score[bi])) bi = e;
      }
      sel[r] = bi;
    }
  }
  __syncthreads();
  for (int r = 0; r < topk; ++r) {
    const int e = sel[r];
    const size_t row = (size_t)b * topk + r;
    int ntok = 0;
    if (e >= 0) ntok = e_src[e] < 0 ? steps_done : s.fin_len[b * beam + e_src[e]];
    const int len = ntok > 0 ? ntok - 1 : 0;
    if (tid == 0) {
      out_score[row] = e >= 0 ? e_score[e] : 0.0f;
      out_len[row] = len;
    }
    for (int p = tid; p < Lout; p += 256) {  // Lout = row pitch of the outputs (max_steps); Lmax = pitch of the tables
      int tok = 0;
      float lp = 0.0f;
      if (e >= 0 && p < ntok) {  // log-probs keep the stripped position too, like the reference
        const size_t o = e_src[e] < 0 ? ((size_t)b * beam + (-e_src[e] - 1)) * Lmax + p
                                      : ((size_t)b * beam + e_src[e]) * Lmax + p;
        tok = e_src[e] < 0 ? s.seq[cur][o] : s.fin_seq[o];
        lp = e_src[e] < 0 ? s.lp[cur][o] : s.fin_lp[o];
      }
      out_tok[row * Lout + p] = (topk > 1 ? p < ntok : p < len) ? tok : 0;
      out_lp[row * Lout + p] = lp;
    }
  }
}

// ---------------------------------------------------------------- greedy helpers
__global__ void __launch_bounds__(256) greedy_pick_kernel(const float* __restrict__ logits, int32_t* __restrict__ tok_out,
                                                          int32_t* __restrict__ ended, int32_t* __restrict__ hyp,
                                                          float* __restrict__ score, int32_t* __restrict__ n_ended,
                                                          int V, int k, int Lmax, int eos,
                                                          const float* __restrict__ bias1,
                                                          const float* __restrict__ bias2) {
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ float red[4];
  __shared__ float row_lse;
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (size_t)i * V;
  // optional additive logit masks (0 / -inf: suppressed tokens never win and drop out of the normaliser)
  auto at = [&](int c) { return x[c] + (bias1 ? bias1[c] : 0.0f) + (bias2 ? bias2[c] : 0.0f); };
  float v = -INFINITY;
  int a = INT_MAX;
  for (int c = tid; c < V; c += 256) {
    const float xc = at(c);
    if (better(xc, c, v, a)) {
      v = xc;
      a = c;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = sbk::shfl_xor(v, m);
    const int oi = sbk::shfl_xor(a, m);
    if (better(ov, oi, v, a)) {
      v = ov;
      a = oi;
    }
  }
  if ((tid & 63) == 0) {
    wv[tid >> 6] = v;
    wi[tid >> 6] = a;
  }
  __syncthreads();
  float bv = wv[0];
  int bi = wi[0];
  for (int w = 1; w < 4; ++w)
    if (better(wv[w], wi[w], bv, bi)) {
      bv = wv[w];
      bi = wi[w];
    }
  // log-softmax value of the arg-max: max - logsumexp
  float se = 0.0f;
  for (int c = tid; c < V; c += 256) se += expf(at(c) - bv);
  se = sbk::wave_sum(se);
  if ((tid & 63) == 0) red[tid >> 6] = se;
  __syncthreads();
  if (tid == 0) {
    row_lse = logf((red[0] + red[1]) + (red[2] + red[3]));
    const int was = ended[i];
    const int now = was | (bi == eos);
    if (now && !was) atomicAdd(n_ended, 1);
    ended[i] = now;
    // after the end (and on the eos step itself) the reference reports eos with score 0 (seq2seq.py:262-275)
    hyp[(size_t)i * Lmax + k] = now ? eos : bi;
    score[(size_t)i * Lmax + k] = now ? 0.0f : -row_lse;
    tok_out[i] = now ? eos : bi;
  }
}

__global__ void greedy_init_kernel(int32_t* tok, int32_t* ended, int32_t* n_ended, int32_t* kv_slot, int n, int Lmax,
                                   int bos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_ended = 0;
  if (i >= n) return;
  tok[i] = bos;
  ended[i] = 0;
  for (int p = 0; p < Lmax; ++p) kv_slot[(size_t)i * Lmax + p] = i;
}

// out[i] = softmax(logits[i])[token]  (S2SWhisperGreedySearcher: no-speech probability at the start-of-transcript position)
__global__ void __launch_bounds__(256) softmax_prob_kernel(const float* __restrict__ logits, float* __restrict__ out, int V,
                                                           int token) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (size_t)i * V;
  float m = -INFINITY;
  for (int c = tid; c < V; c += 256) m = fmaxf(m, x[c]);
  m = sbk::wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.0f;
  for (int c = tid; c < V; c += 256) se += expf(x[c] - m);
  se = sbk::wave_sum(se);
  if ((tid & 63) == 0) red[tid >> 6] = se;
  __syncthreads();
  if (tid == 0) out[i] = expf(x[token] - m) / ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int d, long dst_stride) {
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dst[(size_t)i * dst_stride + c] = src[(size_t)i * d + c];
}

// ---------------------------------------------------------------- workspace carving
struct Carver {
  char* p;
  size_t used;
  bool dry;
  template <typename T>
  T* take(size_t n) {
    const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    T* r = dry ? nullptr : reinterpret_cast<T*>(p + used);
    used += bytes;
    return r;
  }
};

struct DecoderBufs {
  float *x, *h, *qkv, *ctx, *q, *ff, *logits;
  float* splitk;        // split-K partials of the skinny GEMMs
  size_t splitk_floats;
  float* xpart;         // cross-attention split partials
  int32_t* pbar;        // arrival counter of the persistent few-row step's grid barriers (zeroed by project_memory)
  mutable int pseq;     // persistent launches issued on it since
  mutable int pgrid;    // workgroups of this search's persistent launches (0 until the first one chooses)
  float* ckv[64];
  float *kcache[64], *vcache[64];
};

void carve_decoder(Carver& c, DecoderBufs& d, const sbk_decoder_weights* W, int n, int B, int T, int Lmax) {
  const int dm = W->d_model;
  d.x = c.take<float>((size_t)n * dm);
  d.h = c.take<float>((size_t)n * dm);
  d.qkv = c.take<float>((size_t)n * 3 * dm);
  d.ctx = c.take<float>((size_t)n * dm);
  d.q = c.take<float>((size_t)n * dm);
  d.ff = c.take<float>((size_t)n * W->d_ffn);
  d.logits = c.take<float>((size_t)n * W->vocab);
  d.splitk_floats = (size_t)4 * n * (size_t)dm;  // global split-K is used for the long-K FFN2 only
  d.splitk = c.take<float>(d.splitk_floats);
  d.xpart = c.take<float>(sbk::cross_attn_partial_floats(B, T, W->nhead, dm / W->nhead, n / (B > 0 ? B : 1)) + 64);
  d.pbar = c.take<int32_t>(64 + 512 + 256);  // (+ 256 eight-byte phase stamps of the measurement knob 49, + eight sub-counters of the
                                             //  two-level grid barrier, a 128-byte line each)
  d.pseq = 0;
  d.pgrid = 0;
  for (int l = 0; l < W->n_layers; ++l) {
    d.ckv[l] = c.take<float>((size_t)B * T * 2 * dm);
    d.kcache[l] = c.take<float>((size_t)Lmax * n * dm);
    d.vcache[l] = c.take<float>((size_t)Lmax * n * dm);
  }
}

#define SBK_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) return sbk::fail((int)e__, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

// row pitch (floats) of the CTC posteriors the search keeps: whole 128-byte lines, so that a score workgroup's 1 KB segment of a
// frame row is 8 lines (20 000-byte rows at V = 5 000 are not line-aligned: 9 lines, 1.25 x the bytes at the fabric counter)
static inline int ctc_pitch(int V) { return (V + 31) & ~31; }

#define SBK_TRY(expr)        \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

// Project the encoder memory to the cross-attention K/V of every layer, once per batch.
int project_memory(const sbk_decoder_weights* W, const DecoderBufs& d, const float* enc, int B, int T,
                   hipStream_t st) {
  const int dm = W->d_model;
  SBK_HIP(hipMemsetAsync(d.pbar, 0, (64 + 512 + 256) * sizeof(int32_t), st));
  d.pseq = 0;
  d.pgrid = 0;
  for (int l = 0; l < W->n_layers; ++l) {
    const sbk_decoder_layer& L = W->layers[l];
    float* dst = d.ckv[l];
    int rc = -1;
    if (L.ca_kv_w3 && sbk::x3_routed(B * T, 2 * dm, dm))
      rc = sbk::gemm_nt_x3(enc, dm, L.ca_kv_w3, L.ca_in_b + dm, nullptr, 0, dst, 2 * dm, B * T, 2 * dm, dm, SBK_ACT_NONE, 1.0f,
                           nullptr, 0, st);
    if (rc == -1)  // (not routed, or no workspace for this stream yet: first use inside a graph capture)
      rc = sbk::gemm_nt(enc, dm, L.ca_in_w + (size_t)dm * dm, dm, L.ca_in_b + dm, nullptr, 0, dst, 2 * dm, B * T, 2 * dm, dm,
                        SBK_ACT_NONE, 1.0f, nullptr, 0, st);
    SBK_TRY(rc);
  }
  return 0;
}

// One decoder step for n = B*beam hypotheses at position `step`.  With want_logits: the seq_lin logits in d.logits (d.h is then
// NOT defined: the routes that run decoder.norm inside the vocabulary projection -- sbk_gemm_ln_nt_f32 / _x3r -- never write it);
// without: the final-LayerNorm output in d.h (sbk_decoder_prefix_f32 reads it).
int decoder_step(const sbk_decoder_weights* W, const DecoderBufs& d, const int32_t* tokens, const int32_t* kv_slot,
                 const int32_t* enc_len, int step, int n, int B, int T, int beam, int Lmax, bool want_logits,
                 hipStream_t st) {
  const int dm = W->d_model, H = W->nhead;
  // a beam's worth of rows (<= 16): the whole stack of the step as ONE cooperative launch (csrc/decoder_persist.hip)
  if (!sbk::g_step_ptr && sbk::persist_eligible(W, n, B, beam, Lmax)) {
    const int rc = sbk::decoder_step_persist(W, tokens, kv_slot, enc_len, d.x, d.qkv, d.ctx, d.q, d.ff, d.h, d.logits, d.kcache,
                                             d.vcache, d.ckv, d.pbar, d.pseq, &d.pgrid, step, n, B, T, beam, Lmax, want_logits, st);
    if (rc == 0) {
      ++d.pseq;
      return 0;
    }
    if (rc != -1) return rc;
  }
  const float emb_scale = W->emb_scale > 0.0f ? W->emb_scale : sqrtf((float)dm);  // NormalizedEmbedding: sqrt(d_model)
  SBK_TRY(sbk::embed_pos(tokens, W->emb, W->pe + (size_t)step * dm, d.x, n, dm, emb_scale, st));
  // a projection with a panel image of its weights and enough hypothesis rows: sbk_gemm_nt_x3r (fp32 result on the bf16
  // matrix pipe); -1 = not routed (the register-operand / LDS-tiled fp32-MFMA kernels below)
  auto x3r = [&](const float* A, int lda, const uint16_t* WP, const float* b, const float* R, float* C, int N, int K, int act) -> int {
    if (!WP || !sbk::x3r_routed(n, N, K)) return -1;
    return sbk::gemm_nt_x3r(A, lda, WP, b, R, N, C, N, n, N, K, act, 1.0f, st);
  };
  // the routed kernels move bias rows as 16-byte vectors: a layer with an unaligned bias view takes the fp32 kernels below
  // instead of failing inside the route
  auto biases_aligned = [](const sbk_decoder_layer& L) {
    return sbk::aligned16(L.sa_in_b) && sbk::aligned16(L.sa_out_b) && sbk::aligned16(L.ca_in_b) && sbk::aligned16(L.ca_out_b) &&
           sbk::aligned16(L.ff1_b) && sbk::aligned16(L.ff2_b) && sbk::aligned16(L.sa_in_bf) && sbk::aligned16(L.ca_q_bf) &&
           sbk::aligned16(L.ff1_bf);
  };
  for (int l = 0; l < W->n_layers; ++l) {
    const sbk_decoder_layer& L = W->layers[l];
    if (L.sa_in_wp && L.sa_out_wp && L.ca_q_wp && L.ca_out_wp && L.ff1_wp && L.ff2_wp && sbk::x3r_routed(n, dm, dm) &&
        sbk::x3r_routed(n, dm, W->d_ffn) && biases_aligned(L)) {
      // norm1 / norm2 / norm3 inside the projection they feed (folded panel images, ABI 9; knob 45), else as launches
      const bool lnf = sbk::x3r_ln_routed(dm) && L.sa_in_wfp && L.sa_in_bf && L.ca_q_wfp && L.ca_q_bf && L.ff1_wfp && L.ff1_bf;
      auto ln_proj = [&](const float* g_, const float* b_, const uint16_t* WFP, const float* bfold, const uint16_t* WP, const float* bias,
                         float* C, int N, int act) -> int {
        if (lnf) return sbk::gemm_ln_nt_x3r(d.x, dm, WFP, bfold, nullptr, 0, C, N, n, N, dm, W->ln_eps, act, 1.0f, st);
        SBK_TRY(sbk::layernorm(d.x, g_, b_, d.h, n, dm, W->ln_eps, SBK_ACT_NONE, st));
        return x3r(d.h, dm, WP, bias, nullptr, C, N, dm, act);
      };
      SBK_TRY(ln_proj(L.ln1_g, L.ln1_b, L.sa_in_wfp, L.sa_in_bf, L.sa_in_wp, L.sa_in_b, d.qkv, 3 * dm, SBK_ACT_NONE));
      SBK_TRY(sbk::self_attn_step(d.qkv, d.kcache[l], d.vcache[l], kv_slot, d.ctx, n, dm, H, step, n, Lmax, st, nullptr, 0, 0,
                                  0, 0, beam));
      SBK_TRY(x3r(d.ctx, dm, L.sa_out_wp, L.sa_out_b, d.x, d.x, dm, dm, SBK_ACT_NONE));
      SBK_TRY(ln_proj(L.ln2_g, L.ln2_b, L.ca_q_wfp, L.ca_q_bf, L.ca_q_wp, L.ca_in_b, d.q, dm, SBK_ACT_NONE));
      SBK_TRY(sbk::cross_attn_step(d.q, d.ckv[l], enc_len, d.ctx, d.xpart, B, T, dm, H, beam, st));
      SBK_TRY(x3r(d.ctx, dm, L.ca_out_wp, L.ca_out_b, d.x, d.x, dm, dm, SBK_ACT_NONE));
      SBK_TRY(ln_proj(L.ln3_g, L.ln3_b, L.ff1_wfp, L.ff1_bf, L.ff1_wp, L.ff1_b, d.ff, W->d_ffn, W->ffn_act));
      SBK_TRY(x3r(d.ff, W->d_ffn, L.ff2_wp, L.ff2_b, d.x, d.x, dm, W->d_ffn, SBK_ACT_NONE));
      continue;
    }
    int frc = L.sa_in_wf ? sbk::gemm_ln_nt(d.x, dm, L.sa_in_wf, dm, L.sa_in_bf, nullptr, 0, d.qkv, 3 * dm, n, 3 * dm, dm,
                                           W->ln_eps, SBK_ACT_NONE, 1.0f, st)
                         : -1;
    if (frc > 0 || frc < -1) return frc;
    if (frc == -1) {
      SBK_TRY(sbk::layernorm(d.x, L.ln1_g, L.ln1_b, d.h, n, dm, W->ln_eps, SBK_ACT_NONE, st));
      SBK_TRY(sbk::gemm_nt_ws(d.h, dm, L.sa_in_w, dm, L.sa_in_b, nullptr, 0, d.qkv, 3 * dm, n, 3 * dm, dm, SBK_ACT_NONE,
                              1.0f, nullptr, 0, d.splitk, d.splitk_floats, st));
    }
    SBK_TRY(sbk::self_attn_step(d.qkv, d.kcache[l], d.vcache[l], kv_slot, d.ctx, n, dm, H, step, n, Lmax, st, nullptr, 0, 0,
                                0, 0, beam));
    SBK_TRY(sbk::gemm_nt_ws(d.ctx, dm, L.sa_out_w, dm, L.sa_out_b, d.x, dm, d.x, dm, n, dm, dm, SBK_ACT_NONE, 1.0f, nullptr,
                         0, d.splitk, d.splitk_floats, st));
    frc = L.ca_q_wf ? sbk::gemm_ln_nt(d.x, dm, L.ca_q_wf, dm, L.ca_q_bf, nullptr, 0, d.q, dm, n, dm, dm, W->ln_eps,
                                      SBK_ACT_NONE, 1.0f, st)
                    : -1;
    if (frc > 0 || frc < -1) return frc;
    if (frc == -1) {
      SBK_TRY(sbk::layernorm(d.x, L.ln2_g, L.ln2_b, d.h, n, dm, W->ln_eps, SBK_ACT_NONE, st));
      SBK_TRY(sbk::gemm_nt_ws(d.h, dm, L.ca_in_w, dm, L.ca_in_b, nullptr, 0, d.q, dm, n, dm, dm, SBK_ACT_NONE, 1.0f, nullptr,
                              0, d.splitk, d.splitk_floats, st));
    }
    SBK_TRY(sbk::cross_attn_step(d.q, d.ckv[l], enc_len, d.ctx, d.xpart, B, T, dm, H, beam, st));
    SBK_TRY(sbk::gemm_nt_ws(d.ctx, dm, L.ca_out_w, dm, L.ca_out_b, d.x, dm, d.x, dm, n, dm, dm, SBK_ACT_NONE, 1.0f, nullptr,
                         0, d.splitk, d.splitk_floats, st));
    frc = L.ff1_wf ? sbk::gemm_ln_nt(d.x, dm, L.ff1_wf, dm, L.ff1_bf, nullptr, 0, d.ff, W->d_ffn, n, W->d_ffn, dm, W->ln_eps,
                                     W->ffn_act, 1.0f, st)
                   : -1;
    if (frc > 0 || frc < -1) return frc;
    if (frc == -1) {
      SBK_TRY(sbk::layernorm(d.x, L.ln3_g, L.ln3_b, d.h, n, dm, W->ln_eps, SBK_ACT_NONE, st));
      SBK_TRY(sbk::gemm_nt_ws(d.h, dm, L.ff1_w, dm, L.ff1_b, nullptr, 0, d.ff, W->d_ffn, n, W->d_ffn, dm, W->ffn_act, 1.0f,
                              nullptr, 0, d.splitk, d.splitk_floats, st));
    }
    SBK_TRY(sbk::gemm_nt_ws(d.ff, W->d_ffn, L.ff2_w, W->d_ffn, L.ff2_b, d.x, dm, d.x, dm, n, dm, W->d_ffn, SBK_ACT_NONE, 1.0f,
                         nullptr, 0, d.splitk, d.splitk_floats, st));
  }
  if (want_logits && W->seq_wf) {  // final LayerNorm fused into seq_lin
    const int frc = sbk::gemm_ln_nt(d.x, dm, W->seq_wf, dm, W->seq_bf, nullptr, 0, d.logits, W->vocab, n, W->vocab, dm,
                                    W->ln_eps, SBK_ACT_NONE, 1.0f, st);
    if (frc != -1) return frc;
  }
  // decoder.norm inside the vocabulary projection -- for narrow vocabularies only: every column tile's workgroup repeats the
  // row statistics, which at V = 5 000 (79 column tiles) costs what the LayerNorm launch does (58.6 vs 59.0 us at 1 280 rows,
  // 105 vs 117 at 2 560: profiles/r04_s_*); knob 45 = 2 forces it; with handed-over statistics (3) the prologue reads 128 B per row
  if (want_logits && W->seq_wfp && W->seq_bf && sbk::x3r_ln_routed(dm) && sbk::x3r_routed(n, W->vocab, dm) &&
      (W->vocab < 4096 || sbk::g_x3r_ln == 2)) {
    const int rc = sbk::gemm_ln_nt_x3r(d.x, dm, W->seq_wfp, W->seq_bf, nullptr, 0, d.logits, W->vocab, n, W->vocab, dm, W->ln_eps,
                                       SBK_ACT_NONE, 1.0f, st);
    if (rc != -1) return rc;
  }
  SBK_TRY(sbk::layernorm(d.x, W->final_ln_g, W->final_ln_b, d.h, n, dm, W->ln_eps, SBK_ACT_NONE, st));
  if (want_logits) {
    int rc = x3r(d.h, dm, W->seq_wp, W->seq_b, nullptr, d.logits, W->vocab, dm, SBK_ACT_NONE);
    if (rc == -1 && W->seq_w3 && sbk::x3_routed(n, W->vocab, dm))
      rc = sbk::gemm_nt_x3(d.h, dm, W->seq_w3, W->seq_b, nullptr, 0, d.logits, W->vocab, n, W->vocab, dm, SBK_ACT_NONE, 1.0f,
                           nullptr, 0, st);
    if (rc == -1)
      rc = sbk::gemm_nt_ws(d.h, dm, W->seq_w, dm, W->seq_b, nullptr, 0, d.logits, W->vocab, n, W->vocab, dm, SBK_ACT_NONE, 1.0f,
                           nullptr, 0, d.splitk, d.splitk_floats, st);
    SBK_TRY(rc);
  }
  return 0;
}

// ---------------------------------------------------------------- TransformerLM scorer (a20)
struct LmBufs {
  float *x, *t, *h, *qkv, *ctx, *ff, *logits;
  float* splitk;
  size_t splitk_floats;
  float *kcache[64], *vcache[64];
};

void carve_lm(Carver& c, LmBufs& d, const sbk_lm_weights* LM, int n, int Lmax) {
  const int dm = LM->d_model;
  d.x = c.take<float>((size_t)n * dm);
  d.t = c.take<float>((size_t)n * dm);
  d.h = c.take<float>((size_t)n * dm);
  d.qkv = c.take<float>((size_t)n * 3 * dm);
  d.ctx = c.take<float>((size_t)n * dm);
  d.ff = c.take<float>((size_t)n * LM->d_ffn);
  d.logits = c.take<float>((size_t)n * LM->vocab);
  d.splitk_floats = (size_t)4 * n * (size_t)dm;
  d.splitk = c.take<float>(d.splitk_floats);
  for (int l = 0; l < LM->n_layers; ++l) {
    d.kcache[l] = c.take<float>((size_t)Lmax * n * dm);
    d.vcache[l] = c.take<float>((size_t)Lmax * n * dm);
  }
}

int check_lm(const sbk_lm_weights* LM) {
  SBK_REQUIRE(LM && LM->layers && LM->emb && LM->pe && LM->final_ln_g && LM->final_ln_b && LM->out0_w && LM->out0_b &&
                  LM->out_ln_g && LM->out_ln_b && LM->out2_w && LM->out2_b,
              "lm weights: null pointer");
  SBK_REQUIRE(LM->n_layers > 0 && LM->n_layers <= 64 && LM->d_model > 0 && LM->nhead > 0 &&
                  LM->d_model % LM->nhead == 0 && LM->vocab > 0,
              "lm weights: bad shape");
  return 0;
}

// One TransformerLM step for n hypotheses at position `step` (TransformerLM.forward restricted to the
// newest position; earlier positions' K/V sit in the slot-addressed cache shared with the decoder's
// ancestry table).  Leaves output_proj logits in d.logits.
int lm_step(const sbk_lm_weights* LM, const LmBufs& d, const int32_t* tokens, const int32_t* kv_slot,
            const int32_t* key_tok, int key_stride, int key_shift, int key_first, int step, int n, int Lmax,
            hipStream_t st) {
  const int dm = LM->d_model, H = LM->nhead, ff = LM->d_ffn;
  const float eps = LM->ln_eps;
  auto gemm = [&](const float* A, int lda, const float* Wt, const float* b, const float* R, float* C, int N, int K,
                  int act) {
    return sbk::gemm_nt_ws(A, lda, Wt, K, b, R, R ? N : 0, C, N, n, N, K, act, 1.0f, nullptr, 0, d.splitk,
                           d.splitk_floats, st);
  };
  SBK_TRY(sbk::embed_pos(tokens, LM->emb, LM->pe + (size_t)step * dm, d.x, n, dm, sqrtf((float)dm), st));
  for (int l = 0; l < LM->n_layers; ++l) {
    const sbk_lm_layer& L = LM->layers[l];
    if (LM->normalize_before) {  // Transformer.py:452-480 with normalize_before
      SBK_TRY(sbk::layernorm(d.x, L.ln1_g, L.ln1_b, d.h, n, dm, eps, SBK_ACT_NONE, st));
      SBK_TRY(gemm(d.h, dm, L.in_w, L.in_b, nullptr, d.qkv, 3 * dm, dm, SBK_ACT_NONE));
      SBK_TRY(sbk::self_attn_step(d.qkv, d.kcache[l], d.vcache[l], kv_slot, d.ctx, n, dm, H, step, n, Lmax, st, key_tok,
                                  key_stride, key_shift, key_first, LM->pad_idx));
      SBK_TRY(gemm(d.ctx, dm, L.out_w, L.out_b, d.x, d.x, dm, dm, SBK_ACT_NONE));
      SBK_TRY(sbk::layernorm(d.x, L.ln2_g, L.ln2_b, d.h, n, dm, eps, SBK_ACT_NONE, st));
      SBK_TRY(gemm(d.h, dm, L.ff1_w, L.ff1_b, nullptr, d.ff, ff, dm, LM->ffn_act));
      SBK_TRY(gemm(d.ff, ff, L.ff2_w, L.ff2_b, d.x, d.x, dm, ff, SBK_ACT_NONE));
    } else {  // post-norm
      SBK_TRY(gemm(d.x, dm, L.in_w, L.in_b, nullptr, d.qkv, 3 * dm, dm, SBK_ACT_NONE));
      SBK_TRY(sbk::self_attn_step(d.qkv, d.kcache[l], d.vcache[l], kv_slot, d.ctx, n, dm, H, step, n, Lmax, st, key_tok,
                                  key_stride, key_shift, key_first, LM->pad_idx));
      SBK_TRY(gemm(d.ctx, dm, L.out_w, L.out_b, d.x, d.t, dm, dm, SBK_ACT_NONE));
      SBK_TRY(sbk::layernorm(d.t, L.ln1_g, L.ln1_b, d.x, n, dm, eps, SBK_ACT_NONE, st));
      SBK_TRY(gemm(d.x, dm, L.ff1_w, L.ff1_b, nullptr, d.ff, ff, dm, LM->ffn_act));
      SBK_TRY(gemm(d.ff, ff, L.ff2_w, L.ff2_b, d.x, d.t, dm, ff, SBK_ACT_NONE));
      SBK_TRY(sbk::layernorm(d.t, L.ln2_g, L.ln2_b, d.x, n, dm, eps, SBK_ACT_NONE, st));
    }
  }
  SBK_TRY(sbk::layernorm(d.x, LM->final_ln_g, LM->final_ln_b, d.h, n, dm, eps, SBK_ACT_NONE, st));
  SBK_TRY(gemm(d.h, dm, LM->out0_w, LM->out0_b, nullptr, d.t, dm, dm, SBK_ACT_NONE));
  SBK_TRY(sbk::layernorm(d.t, LM->out_ln_g, LM->out_ln_b, d.h, n, dm, eps, SBK_ACT_NONE, st));
  SBK_TRY(gemm(d.h, dm, LM->out2_w, LM->out2_b, nullptr, d.logits, LM->vocab, dm, SBK_ACT_NONE));
  return 0;
}

int check_weights(const sbk_decoder_weights* W) {
  SBK_REQUIRE(W && W->layers && W->emb && W->pe && W->final_ln_g && W->final_ln_b, "decoder weights: null pointer");
  SBK_REQUIRE(W->n_layers > 0 && W->n_layers <= 64 && W->d_model > 0 && W->nhead > 0 && W->d_model % W->nhead == 0,
              "decoder weights: bad shape");
  return 0;
}

struct BeamBufs {
  BeamState s;
  float *am, *comb, *psi, *am_max, *ctc_x, *ctc_xb, *phi[2], *psi_prev[2];
  float* topk_val;
  int32_t* topk_idx;
  // attention window of the CTC scorer (ctc_window_size > 0): running arg-max over the decoded positions of the
  // head-averaged cross-attention, per (hypothesis, frame), double-buffered like the other per-hypothesis tables
  float* attn_avg;       // [n_bh][T] this step's probabilities
  float* peak_val[2];    // [n_bh][T]
  int32_t* peak_pos[2];  // [n_bh][T]
  int32_t* win;          // [Lmax][2] {min, max} peak of every step
};

// tables of the attention window: new = old[parent], then the new position takes over where its probability is larger
// (strictly: torch.max returns the FIRST maximum); the step's min / max position over every hypothesis and frame
__global__ void __launch_bounds__(256) attn_peak_kernel(const float* __restrict__ probs, const float* __restrict__ old_val,
                                                        const int32_t* __restrict__ old_pos,
                                                        const int32_t* __restrict__ parent, float* __restrict__ new_val,
                                                        int32_t* __restrict__ new_pos, int32_t* __restrict__ win, int T,
                                                        int step) {
  __shared__ int lo_s[4], hi_s[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int p = step > 0 ? parent[n] : n;
  int lo = INT_MAX, hi = INT_MIN;
  for (int t = tid; t < T; t += 256) {
    float v = step > 0 ? old_val[(size_t)p * T + t] : -1.0f;
    int pos = step > 0 ? old_pos[(size_t)p * T + t] : 0;
    const float a = probs[(size_t)n * T + t];
    if (a > v) {
      v = a;
      pos = step;
    }
    new_val[(size_t)n * T + t] = v;
    new_pos[(size_t)n * T + t] = pos;
    lo = min(lo, pos);
    hi = max(hi, pos);
  }
  for (int m = 32; m >= 1; m >>= 1) {
    lo = min(lo, sbk::shfl_xor(lo, m));
    hi = max(hi, sbk::shfl_xor(hi, m));
  }
  if ((tid & 63) == 0) {
    lo_s[tid >> 6] = lo;
    hi_s[tid >> 6] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    atomicMin(&win[2 * step], min(min(lo_s[0], lo_s[1]), min(lo_s[2], lo_s[3])));
    atomicMax(&win[2 * step + 1], max(max(hi_s[0], hi_s[1]), max(hi_s[2], hi_s[3])));
  }
}
__global__ void win_init_kernel(int32_t* win, int steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < steps) {
    win[2 * i] = INT_MAX;
    win[2 * i + 1] = INT_MIN;
  }
}

void carve_beam(Carver& c, BeamBufs& b, int B, int beam, int T, int V, int Lmax, bool ctc, int window = 0) {
  const size_t n = (size_t)B * beam;
  for (int k = 0; k < 2; ++k) {
    b.s.seq[k] = c.take<int32_t>(n * Lmax);
    b.s.lp[k] = c.take<float>(n * Lmax);
    b.s.kv_slot[k] = c.take<int32_t>(n * Lmax);
    b.s.tokens[k] = c.take<int32_t>(n);
  }
  b.s.seq_scores = c.take<float>(n);
  b.s.cand_val = c.take<float>(n);
  b.s.cand_idx = c.take<int32_t>(n);
  b.s.parent = c.take<int32_t>(n);
  b.s.fin_count = c.take<int32_t>(B);
  b.s.fin_seq = c.take<int32_t>(n * Lmax);
  b.s.fin_lp = c.take<float>(n * Lmax);
  b.s.fin_len = c.take<int32_t>(n);
  b.s.fin_score = c.take<float>(n);
  b.s.n_full = c.take<int32_t>(64);
  b.am = c.take<float>(n * V);
  b.comb = c.take<float>(n * V);
  b.am_max = c.take<float>(n);
  b.topk_val = c.take<float>((size_t)B * kTopkChunks * kMaxBeam);
  b.topk_idx = c.take<int32_t>((size_t)B * kTopkChunks * kMaxBeam);
  if (ctc) {
    b.psi = c.take<float>(n * V);
    b.ctc_x = c.take<float>((size_t)B * T * ctc_pitch(V));  // frame rows padded to whole 128-byte lines
    b.ctc_xb = c.take<float>((size_t)B * T);
    for (int k = 0; k < 2; ++k) {
      b.phi[k] = c.take<float>(sbk::ctc_state_floats(B, beam, T));
      b.psi_prev[k] = c.take<float>(n);
    }
    if (window > 0) {
      b.attn_avg = c.take<float>(n * T);
      for (int k = 0; k < 2; ++k) {
        b.peak_val[k] = c.take<float>(n * T);
        b.peak_pos[k] = c.take<int32_t>(n * T);
      }
      b.win = c.take<int32_t>((size_t)2 * Lmax);
    }
  }
}

}  // namespace

extern "C" size_t sbk_beam_search_workspace_bytes(const sbk_decoder_weights* W, const sbk_search_config* cfg, int B,
                                                  int T) {
  if (!W || !cfg) return 0;
  Carver c{nullptr, 0, true};
  DecoderBufs d;
  BeamBufs bb;
  const int Lmax = (cfg->max_steps > 0 ? cfg->max_steps : 1) + (cfg->prompt && cfg->prompt_len > 1 ? cfg->prompt_len - 1 : 0);
  carve_decoder(c, d, W, B * cfg->beam, B, T, Lmax);
  carve_beam(c, bb, B, cfg->beam, T, W->vocab, Lmax, cfg->ctc_weight > 0.0f, cfg->ctc_window_size);
  if (cfg->lm && cfg->lm_weight != 0.0f) {
    LmBufs lb;
    carve_lm(c, lb, cfg->lm, B * cfg->beam, Lmax);
  }
  return c.used + 256;
}

extern "C" int sbk_beam_search_f32(const sbk_decoder_weights* W, const sbk_search_config* cfg, const float* enc,
                                   const int32_t* enc_len, const float* ctc_w, const float* ctc_b, void* workspace,
                                   size_t workspace_bytes, int32_t* out_tokens, int32_t* out_len, float* out_score,
                                   float* out_logp, int32_t* out_max_len, int32_t* out_longest, int32_t* host_flag,
                                   int32_t* steps_run, int B, int T, sbk_stream_t stream) {
  SBK_TRY(check_weights(W));
  if (B == 0) {  // empty batch: nothing to launch, the data pointers may be NULL
    if (steps_run) *steps_run = 0;
    return 0;
  }
  SBK_REQUIRE(cfg && enc && enc_len && workspace && out_tokens && out_len && out_score && out_logp, "beam_search: null");
  SBK_REQUIRE(W->seq_w && W->seq_b, "beam_search: seq_lin weights missing");
  SBK_REQUIRE(cfg->beam >= 1 && cfg->beam <= kMaxBeamLarge, "beam_search: beam %d outside [1,%d]", cfg->beam,
              kMaxBeamLarge);
  const int pos_off = cfg->prompt && cfg->prompt_len > 1 ? cfg->prompt_len - 1 : 0;  // decoder position of search step 0
  SBK_REQUIRE(pos_off + cfg->max_steps <= W->max_len, "beam_search: %d prompt + %d steps exceed the positional table (%d)",
              pos_off, cfg->max_steps, W->max_len);
  SBK_REQUIRE(!cfg->prompt || (cfg->prompt_len >= 1 && !(cfg->lm && cfg->lm_weight != 0.0f) && !cfg->utt_max_steps &&
                               !cfg->utt_min_steps),
              "beam_search: a token prompt excludes the LM scorer and the grouped search");
  SBK_REQUIRE(!cfg->out_probe || (cfg->probe_pos >= 0 && cfg->probe_pos <= pos_off && cfg->probe_token >= 0 &&
                                  cfg->probe_token < W->vocab), "beam_search: probe position / token out of range");
  const bool ctc = cfg->ctc_weight > 0.0f;
  const int topk = cfg->topk > 1 ? cfg->topk : 1;
  SBK_REQUIRE(topk <= cfg->beam, "beam_search: topk %d exceeds the beam %d", topk, cfg->beam);
  SBK_REQUIRE(!ctc || (ctc_w && ctc_b), "beam_search: ctc_weight > 0 needs the ctc_lin weights");
  SBK_REQUIRE(!ctc || (cfg->bos != cfg->eos && cfg->bos != cfg->blank && cfg->eos != cfg->blank),
              "Set blank, eos and bos to different indexes for joint ATT/CTC or CTC decoding");
  SBK_REQUIRE(workspace_bytes >= sbk_beam_search_workspace_bytes(W, cfg, B, T), "beam_search: workspace too small");
  SBK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "beam_search: workspace must be 256-byte aligned");
  hipStream_t st = sbk::as_stream(stream);
  const int beam = cfg->beam, V = W->vocab, n = B * beam, dm = W->d_model;
  const int Lmax = (cfg->max_steps > 0 ? cfg->max_steps : 1) + pos_off;
  if (steps_run) *steps_run = 0;
  if (B == 0) return 0;

  Carver c{static_cast<char*>(workspace), 0, false};
  DecoderBufs d;
  BeamBufs bb;
  carve_decoder(c, d, W, n, B, T, Lmax);
  carve_beam(c, bb, B, beam, T, V, Lmax, ctc, cfg->ctc_window_size);
  const sbk_lm_weights* LM = (cfg->lm && cfg->lm_weight != 0.0f) ? cfg->lm : nullptr;
  LmBufs lb;
  if (LM) {
    SBK_TRY(check_lm(LM));
    SBK_REQUIRE(LM->vocab == V, "beam_search: the LM scores %d tokens, the decoder %d", LM->vocab, V);
    SBK_REQUIRE(cfg->max_steps <= LM->max_len, "beam_search: %d steps exceed the LM positional table", cfg->max_steps);
    SBK_REQUIRE(cfg->lm_temperature > 0.0f, "beam_search: lm_temperature must be positive");
    carve_lm(c, lb, LM, n, Lmax);
  }
  const float* extra = LM ? lb.logits : nullptr;  // weighted LM log-probs of the step

  const int Vp = ctc_pitch(V);  // row pitch of the CTC posteriors
  SBK_TRY(project_memory(W, d, enc, B, T, st));
  if (ctc) {  // CTCScorer.reset_mem (scorer.py:239-255): log_softmax(ctc_lin(enc)), then the frame mask
    int rc = -1;
    if (cfg->ctc_w3 && sbk::x3_routed(B * T, V, dm))
      rc = sbk::gemm_nt_x3(enc, dm, cfg->ctc_w3, ctc_b, nullptr, 0, bb.ctc_x, Vp, B * T, V, dm, SBK_ACT_NONE, 1.0f, nullptr, 0, st);
    if (rc == -1) rc = sbk::gemm_nt(enc, dm, ctc_w, dm, ctc_b, nullptr, 0, bb.ctc_x, Vp, B * T, V, dm, SBK_ACT_NONE, 1.0f, nullptr, 0, st);
    SBK_TRY(rc);
    SBK_TRY(sbk::log_softmax_rows(bb.ctc_x, bb.ctc_x, B * T, V, 1.0f, 1.0f, st, nullptr, nullptr, Vp));
    SBK_HIP(hipMemsetAsync(bb.phi[1], 0, sbk::ctc_state_floats(B, beam, T) * sizeof(float), st));  // zero table padding
    SBK_TRY(sbk::ctc_prepare(bb.ctc_x, bb.ctc_xb, enc_len, bb.phi[0], bb.psi_prev[0], B, T, V, beam, cfg->blank, st, Vp));
  }
  bb.s.pos_off = pos_off;
  SBK_LAUNCH(beam_init_kernel, dim3(sbk::cdiv(n > B ? n : B, 256)), dim3(256), 0, st, bb.s, B, beam, cfg->bos);
  SBK_TRY(sbk::launch_status("beam_init"));
  if (pos_off > 0) {  // S2SWhisperBeamSearcher.reset_mem (seq2seq.py:2104-2121): the prompt but its last token primes the KV cache
    SBK_LAUNCH(kv_prompt_init_kernel, dim3(sbk::cdiv(n * pos_off, 256)), dim3(256), 0, st, bb.s, n, Lmax, pos_off);
    for (int p = 0; p < pos_off; ++p) {
      const bool probe = cfg->out_probe && p == cfg->probe_pos;
      SBK_LAUNCH(prompt_col_kernel, dim3(sbk::cdiv(n, 256)), dim3(256), 0, st, cfg->prompt, bb.s.tokens[1], n, beam, pos_off + 1, p);
      SBK_TRY(decoder_step(W, d, bb.s.tokens[1], bb.s.kv_slot[0], enc_len, p, n, B, T, beam, Lmax, probe, st));
      if (probe) {
        SBK_LAUNCH(softmax_prob_kernel, dim3(n), dim3(256), 0, st, (const float*)d.logits, bb.am_max, W->vocab, cfg->probe_token);
        SBK_LAUNCH(probe_pick_kernel, dim3(sbk::cdiv(B, 256)), dim3(256), 0, st, (const float*)bb.am_max, cfg->out_probe, B, beam);
      }
    }
    SBK_LAUNCH(prompt_col_kernel, dim3(sbk::cdiv(n, 256)), dim3(256), 0, st, cfg->prompt, bb.s.tokens[0], n, beam, pos_off + 1, pos_off);
    SBK_TRY(sbk::launch_status("beam_search prompt"));
  }

  const int window = ctc ? cfg->ctc_window_size : 0;
  SBK_REQUIRE(window >= 0 && (window == 0 || (!cfg->utt_max_steps && !cfg->utt_min_steps)),
              "beam_search: ctc_window_size excludes the grouped search");
  if (window > 0) {
    SBK_LAUNCH(win_init_kernel, dim3(sbk::cdiv(Lmax, 256)), dim3(256), 0, st, bb.win, Lmax);
    SBK_TRY(sbk::launch_status("ctc window"));
  }
  const float attn_w = ctc ? 1.0f - cfg->ctc_weight : 1.0f;  // seq2seq.py:803-804
  // overlap_ctc bit 0: survivors' CTC state (ctc_advance: one wave per hypothesis, latency-bound) on the
  // helper stream beside the next decoder step; bit 1: the full-vocabulary psi pass there as well.
  SideStream* side = (ctc && cfg->overlap_ctc && window == 0) ? side_stream() : nullptr;
  const bool psi_aside = side && (cfg->overlap_ctc & 2);
  hipStream_t cst = side ? side->s : st;      // stream of ctc_advance
  hipStream_t pst = psi_aside ? side->s : st;  // stream of ctc_psi_step
  if (ctc && psi_aside) {
    SBK_HIP(hipEventRecord(side->fork, st));
    SBK_HIP(hipStreamWaitEvent(cst, side->fork, 0));
    SBK_TRY(sbk::ctc_psi_step(bb.ctc_x, bb.phi[0], bb.s.tokens[0], enc_len, bb.psi, B, T, V, beam, 0, cfg->blank,
                              cfg->eos, pst, nullptr, 0, Vp));
    SBK_HIP(hipEventRecord(side->join, cst));
  }
  // One decoding step as a list of launches.  `counter` = true: the step number lives in device memory
  // (bb.s.n_full[2]; every step-dependent kernel reads it, sbk::g_step_ptr), `step` is then only a
  // placeholder 0 and the launches are identical for every step -- what a captured hipGraph needs.
  int32_t* step_dev = bb.s.n_full + 2;
  auto issue_step = [&](int step, int cur, bool counter) -> int {
    SBK_TRY(decoder_step(W, d, bb.s.tokens[cur], bb.s.kv_slot[cur], enc_len, pos_off + step, n, B, T, beam, Lmax, true, st));
    if (step == 0 && cfg->out_probe && cfg->probe_pos == pos_off) {  // (the probe sits on the last prompt token)
      SBK_LAUNCH(softmax_prob_kernel, dim3(n), dim3(256), 0, st, (const float*)d.logits, bb.am_max, W->vocab, cfg->probe_token);
      SBK_LAUNCH(probe_pick_kernel, dim3(sbk::cdiv(B, 256)), dim3(256), 0, st, (const float*)bb.am_max, cfg->out_probe, B, beam);
    }
    // scoring: one fused pass per hypothesis row (score_topk_row_kernel) unless the beam / vocabulary exceed its
    // register lists or CTC is a PARTIAL scorer (its k-th-largest mask needs the whole row first)
    const bool fused = sbk::g_score_fused && beam <= kMaxBeam && V <= 256 * 32 && !(ctc && cfg->ctc_candidates > 0);
    // Whisper: log_softmax(logits + masks) / temperature (seq2seq.py:2176-2192); otherwise log_softmax(logits / temperature)
    const float ls_temp = cfg->temperature_post ? 1.0f : cfg->temperature;
    const float ls_w = cfg->temperature_post ? attn_w / cfg->temperature : attn_w;
    const float* first_bias = step == 0 ? cfg->first_bias : nullptr;
    if (!fused) {
      SBK_TRY(sbk::log_softmax_rows(d.logits, bb.am, n, V, ls_temp, ls_w, st, cfg->logit_bias, first_bias));
      if (cfg->using_eos_threshold) SBK_TRY(sbk::row_max(bb.am, bb.am_max, n, V, st));
    }
    const int eos_floor = step < cfg->min_steps;
    if (LM) {  // TransformerLMScorer.score (scorer.py:510-543): the prefix is the decoder's own token history
      SBK_TRY(lm_step(LM, lb, bb.s.tokens[cur], bb.s.kv_slot[cur], bb.s.seq[cur], Lmax, 1, cfg->bos, step, n, Lmax, st));
      SBK_TRY(sbk::log_softmax_rows(lb.logits, lb.logits, n, V, cfg->lm_temperature, cfg->lm_weight, st));
    }
    const int32_t* win = nullptr;
    if (window > 0) {  // attention peaks of this step (the last decoder layer's head-averaged cross-attention)
      SBK_TRY(sbk::cross_attn_avg_probs(d.q, d.ckv[W->n_layers - 1], enc_len, bb.attn_avg, n, T, dm, W->nhead, beam, st));
      SBK_LAUNCH(attn_peak_kernel, dim3(n), dim3(256), 0, st, (const float*)bb.attn_avg, (const float*)bb.peak_val[cur],
                 (const int32_t*)bb.peak_pos[cur], (const int32_t*)bb.s.parent, bb.peak_val[cur ^ 1], bb.peak_pos[cur ^ 1],
                 bb.win, T, step);
      SBK_TRY(sbk::launch_status("attn_peak"));
      win = bb.win + 2 * step;
    }
    if (ctc) {
      if (side && (psi_aside || step > 0)) SBK_HIP(hipStreamWaitEvent(st, side->join, 0));  // helper-stream work done
      if (!psi_aside)
        SBK_TRY(sbk::ctc_psi_step(bb.ctc_x, bb.phi[cur], bb.s.tokens[cur], enc_len, bb.psi, B, T, V, beam, step,
                                  cfg->blank, cfg->eos, st, win, window, Vp));
      if (fused) {  // (combined inside score_topk_row_kernel)
      } else if (cfg->ctc_candidates > 0) {  // CTC as a partial scorer: only the top candidates of every hypothesis are scored
        SBK_TRY(sbk::am_only(bb.am, bb.comb, n, V, cfg->eos, eos_floor, cfg->using_eos_threshold, cfg->eos_threshold,
                             cfg->minus_inf, bb.am_max, extra, st, cfg->utt_min_steps, beam, step));
        SBK_TRY(sbk::ctc_partial_combine(bb.comb, bb.topk_val, bb.psi, bb.psi_prev[cur], n, V, cfg->ctc_candidates,
                                         cfg->blank, cfg->eos, cfg->ctc_weight, cfg->minus_inf, st));
      } else {
        SBK_TRY(sbk::ctc_combine(bb.am, bb.am_max, bb.psi, bb.psi_prev[cur], bb.comb, n, V, cfg->blank, cfg->eos,
                                 cfg->ctc_weight, eos_floor, cfg->using_eos_threshold, cfg->eos_threshold,
                                 cfg->minus_inf, extra, st, cfg->utt_min_steps, beam, step));
      }
    } else if (!fused) {
      SBK_TRY(sbk::am_only(bb.am, bb.comb, n, V, cfg->eos, eos_floor, cfg->using_eos_threshold, cfg->eos_threshold,
                           cfg->minus_inf, bb.am_max, extra, st, cfg->utt_min_steps, beam, step));
    }
    const float norm = cfg->length_normalization ? (float)(step + 1) : 0.0f;
    const int32_t* sp = sbk::g_step_ptr;  // a local: launch arguments must not name the thread_local itself
    const int32_t* umax = cfg->utt_max_steps;
    if (fused) {
      sbk::ProfScope prof("score_topk", 8.0 * n * V, (ctc ? 12.0 : 8.0) * n * V, st);
      ScoreArgs a{d.logits, cfg->logit_bias, first_bias, bb.am, ctc ? bb.psi : nullptr, ctc ? bb.psi_prev[cur] : nullptr,
                  extra, bb.s.seq_scores, bb.topk_val, bb.topk_idx, sp, cfg->utt_min_steps, umax, V, beam, step,
                  cfg->min_steps, eos_floor, cfg->using_eos_threshold, cfg->eos, cfg->blank, 1.0f / ls_temp, ls_w,
                  cfg->ctc_weight, cfg->eos_threshold, cfg->minus_inf, norm};
      if (V <= 256 * 4)
        SBK_LAUNCH(score_topk_row_kernel<4>, dim3(n), dim3(256), 0, st, a);
      else if (V <= 256 * 20)
        SBK_LAUNCH(score_topk_row_kernel<20>, dim3(n), dim3(256), 0, st, a);
      else
        SBK_LAUNCH(score_topk_row_kernel<32>, dim3(n), dim3(256), 0, st, a);
      // (the merge of the rows' winners rides in front of the bookkeeping: beam_merge_update_kernel below)
    } else if (beam > kMaxBeam) {
      sbk::ProfScope prof("beam_topk_large", 10.0 * n * V, 20.0 * n * V, st);
      SBK_LAUNCH(beam_topk_large_kernel, dim3(B), dim3(1024), 0, st, (const float*)bb.comb,
                 (const float*)bb.s.seq_scores, bb.s.cand_val, bb.s.cand_idx, V, beam, norm, sp, step, umax);
    } else {
      sbk::ProfScope prof("beam_topk", 2.0 * n * V, 4.0 * n * V, st);
      SBK_LAUNCH(beam_topk_stage1_kernel, dim3(kTopkChunks, B), dim3(256), 0, st, (const float*)bb.comb,
                 (const float*)bb.s.seq_scores, bb.topk_val, bb.topk_idx, V, beam, norm, sp, step, umax);
      SBK_LAUNCH(beam_topk_stage2_kernel, dim3(B), dim3(256), 0, st, (const float*)bb.topk_val,
                 (const int32_t*)bb.topk_idx, bb.s.cand_val, bb.s.cand_idx, beam, sp, step, umax, kTopkChunks);
    }
    SBK_TRY(sbk::launch_status("beam_topk"));
    {
      sbk::ProfScope prof("beam_update", 0.0, 24.0 * n * (step + 1), st);
      if (fused)
        SBK_LAUNCH(beam_merge_update_kernel, dim3(B), dim3(256), 0, st, bb.s, (const float*)bb.topk_val,
                   (const int32_t*)bb.topk_idx, (const float*)bb.am, cur, step, V, beam, Lmax, cfg->eos,
                   cfg->length_normalization, sp, umax, beam);
      else
        SBK_LAUNCH(beam_update_kernel, dim3(B), dim3(256), 0, st, bb.s, (const float*)bb.am, cur, step, V, beam, Lmax,
                   cfg->eos, cfg->length_normalization, sp, umax);
    }
    SBK_TRY(sbk::launch_status("beam_update"));
    // survivors' CTC state, then the next step's psi -- beside the next decoder step (with a device-side
    // counter the last step cannot be told apart: its update is computed and never read)
    if (ctc && (counter || step + 1 < cfg->max_steps)) {
      if (side) {
        SBK_HIP(hipEventRecord(side->fork, st));
        SBK_HIP(hipStreamWaitEvent(cst, side->fork, 0));
      }
      SBK_TRY(sbk::ctc_advance(bb.ctc_x, bb.phi[cur], bb.psi, bb.s.parent, bb.s.tokens[cur ^ 1], bb.s.tokens[cur],
                               bb.phi[cur ^ 1], bb.psi_prev[cur ^ 1], n, T, V, beam, step, cfg->blank, cst, win, window, Vp));
      if (psi_aside)
        SBK_TRY(sbk::ctc_psi_step(bb.ctc_x, bb.phi[cur ^ 1], bb.s.tokens[cur ^ 1], enc_len, bb.psi, B, T, V, beam,
                                  step + 1, cfg->blank, cfg->eos, pst, nullptr, 0, Vp));
      if (side) SBK_HIP(hipEventRecord(side->join, cst));
    }
    if (counter) {
      SBK_LAUNCH(step_inc_kernel, dim3(1), dim3(1), 0, st, step_dev);
      SBK_TRY(sbk::launch_status("step_inc"));
    }
    return 0;
  };
  const bool polling = host_flag && cfg->check_every > 0 && g_poll.begin();
  auto poll_full = [&](int steps_done) -> int {  // 1: every utterance has its beam of finished hypotheses
    if (!(polling && (steps_done % cfg->check_every == 0) && steps_done < cfg->max_steps)) return 0;
    return g_poll.poll(bb.s.n_full, B, st);
  };

  int cur = 0, steps = 0;
  // graph_mode 1: two consecutive steps (the double-buffered tables flip back after two) are captured once
  // into a hipGraph and replayed; 2: the same device-side step counter with plain launches (tests, fallback).
  int graph_mode = cfg->graph_mode;
  if (graph_mode && (side || sbk::prof_enabled() || T > 900 || cfg->max_steps < 2 || pos_off > 0 || cfg->first_bias || window > 0)) graph_mode = 0;
  if (graph_mode && sbk::persist_eligible(W, n, B, beam, Lmax)) graph_mode = 0;  // (a cooperative launch is not captured)
  if (graph_mode) {
    struct Scope {  // the step source is per host thread; never leave it set
      ~Scope() {
        sbk::g_step_ptr = nullptr;
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
      }
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
    } scope;
    SBK_HIP(hipMemsetAsync(step_dev, 0, sizeof(int32_t), st));
    sbk::g_step_ptr = step_dev;
    sbk::g_step_min_steps = cfg->min_steps;
    if (graph_mode == 1) {
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        int rc = issue_step(0, 0, true);
        if (!rc) rc = issue_step(0, 1, true);
        const hipError_t e = hipStreamEndCapture(st, &scope.graph);
        if (rc) return rc;
        if (e != hipSuccess || hipGraphInstantiate(&scope.exec, scope.graph, nullptr, nullptr, 0) != hipSuccess) {
          scope.exec = nullptr;  // replay unavailable: plain launches below
          (void)hipGetLastError();
        }
      } else {
        (void)hipGetLastError();  // e.g. the legacy default stream cannot capture: plain launches below
      }
    }
    const int pairs = cfg->max_steps / 2;
    bool done = false;
    for (int p = 0; p < pairs && !done; ++p) {
      if (scope.exec) {
        SBK_HIP(hipGraphLaunch(scope.exec, st));
      } else {
        SBK_TRY(issue_step(0, 0, true));
        SBK_TRY(issue_step(0, 1, true));
      }
      steps += 2;
      const int r = poll_full(steps);  // (running a step past the stop point cannot change the result)
      if (r < 0) return sbk::fail(1, "beam_search: stop-rule poll failed");
      done = r == 1;
    }
    if (!done && (cfg->max_steps & 1)) {
      SBK_TRY(issue_step(0, 0, true));
      ++steps;
    }
    cur = steps & 1;
  } else {
    for (int step = 0; step < cfg->max_steps; ++step) {
      SBK_TRY(issue_step(step, cur, false));
      cur ^= 1;
      steps = step + 1;
      const int r = poll_full(steps);
      if (r < 0) return sbk::fail(1, "beam_search: stop-rule poll failed");
      if (r == 1) break;
    }
  }
  if (side) SBK_HIP(hipStreamWaitEvent(st, side->join, 0));  // nothing of this call outlives it on the helper stream
  if (steps_run) *steps_run = steps;
  SBK_LAUNCH(beam_finalize_kernel, dim3(B), dim3(256), 0, st, bb.s, cur, steps, beam, Lmax, Lmax - pos_off, topk, out_tokens, out_len,
             out_score, out_logp, cfg->utt_max_steps, out_longest);
  SBK_TRY(sbk::launch_status("beam_finalize"));
  if (out_max_len)
    SBK_HIP(hipMemcpyAsync(out_max_len, bb.s.n_full + 1, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return 0;
}

// Teacher-forced run of the KV-cached decoder over a given prefix: the TransformerASR.decode
// surface (TransformerASR.py:426-473).  tokens [n,L] -> pred [n,L,d] (final LayerNorm output).
extern "C" size_t sbk_decoder_prefix_workspace_bytes(const sbk_decoder_weights* W, int n, int T, int L) {
  if (!W) return 0;
  Carver c{nullptr, 0, true};
  DecoderBufs d;
  carve_decoder(c, d, W, n, n, T, L);
  c.take<int32_t>((size_t)n * L);  // kv_slot
  c.take<int32_t>((size_t)n);      // tok column
  c.take<int32_t>(64 + 2 * (size_t)n);
  return c.used + 256;
}

__global__ void gather_col_kernel(const int32_t* __restrict__ tokens, int32_t* __restrict__ col, int n, int L, int p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) col[i] = tokens[(size_t)i * L + p];
}

extern "C" int sbk_decoder_prefix_f32(const sbk_decoder_weights* W, const int32_t* tokens, const float* enc,
                                      const int32_t* enc_len, void* workspace, size_t workspace_bytes, float* pred,
                                      int n, int T, int L, sbk_stream_t stream) {
  SBK_TRY(check_weights(W));
  if (n == 0 || L == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(tokens && enc && enc_len && workspace && pred, "decoder_prefix: null");
  SBK_REQUIRE(L <= W->max_len, "decoder_prefix: prefix longer than the positional table");
  SBK_REQUIRE(workspace_bytes >= sbk_decoder_prefix_workspace_bytes(W, n, T, L), "decoder_prefix: workspace too small");
  hipStream_t st = sbk::as_stream(stream);
  if (n == 0 || L == 0) return 0;
  Carver c{static_cast<char*>(workspace), 0, false};
  DecoderBufs d;
  carve_decoder(c, d, W, n, n, T, L);
  int32_t* kv_slot = c.take<int32_t>((size_t)n * L);
  int32_t* col = c.take<int32_t>((size_t)n);
  int32_t* misc = c.take<int32_t>(64 + 2 * (size_t)n);
  SBK_TRY(project_memory(W, d, enc, n, T, st));
  SBK_LAUNCH(greedy_init_kernel, dim3(sbk::cdiv(n, 256)), dim3(256), 0, st, misc + 64, misc + 64 + n, misc, kv_slot, n, L, 0);
  SBK_TRY(sbk::launch_status("prefix_init"));
  for (int p = 0; p < L; ++p) {
    SBK_LAUNCH(gather_col_kernel, dim3(sbk::cdiv(n, 256)), dim3(256), 0, st, tokens, col, n, L, p);
    SBK_TRY(decoder_step(W, d, col, kv_slot, enc_len, p, n, n, T, 1, L, false, st));
    SBK_LAUNCH(copy_rows_kernel, dim3(n), dim3(256), 0, st, (const float*)d.h, pred + (size_t)p * W->d_model, n,
               W->d_model, (long)L * W->d_model);
    SBK_TRY(sbk::launch_status("prefix_copy"));
  }
  return 0;
}

// TransformerLM.forward (TransformerLM.py:116-158) through the KV-cached step.
extern "C" size_t sbk_lm_prefix_workspace_bytes(const sbk_lm_weights* LM, int n, int L) {
  if (!LM) return 0;
  Carver c{nullptr, 0, true};
  LmBufs d;
  carve_lm(c, d, LM, n, L > 0 ? L : 1);
  c.take<int32_t>((size_t)n * (L > 0 ? L : 1));
  c.take<int32_t>((size_t)n);
  c.take<int32_t>(64 + 2 * (size_t)n);
  return c.used + 256;
}

extern "C" int sbk_lm_prefix_f32(const sbk_lm_weights* LM, const int32_t* tokens, void* workspace,
                                 size_t workspace_bytes, float* logits, int n, int L, sbk_stream_t stream) {
  SBK_TRY(check_lm(LM));
  if (n == 0 || L == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(tokens && workspace && logits, "lm_prefix: null");
  SBK_REQUIRE(L <= LM->max_len, "lm_prefix: prefix longer than the positional table");
  SBK_REQUIRE(workspace_bytes >= sbk_lm_prefix_workspace_bytes(LM, n, L), "lm_prefix: workspace too small");
  hipStream_t st = sbk::as_stream(stream);
  if (n == 0 || L == 0) return 0;
  Carver c{static_cast<char*>(workspace), 0, false};
  LmBufs d;
  carve_lm(c, d, LM, n, L);
  int32_t* kv_slot = c.take<int32_t>((size_t)n * L);
  int32_t* col = c.take<int32_t>((size_t)n);
  int32_t* misc = c.take<int32_t>(64 + 2 * (size_t)n);
  SBK_LAUNCH(greedy_init_kernel, dim3(sbk::cdiv(n, 256)), dim3(256), 0, st, misc + 64, misc + 64 + n, misc, kv_slot, n, L, 0);
  SBK_TRY(sbk::launch_status("lm_prefix_init"));
  for (int p = 0; p < L; ++p) {
    SBK_LAUNCH(gather_col_kernel, dim3(sbk::cdiv(n, 256)), dim3(256), 0, st, tokens, col, n, L, p);
    SBK_TRY(lm_step(LM, d, col, kv_slot, tokens, L, 0, 0, p, n, L, st));
    SBK_LAUNCH(copy_rows_kernel, dim3(n), dim3(256), 0, st, (const float*)d.logits, logits + (size_t)p * LM->vocab, n,
               LM->vocab, (long)L * LM->vocab);
    SBK_TRY(sbk::launch_status("lm_prefix_copy"));
  }
  return 0;
}

// S2STransformerGreedySearcher (seq2seq.py:176-367, temperature 0).
extern "C" size_t sbk_greedy_search_workspace_bytes(const sbk_decoder_weights* W, int B, int T, int max_steps) {
  if (!W) return 0;
  Carver c{nullptr, 0, true};
  DecoderBufs d;
  const int L = max_steps > 0 ? max_steps : 1;
  carve_decoder(c, d, W, B, B, T, L);
  c.take<int32_t>((size_t)B * L);
  c.take<int32_t>(64 + 2 * (size_t)B);
  return c.used + 256;
}

extern "C" int sbk_greedy_search_f32(const sbk_decoder_weights* W, const float* enc, const int32_t* enc_len,
                                     void* workspace, size_t workspace_bytes, int32_t* out_tokens, float* out_scores,
                                     int32_t* host_flag, int32_t* steps_run, int B, int T, int min_steps, int max_steps,
                                     int bos, int eos, int check_every, sbk_stream_t stream) {
  SBK_TRY(check_weights(W));
  if (B == 0) {  // empty batch: nothing to launch, the data pointers may be NULL
    if (steps_run) *steps_run = 0;
    return 0;
  }
  SBK_REQUIRE(enc && enc_len && workspace && out_tokens && out_scores && W->seq_w && W->seq_b, "greedy_search: null");
  SBK_REQUIRE(max_steps <= W->max_len, "greedy_search: too many steps for the positional table");
  SBK_REQUIRE(workspace_bytes >= sbk_greedy_search_workspace_bytes(W, B, T, max_steps), "greedy: workspace too small");
  hipStream_t st = sbk::as_stream(stream);
  if (steps_run) *steps_run = 0;
  const int L = max_steps > 0 ? max_steps : 1;
  if (B == 0) return 0;
  Carver c{static_cast<char*>(workspace), 0, false};
  DecoderBufs d;
  carve_decoder(c, d, W, B, B, T, L);
  int32_t* kv_slot = c.take<int32_t>((size_t)B * L);
  int32_t* misc = c.take<int32_t>(64 + 2 * (size_t)B);
  int32_t *n_ended = misc, *tok = misc + 64, *ended = misc + 64 + B;
  SBK_TRY(project_memory(W, d, enc, B, T, st));
  SBK_LAUNCH(greedy_init_kernel, dim3(sbk::cdiv(B, 256)), dim3(256), 0, st, tok, ended, n_ended, kv_slot, B, L, bos);
  SBK_TRY(sbk::launch_status("greedy_init"));
  SBK_HIP(hipMemsetAsync(out_tokens, 0, (size_t)B * L * sizeof(int32_t), st));
  SBK_HIP(hipMemsetAsync(out_scores, 0, (size_t)B * L * sizeof(float), st));
  int k = 0;
  const bool polling = host_flag && check_every > 0 && g_poll.begin();
  for (int step = min_steps; step < max_steps; ++step, ++k) {  // positions count from 0 (seq2seq.py:227)
    SBK_TRY(decoder_step(W, d, tok, kv_slot, enc_len, k, B, B, T, 1, L, true, st));
    SBK_LAUNCH(greedy_pick_kernel, dim3(B), dim3(256), 0, st, (const float*)d.logits, tok, ended, out_tokens,
               out_scores, n_ended, W->vocab, k, L, eos, (const float*)nullptr, (const float*)nullptr);
    SBK_TRY(sbk::launch_status("greedy_pick"));
    if (polling && ((k + 1) % check_every == 0)) {
      const int r = g_poll.poll(n_ended, B, st);
      if (r < 0) return sbk::fail(1, "greedy_search: stop-rule poll failed");
      if (r == 1) {
        ++k;
        break;
      }
    }
  }
  if (steps_run) *steps_run = k;
  return 0;
}

// S2SWhisperGreedySearcher.forward (seq2seq.py:176-327 + :421-636): a token prompt common in length to the batch
// primes the KV cache, then arg-max decoding with additive logit masks.
extern "C" size_t sbk_prompted_greedy_search_workspace_bytes(const sbk_decoder_weights* W, int B, int T, int P,
                                                             int max_new) {
  if (!W) return 0;
  Carver c{nullptr, 0, true};
  DecoderBufs d;
  const int L = (P > 0 ? P - 1 : 0) + (max_new > 0 ? max_new : 1);
  carve_decoder(c, d, W, B, B, T, L);
  c.take<int32_t>((size_t)B * L);
  c.take<int32_t>((size_t)B);
  c.take<int32_t>(64 + 2 * (size_t)B);
  return c.used + 256;
}

extern "C" int sbk_prompted_greedy_search_f32(const sbk_decoder_weights* W, const float* enc, const int32_t* enc_len,
                                              const int32_t* prompt, int P, const float* logit_bias,
                                              const float* first_bias, void* workspace, size_t workspace_bytes,
                                              int32_t* out_tokens, float* out_scores, int probe_pos, int probe_token,
                                              float* out_probe, int32_t* host_flag, int32_t* steps_run, int B, int T,
                                              int max_new, int eos, int check_every, sbk_stream_t stream) {
  SBK_TRY(check_weights(W));
  if (steps_run) *steps_run = 0;
  if (B == 0 || max_new <= 0) return 0;  // empty batch / nothing to sample: nothing to launch
  SBK_REQUIRE(enc && enc_len && prompt && workspace && out_tokens && out_scores && W->seq_w && W->seq_b,
              "prompted_greedy_search: null");
  SBK_REQUIRE(P >= 1, "prompted_greedy_search: the prompt holds at least the first decoder input token");
  const int L = P - 1 + max_new;
  SBK_REQUIRE(L <= W->max_len, "prompted_greedy_search: prompt + new tokens (%d) exceed the positional table (%d)", L,
              W->max_len);
  SBK_REQUIRE(!out_probe || (probe_pos >= 0 && probe_pos < P && probe_token >= 0 && probe_token < W->vocab),
              "prompted_greedy_search: probe position / token out of range");
  SBK_REQUIRE(workspace_bytes >= sbk_prompted_greedy_search_workspace_bytes(W, B, T, P, max_new),
              "prompted_greedy_search: workspace too small");
  hipStream_t st = sbk::as_stream(stream);
  Carver c{static_cast<char*>(workspace), 0, false};
  DecoderBufs d;
  carve_decoder(c, d, W, B, B, T, L);
  int32_t* kv_slot = c.take<int32_t>((size_t)B * L);
  int32_t* col = c.take<int32_t>((size_t)B);
  int32_t* misc = c.take<int32_t>(64 + 2 * (size_t)B);
  int32_t *n_ended = misc, *tok = misc + 64, *ended = misc + 64 + B;
  SBK_TRY(project_memory(W, d, enc, B, T, st));
  SBK_LAUNCH(greedy_init_kernel, dim3(sbk::cdiv(B, 256)), dim3(256), 0, st, tok, ended, n_ended, kv_slot, B, L, 0);
  SBK_TRY(sbk::launch_status("greedy_init"));
  SBK_HIP(hipMemsetAsync(out_tokens, 0, (size_t)B * max_new * sizeof(int32_t), st));
  SBK_HIP(hipMemsetAsync(out_scores, 0, (size_t)B * max_new * sizeof(float), st));
  for (int p = 0; p + 1 < P; ++p) {  // the prompt but its last token: K/V only (logits where the probe sits)
    const bool probe = out_probe && p == probe_pos;
    SBK_LAUNCH(gather_col_kernel, dim3(sbk::cdiv(B, 256)), dim3(256), 0, st, prompt, col, B, P, p);
    SBK_TRY(decoder_step(W, d, col, kv_slot, enc_len, p, B, B, T, 1, L, probe, st));
    if (probe) SBK_LAUNCH(softmax_prob_kernel, dim3(B), dim3(256), 0, st, (const float*)d.logits, out_probe, W->vocab, probe_token);
  }
  SBK_LAUNCH(gather_col_kernel, dim3(sbk::cdiv(B, 256)), dim3(256), 0, st, prompt, tok, B, P, P - 1);
  SBK_TRY(sbk::launch_status("prompt"));
  int k = 0;
  const bool polling = host_flag && check_every > 0 && g_poll.begin();
  for (; k < max_new; ++k) {
    SBK_TRY(decoder_step(W, d, tok, kv_slot, enc_len, P - 1 + k, B, B, T, 1, L, true, st));
    if (k == 0 && out_probe && probe_pos == P - 1)
      SBK_LAUNCH(softmax_prob_kernel, dim3(B), dim3(256), 0, st, (const float*)d.logits, out_probe, W->vocab, probe_token);
    SBK_LAUNCH(greedy_pick_kernel, dim3(B), dim3(256), 0, st, (const float*)d.logits, tok, ended, out_tokens, out_scores,
               n_ended, W->vocab, k, max_new, eos, logit_bias, k == 0 ? first_bias : (const float*)nullptr);
    SBK_TRY(sbk::launch_status("greedy_pick"));
    if (polling && ((k + 1) % check_every == 0)) {
      const int r = g_poll.poll(n_ended, B, st);
      if (r < 0) return sbk::fail(1, "greedy_search: stop-rule poll failed");
      if (r == 1) {
        ++k;
        break;
      }
    }
  }
  if (steps_run) *steps_run = k < max_new ? k : max_new;
  return 0;
}
