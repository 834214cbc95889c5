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
// instead of two per position.  (Round 5 measured the whole prefix requested at once -- the K and V rows of 16 / 32 / 64
// positions in flight per wave before the first score, unconditional loads on clamped positions, bit-identical results: 41.7 /
// 55.5 / 73.2 us per launch against 43.3 at 1 280 rows and prefixes of 1 .. 60 tokens, profiles/r05_p_*: the kernel is bound
// by the number of row requests it issues, not by their latency -- every surplus (clamped) request and every wave lost to
// the 112 / 188 registers costs more than the shorter chain returns.  Removed.)
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

// ---- self-attention of the new token over SHARED ANCESTRY (head_dim 64, 2 .. 16 beams per utterance, no key mask): a WAVE = one
// (utterance, head) and ALL of its beams.  The beams of an utterance descend from few common ancestors: at position p their `beam`
// ancestry slots kv_slot[.][p] name 1 .. beam DISTINCT cache rows (one or two for all but the newest positions), and the kernel
// above asks for every one of them once per beam (5.2 MB of row requests per prefix token and launch at 1 280 hypotheses:
// it is bound by the number of requests it issues).  Here the wave first lists the distinct (slot, position)
// rows of its utterance -- a lane per position compares the beams' slots, a wave scan packs the list into LDS together with a
// 16-bit mask of the beams that descend from each row; the new token's own K / V rows (one per beam, read from qkv) close the
// list --, then walks the list exactly like cross_attn_ring_kernel walks a run of memory frames: tiles of 16 rows straight into
// MFMA operand registers D tiles deep, transposed scores S^T[row][beam] = K Q^T and context O^T += V^T P^T on
// v_mfma_f32_16x16x4_f32, online softmax -- a (row, beam) pair whose beam does not descend from the row is masked to -inf
// (probability exactly 0), so every beam's softmax runs over exactly its own prefix.  Each distinct row is fetched ONCE.
// The wave also appends the new token's K / V head slices to the cache (slot = hypothesis index), as the kernel above does.
// A WORKGROUP = one (utterance, head); its four waves take a quarter of the prefix positions each (their own list, their own walk:
// nothing shared until the end) and the four partial (context, max, sum) triples are merged through LDS by wave 0 -- the
// flash-decoding split of cross_attn_ring_kernel inside a workgroup.  (Round 6 first built it as one wave per (utterance, head):
// a tenth of the row requests, but a 3-4 x longer dependent chain per wave -- 55-60 against 36-43 us per launch on one stream,
// +1-2 % under the eight workers; profiles/r06_d_*.)
template <int D>
__global__ void __launch_bounds__(256) self_attn_anc_kernel(SelfAttnArgs a, int cap) {
  SBK_DYN_LDS(float, lds);  // [4 waves][cap] x int2 {packed row, beam mask} | [4 waves][64 lanes][18] partials
  if (a.step_ptr) a.step = a.step_ptr[0];
  const int lane = threadIdx.x & 63, wave = sbk::uniform(threadIdx.x >> 6);
  const int beam = a.group, gw = blockIdx.x;  // (utterance, head)
  const int u = gw / a.H, h = gw - u * a.H;
  const int d = a.d, step = a.step, Lmax = a.Lmax;
  int2* rows = reinterpret_cast<int2*>(lds) + (size_t)wave * cap;
  float* part = lds + (size_t)8 * cap;  // behind the four lists (2 floats per record)
  const int col = lane & 15, g = lane >> 4;
  const size_t hoff = (size_t)h * 64;
  // the new token's K / V head slices -> cache rows [hypothesis][step] (beam j by wave j mod 4)
  for (int j = wave; j < beam; j += 4) {
    const size_t i = (size_t)u * beam + j;
    const float* src = a.qkv + i * 3 * d + d + hoff;
    const size_t o = (i * Lmax + step) * d + hoff + lane;
    a.kcache[o] = src[lane];
    a.vcache[o] = src[d + lane];
  }
  // this wave's positions [p_lo, p_hi): a quarter of the prefix, in whole lanes' worth of 16
  const int per = ((step + 63) / 64) * 16;
  const int p_lo = min(wave * per, step), p_hi = min(p_lo + per, step);
  // distinct (slot, position) rows of the utterance's prefixes over these positions, in position order
  int nrows = 0;
  for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
    const int p = p0 + lane;
    const bool act = p < p_hi;
    int sl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) sl[j] = (act && j < beam) ? a.kv_slot[((size_t)u * beam + j) * Lmax + p] : -1 - j;
    unsigned msk[16];
    unsigned first = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      unsigned m = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) m |= (sl[k] == sl[j]) ? (1u << k) : 0u;
      msk[j] = m;
      if (act && j < beam && (m & ((1u << j) - 1u)) == 0u) first |= 1u << j;  // no earlier beam names the same row
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) cnt += (first >> j) & 1;
    int incl = cnt;  // inclusive scan over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = sbk::shfl(incl, lane >= off ? lane - off : lane);
      if (lane >= off) incl += t;
    }
    int w = nrows + incl - cnt;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if ((first >> j) & 1) rows[w++] = make_int2((sl[j] << 12) | p, (int)msk[j]);
    nrows += sbk::shfl(incl, 63);
  }
  // the new token itself (qkv row of beam j): beam j by wave j mod 4, like the cache append
  {
    const int mine = beam > wave ? (beam - wave + 3) / 4 : 0;
    if (lane < mine) rows[nrows + lane] = make_int2(((wave + 4 * lane) << 12) | step, 1 << (wave + 4 * lane));
    nrows += mine;
  }
  sbk::wave_sync();
  // row record -> address of the row's K head slice (V: the same offset in vcache, or + d inside qkv)
  auto kaddr = [&](int rec, const float*& kp, const float*& vp) SBK_INLINE_LAMBDA {
    const int pos = rec & 4095, slot = rec >> 12;
    if (pos == step) {
      kp = a.qkv + ((size_t)u * beam + slot) * 3 * d + d + hoff;
      vp = kp + d;
    } else {
      const size_t o = ((size_t)slot * Lmax + pos) * d + hoff;
      kp = a.kcache + o;
      vp = a.vcache + o;
    }
  };
  sbk::f32x4 o[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[ct][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  if (nrows > 0) {  // (uniform; a wave without rows -- a prefix shorter than four positions -- hands over an empty partial)
    float qf[16];
    {
      const float* qp = a.qkv + ((size_t)u * beam + min(col, beam - 1)) * 3 * d + hoff + 4 * g;
      const float qs = col < beam ? a.scale : 0.0f;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 16 * jq);
        qf[4 * jq] = t.x * qs, qf[4 * jq + 1] = t.y * qs, qf[4 * jq + 2] = t.z * qs, qf[4 * jq + 3] = t.w * qs;
      }
    }
    const int ntiles = (nrows + 15) / 16;
    float4 kr[D][4], vr[D][4];
    int vm[D][4];  // beam masks of the rows 4 g + i of the tile (0 past the list)
    auto fetch = [&](float4(&kk)[4], float4(&vv)[4], int(&mm)[4], int tile) SBK_INLINE_LAMBDA {
      const float *kp, *vp;
      kaddr(rows[min(tile * 16 + col, nrows - 1)].x, kp, vp);
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) kk[jq] = *reinterpret_cast<const float4*>(kp + 4 * g + 16 * jq);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = tile * 16 + 4 * g + i;
        const int2 rec = rows[min(r, nrows - 1)];
        kaddr(rec.x, kp, vp);
        vv[i] = *reinterpret_cast<const float4*>(vp + 4 * col);
        mm[i] = r < nrows ? rec.y : 0;
      }
    };
    auto consume = [&](const float4(&kk)[4], const float4(&vv)[4], const int(&mm)[4]) SBK_INLINE_LAMBDA {
      sbk::f32x4 sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[r] = 0.0f;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        sc = sbk::mfma_16x16x4(kk[jq].x, qf[4 * jq], sc);
        sc = sbk::mfma_16x16x4(kk[jq].y, qf[4 * jq + 1], sc);
        sc = sbk::mfma_16x16x4(kk[jq].z, qf[4 * jq + 2], sc);
        sc = sbk::mfma_16x16x4(kk[jq].w, qf[4 * jq + 3], sc);
      }
      float mt = -INFINITY;  // sc[r] = score of (row 4 g + r of the tile, beam col)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!((mm[r] >> col) & 1)) sc[r] = -INFINITY;
        mt = fmaxf(mt, sc[r]);
      }
      mt = fmaxf(mt, sbk::shfl_xor(mt, 16));
      mt = fmaxf(mt, sbk::shfl_xor(mt, 32));
      const float m_new = fmaxf(m_run, mt);  // -inf while the beam has met none of its rows (a tile of other beams' rows)
      const float alpha = m_new == -INFINITY ? 1.0f : expf(m_run - m_new);
      float pr[4], ps = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = sc[r] == -INFINITY ? 0.0f : expf(sc[r] - m_new);
        ps += pr[r];
      }
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[ct][r] *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[0] = sbk::mfma_16x16x4(vv[i].x, pr[i], o[0]);
        o[1] = sbk::mfma_16x16x4(vv[i].y, pr[i], o[1]);
        o[2] = sbk::mfma_16x16x4(vv[i].z, pr[i], o[2]);
        o[3] = sbk::mfma_16x16x4(vv[i].w, pr[i], o[3]);
      }
    };
#pragma unroll
    for (int s = 0; s < D - 1; ++s) fetch(kr[s], vr[s], vm[s], s);  // (tiles past the list re-read its last row, masks 0)
    int k0 = 0;
    for (; k0 + D <= ntiles; k0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        fetch(kr[(s + D - 1) % D], vr[(s + D - 1) % D], vm[(s + D - 1) % D], k0 + s + D - 1);
        sbk::sched_fence();
        consume(kr[s], vr[s], vm[s]);
        sbk::sched_fence();
      }
    }
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
      if (k0 + s < ntiles) consume(kr[s], vr[s], vm[s]);
  }
  float l_tot = l_run + sbk::shfl_xor(l_run, 16);
  l_tot += sbk::shfl_xor(l_tot, 32);
  // hand the partial over: [wave][lane][16 context values | max | sum]
  {
    float* pp = part + ((size_t)wave * 64 + lane) * 18;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[4 * ct + r] = o[ct][r];
    pp[16] = m_run;
    pp[17] = l_tot;
  }
  __syncthreads();
  if (wave == 0 && col < beam) {  // (every beam owns a new-token row, so the merged maximum is finite)
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, part[((size_t)w * 64 + lane) * 18 + 16]);
    float acc[16], den = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {  // fixed order: the result does not depend on which wave finished first
      const float* pp = part + ((size_t)w * 64 + lane) * 18;
      const float mw = pp[16];
      const float sc = mw == -INFINITY ? 0.0f : expf(mw - M);
      den = fmaf(sc, pp[17], den);
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = fmaf(sc, pp[e], acc[e]);
    }
    float* op = a.out + ((size_t)u * beam + col) * d + hoff + 16 * g;
    const float inv = 1.0f / den;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(op + 4 * r) = make_float4(acc[r] * inv, acc[4 + r] * inv, acc[8 + r] * inv, acc[12 + r] * inv);
  }
}

// ---------------------------------------------------------------- cross attention, all beams of an utterance
constexpr int kQT = 16;   // queries (beams) served per workgroup
constexpr int kFC = 128;  // memory frames per workgroup (flash-decoding style split of the memory)

struct CrossAttnArgs {
  const float* q;         // [n,d]   n = B*beam, hypothesis i belongs to utterance i / beam
  const float* kv;        // [B,T,2d] per frame K (d) then V (d): the projection GEMM's output (a head-major copy
                          // [B,H,T,2*Dh] was knob 5 of rounds 1-4: within 5 % either way, removed with its re-layout pass)
  const int32_t* enc_len; // [B]
  float* out;             // [n,d]
  float* part;            // [B,H,NS,kQT,DH+2] partial (context, max, sum) when NS > 1
  int B, T, d, H, Dh, beam, NS;
  float scale;
  int fc;  // memory frames per workgroup of the frame-per-thread kernel (64, 128 or 256)
};

// element offsets of (utterance b, head h): base of frame 0, frame stride, K -> V distance
struct KvView {
  size_t base;
  int row, voff;
};
__device__ __forceinline__ KvView kv_view(const CrossAttnArgs& a, int b, int h, int DH) {
  return {(size_t)b * a.T * 2 * a.d + (size_t)h * DH, 2 * a.d, a.d};
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

// ---- cross-attention step on a REGISTER ring and the matrix cores (head_dim 64, beam <= 16, row-major K/V): a WAVE = one
// (utterance, head) x one run of frames -- no LDS, no barrier, nothing shared between the waves of a workgroup.  Each wave streams
// its own 256-byte head segments of the K|V rows straight into MFMA operand registers, D tiles of 16 frames deep (D x 8 KB per
// wave, 4 waves per CU), transposed scores and context on v_mfma_f32_16x16x4_f32 (beams padded to 16), online softmax:
//   scores  S^T[frame][beam] = K Q^T:  A[m = frame][k slot g] of MFMA (jq, e) = K[frame][16 jq + 4 g + e] -- the lane's float4
//           load jq covers 64 contiguous bytes of each of the tile's 16 rows across g; in the result a lane owns one beam and the
//           four frames 4 g .. 4 g + 3, so the softmax statistics need two cross-lane steps and the rescale is one scalar per lane;
//   context O^T[channel][beam] += V^T P^T:  k slot g of MFMA (ct, i) <-> frame 4 g + i (the frames whose probabilities the lane
//           holds: P IS the B operand), A[m][g] = V[frame 4g + i][4 m + ct] -- the lane's float4 load i covers the 256 contiguous
//           bytes of four rows; result row m of block ct is channel 4 m + ct, so a lane ends with the 16 CONSECUTIVE channels
//           16 g .. 16 g + 15 of its beam (four float4 stores).
// Frames past the run re-read the utterance's last row and are masked to -inf before the softmax (probability exactly 0).
// History: round 3's LDS-DMA kernel (a workgroup = one utterance, wave = head, a double-buffered 16-frame
// K|V image in LDS) kept ONE 64 KB tile in flight per CU behind a workgroup barrier -- two fp32 tiles are all that fit in 160 KB --
// and its tile period was the latency of one burst (3.4 us: 92 us per launch at 128 utterances x 430 frames, 2.4 TB/s); the ring
// has 2 x 8 KB x 4 waves in flight per CU with no barrier: 46 us, 4.8 TB/s (profiles/r05_m_*, r05_n_*).  Partial (context, max,
// sum) per run of frames -> cross_merge_kernel when an utterance is split into runs.
template <int D, bool KNT, bool VNT>  // KNT / VNT: non-temporal loads of the K / V tiles (knob 53, bits 1 / 2)
__global__ void __launch_bounds__(256) cross_attn_ring_kernel(CrossAttnArgs a, int chunk) {
  const int lane = threadIdx.x & 63, wave = sbk::uniform(threadIdx.x >> 6);
  const int split = blockIdx.x, gw = sbk::uniform(blockIdx.y * 4 + wave);
  if (gw >= a.B * a.H) return;  // (no barrier anywhere below)
  const int b = gw / a.H, h = gw - b * a.H;
  const int ROW = 2 * a.d, nq = a.beam, col = lane & 15, g = lane >> 4;
  const int klen = min(max(a.enc_len[b], 1), a.T);
  const int t0 = split * chunk, t1 = min(klen, t0 + chunk);
  float* pp = a.part ? a.part + ((((size_t)b * a.H + h) * a.NS + split) * nq + col) * (64 + 2) : nullptr;
  if (t0 >= t1) {
    if (a.NS > 1 && g == 0 && col < nq) {
      pp[64] = -INFINITY;
      pp[65] = 0.0f;
    }
    return;
  }
  float qf[16];
  {  // unconditional loads of a clamped row (a load under a lane mask is a branch + a full wait each: 16 round trips in a row)
    const float* qp = a.q + ((size_t)b * nq + min(col, nq - 1)) * a.d + h * 64 + 4 * g;
    const float qs = col < nq ? a.scale : 0.0f;
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 16 * jq);
      qf[4 * jq] = t.x * qs;
      qf[4 * jq + 1] = t.y * qs;
      qf[4 * jq + 2] = t.z * qs;
      qf[4 * jq + 3] = t.w * qs;
    }
  }
  const float* kb = a.kv + (size_t)b * a.T * ROW + h * 64 + 4 * g;  // + row * ROW + 16 jq
  const float* vb = a.kv + (size_t)b * a.T * ROW + a.d + h * 64 + 4 * col;  // + row * ROW
  const int ntiles = (t1 - t0 + 15) / 16;
  float4 kr[D][4], vr[D][4];
  auto fetch = [&](float4(&kk)[4], float4(&vv)[4], int tile) SBK_INLINE_LAMBDA {
    // unconditional (tiles past the run fetch the utterance's last row again: cache hits) -- straight-line code between the
    // fetch of tile k + D - 1 and the use of tile k is what lets the compiler count the loads in between (s_waitcnt vmcnt(8 (D - 1)
    // + ...)); behind a branch it falls back to waiting for everything in flight
    const float* kp = kb + (size_t)min(t0 + tile * 16 + col, klen - 1) * ROW;
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) kk[jq] = sbk::ld16<KNT>(kp + 16 * jq);
#pragma unroll
    for (int i = 0; i < 4; ++i) vv[i] = sbk::ld16<VNT>(vb + (size_t)min(t0 + tile * 16 + 4 * g + i, klen - 1) * ROW);
  };
  sbk::f32x4 o[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[ct][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  auto consume = [&](const float4(&kk)[4], const float4(&vv)[4], int tile) SBK_INLINE_LAMBDA {
    sbk::f32x4 sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[r] = 0.0f;
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      sc = sbk::mfma_16x16x4(kk[jq].x, qf[4 * jq], sc);
      sc = sbk::mfma_16x16x4(kk[jq].y, qf[4 * jq + 1], sc);
      sc = sbk::mfma_16x16x4(kk[jq].z, qf[4 * jq + 2], sc);
      sc = sbk::mfma_16x16x4(kk[jq].w, qf[4 * jq + 3], sc);
    }
    const int fb = t0 + tile * 16 + 4 * g;  // sc[r] = score of (frame fb + r, beam col)
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (fb + r >= t1) sc[r] = -INFINITY;
      mt = fmaxf(mt, sc[r]);
    }
    mt = fmaxf(mt, sbk::shfl_xor(mt, 16));
    mt = fmaxf(mt, sbk::shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);  // finite: the tile's first frame is inside the run
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[0] = sbk::mfma_16x16x4(vv[i].x, p[i], o[0]);
      o[1] = sbk::mfma_16x16x4(vv[i].y, p[i], o[1]);
      o[2] = sbk::mfma_16x16x4(vv[i].z, p[i], o[2]);
      o[3] = sbk::mfma_16x16x4(vv[i].w, p[i], o[3]);
    }
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) fetch(kr[s], vr[s], s);
  int k0 = 0;
  for (; k0 + D <= ntiles; k0 += D) {  // whole rounds of the ring: straight-line code
#pragma unroll
    for (int s = 0; s < D; ++s) {
      fetch(kr[(s + D - 1) % D], vr[(s + D - 1) % D], k0 + s + D - 1);
      sbk::sched_fence();
      consume(kr[s], vr[s], k0 + s);
      sbk::sched_fence();
    }
  }
#pragma unroll
  for (int s = 0; s < D - 1; ++s)  // the last < D tiles are in flight already (slots 0 ..)
    if (k0 + s < ntiles) consume(kr[s], vr[s], k0 + s);
  float l_tot = l_run + sbk::shfl_xor(l_run, 16);
  l_tot += sbk::shfl_xor(l_tot, 32);
  if (col < nq) {
    if (a.NS == 1) {
      float* op = a.out + ((size_t)b * nq + col) * a.d + h * 64 + 16 * g;
      const float inv = 1.0f / l_tot;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(op + 4 * r) = make_float4(o[0][r] * inv, o[1][r] * inv, o[2][r] * inv, o[3][r] * inv);
    } else {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[16 * g + 4 * r + ct] = o[ct][r];
      if (g == 0) {
        pp[64] = m_run;
        pp[65] = l_tot;
      }
    }
  }
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
  // profiler names: "cross_attn_ring" = the register-ring kernel, "cross_attn_step" = the frame-per-thread kernel (a test or the
  // bench can tell from the report which one a search ran), "cross_merge" = the merge of the runs' partials (a launch of its own)
  const double pflops = 4.0 * a.B * a.beam * (double)a.T * a.d, pbytes = 8.0 * a.B * (double)a.T * a.d;
  auto merge = [&](const CrossAttnArgs& c) {
    sbk::ProfScope pm("cross_merge", 0.0, 4.0 * c.B * c.beam * (double)c.d * (c.NS + 1.0), st);
    SBK_LAUNCH(cross_merge_kernel, dim3(c.B * c.beam), dim3(256), 0, st, (const float*)c.part, c.out, c.H, c.NS, c.beam, DH, c.d);
    return sbk::launch_status("cross_merge");
  };
  if constexpr (DH == 64) {
    // The register-ring / MFMA kernel from 128 (utterance, head) pairs on (knob 4 = 7, default; 5 = always, 0 = never): measured
    // per launch at T' 430, d 512, beam 10 (tools/decode_probe.py, profiles/r05_o_*): 16 utterances 25.1 us against 30.0 for the
    // frame-per-thread kernel, 32: 28.6 / 38.1, 64: 35.4 / 55.9, 128: 46.5 / (LDS-DMA kernel) 92.7, 256: 81.0 / 95.9; at 2 and 8
    // utterances the frame-per-thread kernel wins (25.3 / 45.5, 26.9 / 25.6-31.5: an utterance would be cut into 16-frame runs).
    // Runs per utterance: ~1 024 waves (one 4-wave workgroup per CU) but runs of >= 100 frames -- every further run is a partial
    // to write and merge (128 utterances: 46.5 us as one run, 65.2 as two); knob 8 = 3 forces ONE run.
    // (Measured and removed: the LDS-DMA kernel with 16- / 8-frame tiles, the row-coalesced / fp32-MFMA / VALU wave-streaming
    //  kernels, the merge by the last-arriving run: DESIGN.md section 5.)
    const int uh = a.B * a.H;
    if ((sbk::g_cross_rows == 5 || (sbk::g_cross_rows == 7 && uh >= 128)) && a.beam <= 16 && a.H * 64 == a.d &&
        sbk::aligned16(a.kv) && sbk::aligned16(a.q) && sbk::aligned16(a.out) && (a.part || a.T <= 16)) {
      CrossAttnArgs c = a;
      int ns = sbk::g_cross_fc256 == 3 ? 1 : std::min(sbk::cdiv(1024, uh), std::max(1, a.T / 100));
      if (ns > sbk::cdiv(a.T, 16)) ns = sbk::cdiv(a.T, 16);  // (the partial buffer is sized for 16-frame runs)
      const int chunk = sbk::cdiv(sbk::cdiv(a.T, ns), 16) * 16;
      c.NS = sbk::cdiv(a.T, chunk);
      int rc5;
      {
        sbk::ProfScope prof("cross_attn_ring", pflops, pbytes, st);
        const dim3 grid(c.NS, sbk::cdiv(uh, 4));
        switch (sbk::g_nt_mask & 3) {
          case 0: SBK_LAUNCH((cross_attn_ring_kernel<3, false, false>), grid, dim3(256), 0, st, c, chunk); break;
          case 1: SBK_LAUNCH((cross_attn_ring_kernel<3, true, false>), grid, dim3(256), 0, st, c, chunk); break;
          case 2: SBK_LAUNCH((cross_attn_ring_kernel<3, false, true>), grid, dim3(256), 0, st, c, chunk); break;
          default: SBK_LAUNCH((cross_attn_ring_kernel<3, true, true>), grid, dim3(256), 0, st, c, chunk); break;
        }
        rc5 = sbk::launch_status("cross_attn_ring");
      }
      if (rc5 || c.NS == 1) return rc5;
      return merge(c);
    }
  }
  int rc;
  {
    sbk::ProfScope prof("cross_attn_step", pflops, pbytes, st);
    launch_frames<DH>(a, qtiles, st);
    rc = sbk::launch_status("cross_attn_step");
  }
  if (rc || a.NS == 1) return rc;
  return merge(a);
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
                                                              const float* __restrict__ bias2, int ld) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const float* xr = x + (size_t)blockIdx.x * ld;
  float* orow = out + (size_t)blockIdx.x * ld;
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
int g_self_anc = 0;       // key 55: 1 = self-attention of a decoding step over shared ancestry (self_attn_anc_kernel), 0 (default) = a wave per
                          // (hypothesis, head).  The shared-ancestry kernel fetches each distinct cache row once -- with the bench's random-init
                          // weights (flat posteriors: the beams of an utterance diverge early) that is 0.55 x the L2 misses and 0.83 x the requests of
                          // the default, and 34.0 against 28.2 us per launch at 60 steps, 31 against 16 at 24 (profiles/r06_i_*, r06_l_*): off.
                          // The beams of a trained model share all but their last tokens; there it walks ~L + beam rows instead of L x beam.
int g_nt_mask = 7;        // key 53: non-temporal loads of streamed-once data: 1 = ring K tiles, 2 = ring V tiles, 4 = CTC posteriors
int g_cross_rows = 7;     // key 4: 7 (default) = the register-ring / MFMA kernel from 128 (utterance, head) pairs on, else frame-per-thread;
                          // 0 = the frame-per-thread kernel always, 5 = the ring kernel always

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
  if (group < 1 || n % group != 0) group = 1;
  SelfAttnArgs a{qkv, kcache, vcache, kv_slot, out, n, d, H, d / H, step, nslot, Lmax, 1.0f / sqrtf((float)(d / H)),
                 g_step_ptr, key_tok, key_stride, key_shift, key_first, pad_idx, group};
  // beams of an utterance share their ancestry: each distinct cache row fetched once per (utterance, head) (knob 55, see g_self_anc)
  if (g_self_anc && group >= 2 && group <= 16 && d == H * 64 && !key_tok && nslot < (1 << 19) && Lmax < 4096 && aligned16(qkv) &&
      aligned16(kcache) && aligned16(vcache) && aligned16(out) && d % 4 == 0) {
    // list capacity of a wave: its quarter of the positions (16-position granules) with every row distinct + its new-token rows
    const int cap = ((((Lmax + 63) / 64) * 16) + 4) * group;
    const size_t lds_anc = (size_t)4 * cap * sizeof(int2) + (size_t)4 * 64 * 18 * sizeof(float);
    if (lds_anc <= 64 * 1024) {
      ProfScope prof("self_attn_anc", 4.0 * n * d * (step + 1), 8.0 * n * d * (step + 1), st);
      SBK_LAUNCH(self_attn_anc_kernel<3>, dim3((n / group) * H), dim3(256), lds_anc, st, a, cap);
      return launch_status("self_attn_anc");
    }
  }
  const size_t lds = (size_t)8 * (((Lmax + 63) / 64) * 64) * sizeof(float);
  if (lds > 64 * 1024) return fail(SBK_EINVAL, "self_attn_step: Lmax=%d too long for the LDS window", Lmax);
  ProfScope prof("self_attn_step", 4.0 * n * d * (step + 1), 8.0 * n * d * (step + 1), st);
  SBK_LAUNCH(self_attn_step_kernel, dim3(cdiv(n * H, 4)), dim3(256), lds, st, a);
  return launch_status("self_attn_step");
}

// Number of memory splits used for T frames and floats of partial storage they need.
int g_cross_fc256 = 0;  // tuning knob (key 8): frame-per-thread kernel: 1 = 256, 2 = 64 memory frames per workgroup (0 = 128); ring kernel: 3 = one run
                        // per utterance always
int cross_attn_splits(int T) { return cdiv(T, 16); }  // sizes the partial buffer for the finest split (16-frame runs)
size_t cross_attn_partial_floats(int B, int T, int H, int Dh, int beam) {
  const int ns = cross_attn_splits(T);
  return ns > 1 ? (size_t)B * H * ns * beam * (Dh + 2) : 0;
}

int cross_attn_step(const float* q, const float* kv, const int32_t* enc_len, float* out, float* part, int B, int T,
                    int d, int H, int beam, hipStream_t st) {
  if (B == 0) return 0;
  const int Dh = d / H;
  const int fc = g_cross_rows != 0 ? kFC : (g_cross_fc256 == 1 ? 256 : (g_cross_fc256 == 2 ? 64 : kFC));
  const int NS = cdiv(T, fc);
  if (NS > 1 && !part) return fail(SBK_EINVAL, "cross_attn_step: T=%d needs a partial buffer", T);
  CrossAttnArgs a{q, kv, enc_len, out, part, B, T, d, H, Dh, beam, NS, 1.0f / sqrtf((float)Dh), fc};
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
                     const float* bias, const float* bias2, int ld) {
  if (rows == 0) return 0;
  ProfScope prof("log_softmax", 4.0 * rows * V, 8.0 * rows * V, st);
  SBK_LAUNCH(log_softmax_row_kernel, dim3(rows), dim3(256), 0, st, x, out, V, 1.0f / temperature, weight, bias, bias2, ld < V ? V : ld);
  return launch_status("log_softmax_rows");
}

}  // namespace sbk

extern "C" int sbk_log_softmax_f32(const float* x, float* out, int rows, int V, float temperature, float weight,
                                   sbk_stream_t stream) {
  if (rows == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(x && out && rows >= 0 && V > 0 && temperature > 0.0f, "log_softmax: bad arguments");
  return sbk::log_softmax_rows(x, out, rows, V, temperature, weight, sbk::as_stream(stream), nullptr, nullptr, V);
}
