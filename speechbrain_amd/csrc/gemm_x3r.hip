// Few-row fp32 contraction on the bf16 matrix pipe: sbk_gemm_nt_x3r / sbk_gemm_ln_nt_x3r (the decode step's projections of the
// grouped searches; reference: the Linear layers of TransformerDecoderLayer.forward, lobes/models/transformer/Transformer.py:751-834).
// Moved out of gemm.hip in round 5 together with the removal of its measured-and-lost variants.
#include "common.h"
#include "internal.h"

namespace {

using sbk::f32x16;

// ---------------------------------------------------------------------------------------------------------------
// Few-row fp32 contraction on the bf16 matrix pipe ("x3r": the decode step's projections, 200 - 4 000 hypothesis rows).
// A kernel trace of a decoding step (profiles/r04_g_*) shows its launches back to back (45 us of gaps in 2.1 ms): the step
// is the sum of its kernels, and 53 % of that were the decoder layers' projections at 27 - 80 TF/s -- register-operand
// fp32-MFMA tiles whose operands arrive as 16-byte pieces of 64 different rows per load instruction (the texture
// path serialises them) and whose 64 x 64 x 128 wave slices cost 4 096 matrix cycles each.  Here
//   * W arrives pre-split in PANEL layout (sbk_split_x3p: the weights' image the encoder uses): a fragment load of 32
//     rows is two 512-byte runs, and the same three-way operand split as sbk_gemm_nt_f32x3 puts the products on
//     v_mfma_f32_32x32x16_bf16 (6 MFMAs of 8 passes per 16 k instead of 8 MFMAs of 16 passes);
//   * A is fp32 [M, K], cut into its three pieces in registers (a lane holds 8 consecutive k of its row per step).  (Its panel
//     image as the operand -- written by LayerNorm / the previous projection's epilogue -- was built and measured in round 4:
//     no faster inside the contraction and its producers cost more than they save at 1 280 rows, profiles/r04_i_*; removed);
//   * tile 64 x 64, the four waves split K four ways -- each takes K / 64 steps of 16 k, however long K is: no LDS
//     staging, no barrier in the loop, one LDS exchange of the partial tiles at the end.  (The fp32-MFMA kernels split a
//     long K across workgroups as well, to fill the chip, and pay a reduce launch: on the bf16 pipe the 2 048-deep
//     feed-forward projection of 1 280 rows is 10 us of matrix time per workgroup on 160 CUs);
//   * the k steps run through a ring of four single-step operand buffers: the loads of step s + 3
//     are in flight under the MFMAs of step s (<= 256 registers: two workgroups per CU); the step loop is rolled
//     over the ring (K / 64 steps per wave: 8 at K = 512, 32 at 2 048).  The operands of a step are
//     materialised by empty asm anchors where they are used -- without them hipcc hoists the split of an A fragment to
//     its load (a wait in front of the next loads) or sinks the loads to their MFMAs;
//   * W is the FIRST MFMA operand (the wave computes (W tile) . (A tile)^T): a lane owns one row m of C and register
//     quads hold four consecutive columns -- bias / residual / result move as 16-byte vectors.
struct X3rArgs {
  const float* A;    // fp32 [M, K], row stride lda
  const uint4* PW;   // panel image of W [N, K]
  const float* bias;
  const float* R;
  float* C;          // fp32 result
  int ns;            // k steps (of 16) per wave: K / 64, a multiple of 4
  int lda, ldr, ldc, M, N, K, act, tiles_m, tiles_n;
  float alpha;
  float ln_eps;      // LNQ > 0: A is the residual stream, the operand is its LayerNorm (affine folded into PW / bias)
  int xc;            // column groups of the tile space among the 8 XCDs (1, 2, 4 or 8; 8 / xc row groups)
};

// NS > 0: the wave's NS steps fully unrolled (K = 64 NS; measured faster for K = 512 with N >= 1 024: 18.6 / 25.3 / 51.5 us
// against 21.0 / 27.0 / 58.1 rolled at N = 1 536 / 2 048 / 5 000, profiles/r04_i_*, r04_k_*); NS = 0: the loop rolled
// over the ring (any K; faster for N = 512: 12.4 vs 14.2 us, and the only form for long K)
//
// LNQ > 0 (= K / 256; fp32 A only): the LayerNorm in front of the projection runs in its prologue.  gamma / beta are folded
// into the operands by the caller (PW = panel image of W[n,k] gamma[k], bias[n] + sum_k W[n,k] beta[k]: sbk_gemm_ln_nt_f32's
// convention), so  LN(x) . W^T + b = rstd * ((x - mean) . Wf^T) + bf.  Row statistics: wave w takes rows 16 w .. 16 w + 15 of
// the tile, a row per 16 lanes, in the two-pass form of csrc/norm.hip (mean, then the sum of squared deviations) while the
// first operand loads of the step loop are already in flight; x - mean is formed in front of the split (one subtraction
// per element), rstd multiplies the finished tile in the epilogue (a lane owns one row there).  The rows are re-read by
// the step loop through L1 / L2 (the kernel that wrote them ran just before).
//
// (Round 5 measured two more forms on the GPU and removed them: the row statistics handed over by the kernels that write the
// residual stream -- per-block mean / M2 in the producers' epilogues, folded by the consumer, no pass over the rows -- ran the
// decoding step in 2.03 ms against 2.02 for this pre-pass and 2.06 for LayerNorm launches, 11.5-11.6 K audio-s/s either way:
// profiles/r05_a_*.)
// PAIR (round 6, knob 58): the operand loads of TWO consecutive k steps are issued together.  A lane reads 32 bytes of its A row per
// step, so a 128-byte line is touched by two steps -- issued one compute step apart (the rolling schedule below) the line has left
// the 32 KB vector L1 in between (four waves x three steps x 8 KB of lines in flight) and comes from L2 twice; issued back to back
// the second access merges with the first.
template <int NS, int LNQ = 0, bool PAIR = false>
__global__ void __launch_bounds__(256, 2) gemm_x3r_kernel(X3rArgs g) {
  constexpr int DEPTH = 4, PD = 3;  // ring size (= the unrolled body of the step loop), prefetch distance (registers)
  __shared__ float4 red[4][4][4][64];  // [wave][sub-tile][register quad][lane]: partial tiles of the four K slices (64 KB)
  __shared__ float2 stat[LNQ > 0 ? 64 : 1];  // (mean, rstd) of the tile's rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nt, mt;
  {  // XCD-aware tile order: workgroup id = 8 q + x runs on XCD x (observed placement, speed only).  The XCDs own the tile
     // space as xc column groups x 8 / xc row groups: XCD x takes the column tiles nt = x % xc (mod xc) of the row tiles mt =
     // x / xc (mod 8 / xc), so an XCD's L2 fetches 1 / (8 / xc) of A and 1 / xc of W -- xc * |A| + (8 / xc) * |W| bytes cross the
     // fabric per launch (xc = 8, rounds 4-5a: every XCD fetched all of A)
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int c = g.xc, r = 8 / c, rm = (g.tiles_m + r - 1) / r;
    mt = x / c + r * (q % rm);
    nt = x % c + c * (q / rm);
    if (mt >= g.tiles_m || nt >= g.tiles_n) return;
  }
  const int r = lane & 31, half = lane >> 5;
  const int ns = NS > 0 ? NS : g.ns, k_begin = wave * ns * 16, KB = g.K >> 4;
  const float* arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) arow[i] = g.A + (size_t)min(mt * 64 + i * 32 + r, g.M - 1) * g.lda + k_begin + 8 * half;
  // chunk (row block, k step, piece 0, half), slot r: sub-tile i / j is 32 slots on, piece p two chunks (128 slots), a k step six
  const uint4* wp = g.PW + ((size_t)(nt * KB + (k_begin >> 4)) * 6 + half) * 64 + r;
  float4 av[DEPTH][2][2];
  uint4 wv[DEPTH][2][3];
  auto load = [&](int buf, int st) SBK_INLINE_LAMBDA {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) wv[buf][j][p] = wp[(size_t)st * 384 + p * 128 + j * 32];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      av[buf][i][0] = *reinterpret_cast<const float4*>(arow[i] + st * 16);
      av[buf][i][1] = *reinterpret_cast<const float4*>(arow[i] + st * 16 + 4);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
  float mean[2] = {0.0f, 0.0f};  // (LNQ > 0) of this lane's two operand rows
  auto compute = [&](int buf) SBK_INLINE_LAMBDA {
    sbk::bf16x8 ap[2][3], bp[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        sbk::pin(wv[buf][j][p].x), sbk::pin(wv[buf][j][p].y), sbk::pin(wv[buf][j][p].z), sbk::pin(wv[buf][j][p].w);
        const uint4 u = wv[buf][j][p];
        bp[j][p] = sbk::bf16x8_from_words(u.x, u.y, u.z, u.w);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        sbk::pin(av[buf][i][e].x), sbk::pin(av[buf][i][e].y), sbk::pin(av[buf][i][e].z), sbk::pin(av[buf][i][e].w);
      }
      const float4 x0 = av[buf][i][0], x1 = av[buf][i][1];
      float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if constexpr (LNQ > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] -= mean[i];
      }
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {  // x = hi + mid + lo exactly (8 significand bits each, remainders exact in fp32)
        h[p] = sbk::bf16_pair(x[2 * p], x[2 * p + 1]);
        const float r0 = x[2 * p] - __uint_as_float(h[p] << 16), r1 = x[2 * p + 1] - __uint_as_float(h[p] & 0xffff0000u);
        m[p] = sbk::bf16_pair(r0, r1);
        l[p] = sbk::bf16_pair(r0 - __uint_as_float(m[p] << 16), r1 - __uint_as_float(m[p] & 0xffff0000u));
      }
      ap[i][0] = sbk::bf16x8_from_words(h[0], h[1], h[2], h[3]);
      ap[i][1] = sbk::bf16x8_from_words(m[0], m[1], m[2], m[3]);
      ap[i][2] = sbk::bf16x8_from_words(l[0], l[1], l[2], l[3]);
    }
    // the six partial products of relative size >= 2^-17, smallest first; consecutive MFMAs go to different accumulators
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bp[j][PB_[t]], ap[i][PA_[t]], acc[i][j]);
  };
#pragma unroll
  for (int st = 0; st < (PAIR ? DEPTH : PD); ++st) load(st, st);  // (ns >= 4 > PD)
  if constexpr (LNQ > 0) {
    // a row per 16 lanes (four rows per pass, four passes): 16-byte loads 256 B apart, both sums by DPP inside the row of lanes
    const float inv_k = 1.0f / (float)g.K;
    const int grp = lane >> 4, t = lane & 15;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rl = wave * 16 + ps * 4 + grp;
      const float4* xr = reinterpret_cast<const float4*>(g.A + (size_t)min(mt * 64 + rl, g.M - 1) * g.lda);
      float4 xs[4 * LNQ];
#pragma unroll
      for (int j = 0; j < 4 * LNQ; ++j) xs[j] = xr[t + 16 * j];
      float sm = 0.0f;
#pragma unroll
      for (int j = 0; j < 4 * LNQ; ++j) sm += (xs[j].x + xs[j].y) + (xs[j].z + xs[j].w);
      const float mu = sbk::group_sum<16>(sm) * inv_k;
      float qs = 0.0f;
#pragma unroll
      for (int j = 0; j < 4 * LNQ; ++j) {
        const float a = xs[j].x - mu, b = xs[j].y - mu, c = xs[j].z - mu, d = xs[j].w - mu;
        qs += (a * a + b * b) + (c * c + d * d);
      }
      const float rs = rsqrtf(sbk::group_sum<16>(qs) * inv_k + g.ln_eps);
      if (t == 0) stat[rl] = make_float2(mu, rs);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) mean[i] = stat[i * 32 + r].x;
  }
  if constexpr (PAIR && NS > 0) {
#pragma unroll
    for (int st = 0; st < NS; st += 2) {
      compute(st % DEPTH);
      compute((st + 1) % DEPTH);
      sbk::sched_fence();
      if (st + DEPTH < NS) {
        load(st % DEPTH, st + DEPTH);
        load((st + 1) % DEPTH, st + DEPTH + 1);
      }
      sbk::sched_fence();
    }
  } else if constexpr (PAIR) {
#pragma unroll 1
    for (int s0 = 0; s0 < ns; s0 += DEPTH) {  // (ns is a multiple of 4)
#pragma unroll
      for (int j = 0; j < DEPTH; j += 2) {
        compute(j);
        compute(j + 1);
        sbk::sched_fence();
        if (s0 + j + DEPTH < ns) {  // (uniform)
          load(j, s0 + j + DEPTH);
          load(j + 1, s0 + j + DEPTH + 1);
        }
        sbk::sched_fence();
      }
    }
  } else if constexpr (NS > 0) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st + PD < NS) load((st + PD) % DEPTH, st + PD);
      sbk::sched_fence();
      compute(st % DEPTH);
    }
  } else {
#pragma unroll 1
    for (int s0 = 0; s0 < ns; s0 += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        if (s0 + j + PD < ns) load((j + PD) % DEPTH, s0 + j + PD);  // (uniform)
        sbk::sched_fence();
        compute(j);
      }
    }
  }
  // lane = row (sub_m * 32 + r), register quad q4 = columns sub_n * 32 + 8 q4 + 4 half .. +3
  const int row = (mt * 2 + (wave >> 1)) * 32 + r, col0 = (nt * 2 + (wave & 1)) * 32 + 4 * half;
  const bool row_ok = row < g.M;
  const float* rrow = g.R ? g.R + (size_t)(row_ok ? row : 0) * g.ldr : nullptr;
  float4 bv[4], rv[4];
  bool ok[4];
  auto side_loads = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {  // (N % 4 == 0: a vector is inside the matrix or outside as a whole)
      const int col = col0 + 8 * q4;
      ok[q4] = row_ok && col < g.N;
      bv[q4] = (g.bias && ok[q4]) ? *reinterpret_cast<const float4*>(g.bias + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      rv[q4] = (rrow && ok[q4]) ? *reinterpret_cast<const float4*>(rrow + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  side_loads();  // bias / residual requested under the exchange of the partial tiles (the ring's registers are free: -1.2 % per launch, profiles/r05_p_*)
  // every wave publishes its four partial sub-tiles; wave s then owns sub-tile s = 2 i + j (fixed summation order)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        red[wave][2 * i + j][q4][lane] = make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
  __syncthreads();
  float4 v[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4 a0 = red[0][wave][q4][lane], a1 = red[1][wave][q4][lane], a2 = red[2][wave][q4][lane], a3 = red[3][wave][q4][lane];
    v[q4] = make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z,
                        ((a0.w + a1.w) + a2.w) + a3.w);
  }
  if constexpr (LNQ > 0) {
    const float rs = stat[(wave >> 1) * 32 + r].y;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) v[q4].x *= rs, v[q4].y *= rs, v[q4].z *= rs, v[q4].w *= rs;
  }
  float o[16];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    o[4 * q4] = v[q4].x + bv[q4].x, o[4 * q4 + 1] = v[q4].y + bv[q4].y, o[4 * q4 + 2] = v[q4].z + bv[q4].z, o[4 * q4 + 3] = v[q4].w + bv[q4].w;
  }
  switch (g.act) {  // uniform
    case SBK_ACT_SWISH:
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = o[q] / (1.0f + expf(-o[q]));
      break;
    case SBK_ACT_GELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = 0.5f * o[q] * (1.0f + erff(o[q] * 0.70710678118654752440f));
      break;
    case SBK_ACT_RELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = o[q] > 0.0f ? o[q] : 0.0f;
      break;
    case SBK_ACT_LEAKY_RELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = o[q] > 0.0f ? o[q] : 0.01f * o[q];
      break;
    default: break;
  }
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const int col = col0 + 8 * q4;
    const float o0 = o[4 * q4] * g.alpha + rv[q4].x, o1 = o[4 * q4 + 1] * g.alpha + rv[q4].y;
    const float o2 = o[4 * q4 + 2] * g.alpha + rv[q4].z, o3 = o[4 * q4 + 3] * g.alpha + rv[q4].w;
    if (!ok[q4]) continue;
    *reinterpret_cast<float4*>(g.C + (size_t)row * g.ldc + col) = make_float4(o0, o1, o2, o3);
  }
}

// (Round 6 built the same contraction on 128 x 64 tiles -- a W fragment serving four row sub-tiles: 0.29 instead of 0.42 KB of
// operands per MFMA, half the workgroups, 128 accumulator registers at one wave per SIMD -- bit-identical results, and measured it
// on the GPU: 15.9 against 12.6 us at (1 280, 512, 512), 45.4 against 29.5 at K = 2 048, the decoding step 1.98 against 1.67 ms;
// only from 2 560 rows on does it win (16.6 / 46.0 against 20.0 / 56.5 us), and the 8-worker headline is the same either way
// (12 145 / 12 184 against 12 139 / 12 324): a launch of this size is ~10 us of fixed latency -- launch, first operand round trip,
// exchange, stores -- plus its MFMA chain, and halving the waves doubles the chain.  Removed; profiles/r06_e_*.)
// (Also round 6: EIGHT waves per workgroup cutting K eight ways -- a wave's chain of dependent k steps halves, two waves per SIMD, the
// partial tiles met in two stages through the same 64 KB of LDS.  In the device timeline of a decoding step at 1 280 rows: 12.1 against
// 13.3 us at N = K = 512, 13.9 against 13.1 with the LayerNorm prologue, 33.8 against 34.8 at K = 2 048 -- the long-K projection is not
// waiting for its chain: 160 workgroups pull 205 MB of operand fragments out of L2 in 34 us, 6 TB/s --, 41.5 against 30.9 at N = 2 048;
// the step 1.72 ms either way with N <= 512 routed to it, the headline 11.8-11.9 against 12.0 K.  Removed; profiles/r06_k_*.)
// MEASUREMENT ONLY (knob 54 = 2 / 3): every XCD reads the whole buffer once (workgroup b runs on XCD b % 8 and takes slice b / 8
// of 32), so that the launch behind it finds the operand in its XCD's L2 -- prices what a projection loses to cold operands
// inside a decoding step (profiles/r06_b_*).  `sink` is never written (the sum of finite values is not NaN-compared true).
__global__ void __launch_bounds__(256) x3r_touch_kernel(const uint4* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
  const size_t per = (n16 + 31) / 32, b0 = (size_t)(blockIdx.x >> 3) * per, b1 = b0 + per < n16 ? b0 + per : n16;
  unsigned acc = 0;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9e3779b9u && n16 == 1) *sink = acc;
}

}  // namespace

namespace sbk {

// tuning knob (key 41): the decode step's projections on gemm_x3r_kernel: 0 = off (register-operand fp32-MFMA tiles),
// otherwise on (3 = always the rolled step loop)
int g_x3r_mode = 2;
int g_x3r_min_rows = 192;  // key 42: rows from which the search routes a projection with a panel image to it
bool x3r_routed(int M, int N, int K) { return g_x3r_mode != 0 && M >= g_x3r_min_rows && K % 256 == 0 && N % 4 == 0; }
// key 45: the LayerNorm in front of a routed projection of the search: 0 = a launch of its own; 1 (default) = in the projection's
// prologue from a pre-pass over the rows (gemm_ln_nt_x3r; the vocabulary projection only below 4 096 columns: every column
// tile repeats the statistics), 2 = the same, wide vocabularies included
int g_x3r_ln = 1;
int g_x3r_pair = 1;   // key 58: operand loads of two k steps issued together (bit 0: plain kernel -- the default: 31.0 against 34.8 us at K = 2 048,
                      // 13.0-13.2 against 13.3 at N = K = 512 in the step's device timeline; bit 1: with the LayerNorm prologue -- off: 32.6 /
                      // 22.7 / 14.4 against 31.5 / 21.8 / 13.0 us, its rows are in the L1 / L2 from the statistics pre-pass; profiles/r06_l_*)
int g_x3r_probe = 0;  // key 54, MEASUREMENT ONLY (results of in-place launches are wrong with 1): 1 = every launch is issued twice, the second under the
                      // profiler name *_rep (its operands are where the first left them); 2 = the weight panel is read into every XCD's L2 by a
                      // launch in front ("x3r_touch"); 3 = the panel and the A rows
int g_x3r_xc = 0;  // key 51: column groups among the XCDs (1 / 2 / 4 / 8; 0 = the count that minimises the fabric traffic)

// A fp32 [M, K] (row stride lda); ln_eps >= 0: the operand is LayerNorm(A) over K with the affine folded into (PW, bias), row
// statistics by a pre-pass (K = 256 / 512 / 1 024 / 1 280).  -1: shape not eligible
static int launch_x3r(const float* A, int lda, const uint16_t* PW, const float* bias, const float* R, int ldr, float* C, int ldc,
                      int M, int N, int K, int act, float alpha, float ln_eps, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  const bool ln = ln_eps >= 0.0f;
  if (K % 256 != 0 || N % 4 != 0 || !aligned16(PW) || !A || !C || lda % 4 != 0 || !aligned16(A)) return -1;
  if (ldc % 4 != 0 || !aligned16(C) || (R && (ldr % 4 != 0 || !aligned16(R))) || (bias && !aligned16(bias))) return -1;
  if (ln && !(K == 256 || K == 512 || K == 1024 || K == 1280)) return -1;
  const int tm = cdiv(M, 64), tn = cdiv(N, 64);
  int xc = g_x3r_xc;
  if (!(xc == 1 || xc == 2 || xc == 4 || xc == 8)) {  // fewest bytes across the fabric: xc |A| + (8 / xc) |W|
    const double ab = 4.0 * M * (double)K, wb = 6.0 * N * (double)K;
    xc = 8;
    for (int c = 4; c >= 1; c >>= 1)
      if (c * ab + (8 / c) * wb < xc * ab + (8 / xc) * wb) xc = c;
  }
  X3rArgs a{A, reinterpret_cast<const uint4*>(PW), bias, R, C, K / 64, lda, ldr, ldc, M, N, K, act, tm, tn, alpha, ln_eps, xc};
  if (g_x3r_probe >= 2) {
    ProfScope pt("x3r_touch", 0.0, 8.0 * 6.0 * (double)N * K, st);
    SBK_LAUNCH(x3r_touch_kernel, dim3(256), dim3(256), 0, st, reinterpret_cast<const uint4*>(PW), (size_t)N * K * 6 / 16, (unsigned*)nullptr);
    if (g_x3r_probe == 3 && lda == K)
      SBK_LAUNCH(x3r_touch_kernel, dim3(256), dim3(256), 0, st, reinterpret_cast<const uint4*>(A), (size_t)M * K * 4 / 16, (unsigned*)nullptr);
  }
  const int reps = g_x3r_probe == 1 ? 2 : 1;
  int rc_last = 0;
  for (int rep = 0; rep < reps; ++rep) {
  ProfScope prof(rep ? (ln ? "gemm_ln_x3r_rep" : "gemm_x3r_rep") : (ln ? "gemm_ln_x3r" : "gemm_x3r"), 2.0 * M * N * K,
                 4.0 * M * (double)K + 6.0 * (double)N * K + (4.0 + (R ? 4.0 : 0.0)) * M * (double)N, st);
  dim3 grid(8 * cdiv(tm, 8 / xc) * cdiv(tn, xc)), block(256);
  const int pair = g_x3r_pair;  // bit 0: the plain kernel, bit 1: the LayerNorm-prologue kernel
  if (ln) {
    if (K == 512) {  // (the unrolled step loop for every N: the rolled one is at the register limit without the row means)
      if (pair & 2) SBK_LAUNCH((gemm_x3r_kernel<8, 2, true>), grid, block, 0, st, a);
      else SBK_LAUNCH((gemm_x3r_kernel<8, 2>), grid, block, 0, st, a);
    } else if (K == 256) {
      SBK_LAUNCH((gemm_x3r_kernel<4, 1>), grid, block, 0, st, a);
    } else if (K == 1024) {
      SBK_LAUNCH((gemm_x3r_kernel<0, 4>), grid, block, 0, st, a);
    } else {
      SBK_LAUNCH((gemm_x3r_kernel<0, 5>), grid, block, 0, st, a);
    }
    rc_last = launch_status("gemm_ln_x3r");
  } else {
    if (K == 512 && N >= 1024 && g_x3r_mode != 3) {
      if (pair & 1) SBK_LAUNCH((gemm_x3r_kernel<8, 0, true>), grid, block, 0, st, a);
      else SBK_LAUNCH((gemm_x3r_kernel<8>), grid, block, 0, st, a);
    } else {
      if (pair & 1) SBK_LAUNCH((gemm_x3r_kernel<0, 0, true>), grid, block, 0, st, a);
      else SBK_LAUNCH((gemm_x3r_kernel<0>), grid, block, 0, st, a);
    }
    rc_last = launch_status("gemm_x3r");
  }
  if (rc_last) return rc_last;
  }
  return rc_last;
}
int gemm_nt_x3r(const float* A, int lda, const uint16_t* PW, const float* bias, const float* R, int ldr, float* C, int ldc, int M,
                int N, int K, int act, float alpha, hipStream_t st) {
  return launch_x3r(A, lda, PW, bias, R, ldr, C, ldc, M, N, K, act, alpha, -1.0f, st);
}
// C = epilogue(LN(A) . W^T + b) for the rows the search routes to gemm_x3r: PWf = panel image of W[n,k] gamma[k],
// bf[n] = b[n] + sum_k W[n,k] beta[k] (gemm_ln_nt's convention).  -1: shape not eligible
int gemm_ln_nt_x3r(const float* A, int lda, const uint16_t* PWf, const float* bf, const float* R, int ldr, float* C, int ldc, int M,
                   int N, int K, float eps, int act, float alpha, hipStream_t st) {
  if (!(eps >= 0.0f)) return -1;
  return launch_x3r(A, lda, PWf, bf, R, ldr, C, ldc, M, N, K, act, alpha, eps, st);
}
bool x3r_ln_routed(int K) { return g_x3r_ln != 0 && (K == 256 || K == 512 || K == 1024 || K == 1280); }

}  // namespace sbk

extern "C" int sbk_gemm_nt_x3r(const float* A, int lda, const uint16_t* PW, const float* bias, const float* residual, int ldr,
                               float* C, int ldc, int M, int N, int K, int act, float alpha, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && PW && C, "gemm_x3r: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && N % 4 == 0 && K >= 256 && K % 256 == 0,
              "gemm_x3r: bad shape M=%d N=%d K=%d (N: a multiple of 4, K: a multiple of 256)", M, N, K);
  SBK_REQUIRE(lda >= K && lda % 4 == 0 && sbk::aligned16(A), "gemm_x3r: rows of A must be 16-byte aligned (lda=%d)", lda);
  SBK_REQUIRE(sbk::aligned16(PW), "gemm_x3r: the panel image must be 16-byte aligned");
  SBK_REQUIRE(ldc >= N && ldc % 4 == 0 && sbk::aligned16(C), "gemm_x3r: C rows are stored as 16-byte vectors (ldc=%d)", ldc);
  SBK_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && sbk::aligned16(residual)), "gemm_x3r: residual stride / alignment");
  SBK_REQUIRE(!bias || sbk::aligned16(bias), "gemm_x3r: bias alignment");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_x3r: unknown activation %d", act);
  const int rc = sbk::gemm_nt_x3r(A, lda, PW, bias, residual, ldr, C, ldc, M, N, K, act, alpha, sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_x3r: shape not eligible");
  return rc;
}

extern "C" int sbk_gemm_ln_nt_x3r(const float* A, int lda, const uint16_t* PWf, const float* bf, const float* residual, int ldr,
                                  float* C, int ldc, int M, int N, int K, float eps, int act, float alpha, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && PWf && C, "gemm_ln_x3r: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && N % 4 == 0 && (K == 256 || K == 512 || K == 1024 || K == 1280),
              "gemm_ln_x3r: bad shape M=%d N=%d K=%d (N: a multiple of 4, K: 256, 512, 1024 or 1280)", M, N, K);
  SBK_REQUIRE(eps >= 0.0f, "gemm_ln_x3r: eps must be >= 0");
  SBK_REQUIRE(lda >= K && lda % 4 == 0 && sbk::aligned16(A), "gemm_ln_x3r: rows of A must be 16-byte aligned (lda=%d)", lda);
  SBK_REQUIRE(sbk::aligned16(PWf), "gemm_ln_x3r: the panel image must be 16-byte aligned");
  SBK_REQUIRE(ldc >= N && ldc % 4 == 0 && sbk::aligned16(C), "gemm_ln_x3r: C rows are stored as 16-byte vectors (ldc=%d)", ldc);
  SBK_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && sbk::aligned16(residual)), "gemm_ln_x3r: residual stride / alignment");
  SBK_REQUIRE(!bf || sbk::aligned16(bf), "gemm_ln_x3r: bias alignment");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_ln_x3r: unknown activation %d", act);
  const int rc = sbk::gemm_ln_nt_x3r(A, lda, PWf, bf, residual, ldr, C, ldc, M, N, K, eps, act, alpha, sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_ln_x3r: shape not eligible");
  return rc;
}
