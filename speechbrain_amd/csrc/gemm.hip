// fp32 MFMA GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)   (sbk_gemm_nt_f32)
//
// Roofline: MFMA fp32 (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense on MI355X).
// One workgroup owns a BM x BN tile of C; its waves each own a WM x WN
// sub-tile made of 32x32 MFMA accumulators.  A and W panels are staged through
// LDS as [rows][BK+1] (odd pitch => the 32 lanes of an MFMA operand read hit 32
// distinct banks), loaded from HBM as 16-byte vectors along K (both operands
// are K-contiguous, so every global load is a full 128-byte line per 8 lanes)
// into registers one K tile ahead, so HBM latency hides under the MFMAs.
// The epilogue (bias, activation, scaled residual) runs on the accumulators in
// registers and writes 128-byte rows (32 lanes x 4 B) per store instruction.
#include "common.h"
#include "internal.h"

#include <map>
#include <mutex>
#include <type_traits>

namespace {

using sbk::f32x16;

#include "gemm_common.h"

// Epilogue of a 32x32 register tile held by ONE wave (lane: column r, rows (q&3) + 8*(q>>2) + 4*half): straight-line
// code -- the residual rows / sequence lengths are requested together, the activation is chosen by ONE uniform
// switch outside the per-row work, and the sixteen row stores are issued back to back (a branchy per-row loop makes
// the compiler drain the memory counter before every store: sixteen serialized round trips on a 5 us kernel).
__device__ __forceinline__ void tile_epilogue_32x32(const GemmArgs& g, float (&v)[16], int mt, int nt, int r, int half) {
  const int col = nt * 32 + r;
  const bool col_ok = col < g.N;
  const float bv = (g.bias && col_ok) ? g.bias[col] : 0.0f;
  float res[16];
  int len[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
    const bool live = col_ok && row < g.M;
    res[q] = (live && g.R) ? g.R[(size_t)row * g.ldr + col] : 0.0f;
    len[q] = (live && g.seq_len) ? g.seq_len[row / g.rows_per_seq] : 0x7fffffff;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] += bv;
  switch (g.act) {  // uniform
    case SBK_ACT_SWISH:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = v[q] / (1.0f + expf(-v[q]));
      break;
    case SBK_ACT_GELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.70710678118654752440f));
      break;
    case SBK_ACT_RELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = v[q] > 0.0f ? v[q] : 0.0f;
      break;
    case SBK_ACT_LEAKY_RELU:
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = v[q] > 0.0f ? v[q] : 0.01f * v[q];
      break;
    default: break;
  }
  if (g.seq_len) {  // uniform
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      v[q] = (row % g.rows_per_seq) >= len[q] ? 0.0f : v[q] * g.alpha;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] *= g.alpha;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] += res[q];
  sbk::sched_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
    if (col_ok && row < g.M) g.C[(size_t)row * g.ldc + col] = v[q];
  }
}

// W [N][K] fp32 -> [N][K/32][3][32] bf16: the three exact pieces of every element (hi = bf16(x), mid = bf16(x - hi),
// lo = x - hi - mid, round to nearest even), one 192-byte record per row and 32-deep K tile -- the W operand of gemm_nt_sk_kernel<.., X3>
__global__ void __launch_bounds__(256) split_bf16x3_kernel(const float* __restrict__ W, int ldw, unsigned short* __restrict__ out,
                                                           long n, int K) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / K;
    const int k = (int)(i - row * K);
    const float x = W[row * ldw + k];
    const unsigned short hi = sbk::f32_to_bf16(x);
    const float r = x - __uint_as_float((unsigned)hi << 16);
    const unsigned short mid = sbk::f32_to_bf16(r);
    const float q = r - __uint_as_float((unsigned)mid << 16);  // <= 7 significant bits: a bf16 exactly
    unsigned short* o = out + (row * (K / 32) + k / 32) * 96 + (k & 31);
    o[0] = hi;
    o[32] = mid;
    o[64] = sbk::f32_to_bf16(q);
  }
}
// Register-staged panel: global -> registers (issued early, in flight under the MFMAs of the previous
// K tile) -> LDS [rows][BK+1].
template <int ROWS, int BK, int NT>
struct PanelStage {
  static constexpr int V = BK / 4;                       // float4 slots per row
  static constexpr int PER = (ROWS * V + NT - 1) / NT;   // float4 slots per thread
  float4 r[PER];

  template <bool VEC>
  __device__ __forceinline__ void fetch(const float* __restrict__ src, int ld, int row0, int nrows, int k0, int K,
                                        int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      const int rr = s / V, c = (s % V) * 4;
      const int gr = row0 + rr, gk = k0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s < ROWS * V && gr < nrows) {
        const float* p = src + (size_t)gr * ld + gk;
        if (VEC && gk + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
      r[i] = v;
    }
  }
  // Interior tiles (whole panel inside the matrix, K % BK == 0, 16-byte aligned rows): no predicates at all.  The
  // predicated form above costs a branch + mask sequence per load (~150 instructions per K tile and wave).
  __device__ __forceinline__ void fetch_interior(const float* __restrict__ src, int ld, int row0, int k0, int tid) {
    static_assert((ROWS * V) % NT == 0, "fetch_interior: the panel must divide evenly over the threads");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      r[i] = *reinterpret_cast<const float4*>(src + (size_t)(row0 + s / V) * ld + k0 + (s % V) * 4);
    }
  }
  // [rows][BK+4] layout: one 16-byte LDS store per slot (rows stay 16-byte aligned: (BK+4)*4 is a multiple of 16)
  __device__ __forceinline__ void commit_vec(float (*dst)[BK + 4], int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      if (s < ROWS * V) *reinterpret_cast<float4*>(&dst[s / V][(s % V) * 4]) = r[i];
    }
  }
  __device__ __forceinline__ void commit(float (*dst)[BK + 1], int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid + i * NT;
      if ((ROWS * V) % NT == 0 || s < ROWS * V) {  // evenly divided panels: every slot exists (no mask, no branch)
        const int rr = s / V, c = (s % V) * 4;
        dst[rr][c] = r[i].x;
        dst[rr][c + 1] = r[i].y;
        dst[rr][c + 2] = r[i].z;
        dst[rr][c + 3] = r[i].w;
      }
    }
  }
};

template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__device__ __forceinline__ void gemm_nt_tile(const GemmArgs& g, int* tile_x = nullptr, int* tile_y = nullptr) {
  constexpr int WAVES_N = BN / WN;
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  __shared__ float As[BM][BK + 1];
  __shared__ float Ws[BN][BK + 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  // XCD-aware tile order (workgroup id % 8 = XCD): each XCD walks a contiguous range of tiles, so the
  // A row panel shared by neighbouring tiles is fetched into one L2 instead of eight.
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
  if (tile_x) {  // (the caller continues on this tile)
    *tile_x = bx;
    *tile_y = by;
  }
  const int lrow = lane & 31, lk = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  PanelStage<BM, BK, NT> pa;
  PanelStage<BN, BK, NT> pw;
  // (uniform per workgroup) interior tile: unpredicated panel loads
  const bool interior = VEC && m0 + BM <= g.M && n0 + BN <= g.N && (g.K % BK) == 0 && (BM * (BK / 4)) % NT == 0 &&
                        (BN * (BK / 4)) % NT == 0;
  auto fetch = [&](int k0) {
    if constexpr ((BM * (BK / 4)) % NT == 0 && (BN * (BK / 4)) % NT == 0) {
      if (interior) {
        pa.fetch_interior(g.A, g.lda, m0, k0, tid);
        pw.fetch_interior(g.W, g.ldw, n0, k0, tid);
        return;
      }
    }
    pa.template fetch<VEC>(g.A, g.lda, m0, g.M, k0, g.K, tid);
    pw.template fetch<VEC>(g.W, g.ldw, n0, g.N, k0, g.K, tid);
  };
  fetch(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    pa.commit(As, tid);
    pw.commit(Ws, tid);
    __syncthreads();
    if (k0 + BK < g.K) fetch(k0 + BK);  // next K tile: loads fly while this tile is multiplied
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[wm0 + i * 32 + lrow][kk + lk];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Ws[wn0 + j * 32 + lrow][kk + lk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x2(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

  // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5).  (The batched straight-line epilogue of
  // the register-operand kernels was tried here: it costs registers -- 3 -> 2 waves per SIMD -- and 17 % throughput.)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + j * 32 + lrow;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        float v = apply_act(acc[i][j][r] + bv, g.act) * g.alpha;
        if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
        if (g.R) v += g.R[(size_t)row * g.ldr + col];
        g.C[(size_t)row * g.ldc + col] = v;
      }
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool VEC>
// (two waves per SIMD = a 256-register budget wherever the accumulators fit it: with the 512-register budget of one wave
// per SIMD the compiler parks the accumulators in AGPRs and copies every one of them in and out around the K loop body)
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, (WM * WN <= 64 * 64) ? 2 : 1) gemm_nt_kernel(GemmArgs g) {
  gemm_nt_tile<BM, BN, BK, WM, WN, VEC>(g);
}

// The same tile over one K slice per blockIdx.z: raw partial products to ws[z][M][N] (splitk_reduce_kernel applies the
// epilogue).  For few-row, long-K shapes (the decoder's second feed-forward projection at ~1 K rows): 64x64 LDS tiles
// move half the operand bytes of the 32x32 register-operand tiles through L2, and the K split puts 2-3 workgroups on
// every CU so that their MFMA and load phases overlap.
template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, (WM * WN <= 64 * 64) ? 2 : 1) gemm_nt_splitk_kernel(GemmArgs g, float* __restrict__ ws,
                                                                                     int kper) {
  const int z = blockIdx.z;
  g.A += (size_t)z * kper;
  g.W += (size_t)z * kper;
  g.K = (g.K - z * kper) < kper ? (g.K - z * kper) : kper;
  g.C = ws + (size_t)z * g.M * g.N;
  g.ldc = g.N;
  g.bias = nullptr;
  g.R = nullptr;
  g.act = SBK_ACT_NONE;
  g.alpha = 1.0f;
  g.seq_len = nullptr;
  gemm_nt_tile<BM, BN, BK, WM, WN, VEC>(g);
}

// ... and the reduction by whichever of a tile's K slices finishes last (round 3): every workgroup publishes its partial
// tile (agent-scope release), takes a ticket on the tile's counter, and the last ticket sums the SK partial tiles in
// slice order -- the order splitk_reduce_kernel uses, so the result is bit-identical to the two-launch path -- applies
// the epilogue and re-arms the counter.  One launch less per long-K projection of a decoding step; nobody waits.
// ---------------------------------------------------------------------------
// Persistent GEMM on LDS-DMA panels with a stream-K tail (the encoder's contractions: M = frames of a batch, 4-24 K
// rows, i.e. 100-3000 tiles of 128x128 -- tile counts that fill 256 CUs badly when every workgroup takes whole tiles).
//
// The launch has a FIXED number of workgroups G = 8 W (W per XCD, two per CU).  XCD x (= workgroup id % 8, observed)
// owns the contiguous tile range [T x/8, T (x+1)/8) in row-major tile order; its workgroup j takes
//   * whole tiles  t0 + r W + j,  r = 0 .. T_x / W - 1   (at any moment the XCD's W workgroups multiply W CONSECUTIVE
//     tiles: neighbours share the A row panel and every W column panel through that XCD's L2 -- a contiguous range per
//     workgroup instead was measured at a 29 % L2 hit rate and 3.8x the fabric traffic, profiles/r03_*), then
//   * its share of the XCD's T_x % W LEFTOVER tiles, stream-K style: the leftover (tile, 32-deep K tile) units are cut
//     into W equal contiguous ranges, so the tail costs (T_x % W) / W of a tile time instead of a whole one.
// A leftover tile whose K range is cut is finished by whichever of its workgroups arrives last: each writes its partial
// accumulators to its own slab, publishes (agent-scope release, MI355X_MICROARCH.md "Workgroup dispatch ... visibility")
// and takes a ticket on the tile's counter; the last ticket reads ALL the tile's slabs in K order (fixed summation
// order => run-to-run deterministic), applies the epilogue and re-arms the counter.  No workgroup ever waits for
// another one, so nothing depends on residency or dispatch order.  Workgroups of the upper half of an XCD run their
// tail share FIRST (knob 23): the two workgroups of a CU are then half a tile apart and do not sit in their epilogues
// (no MFMA) at the same time.
//
// Panels: global_load_lds_dwordx4 straight into a double-buffered LDS image [stage][A 128 rows | W 128 rows][32 k]
// (no staging registers, no ds_write pass, ONE barrier per K tile).  The LDS image of an LDS-DMA is lane-linear, so
// the bank swizzle is applied to the global SOURCE address: the 16-byte slot s of row r lands in slot s ^ ((r>>1)&7),
// and the MFMA operand fetch (ds_read_b128 of four consecutive k, lane half h of row r reads slot (2g+h) ^ ((r>>1)&7))
// is conflict-free for the 16-lane groups of a b128 read (SQ_LDS_BANK_CONFLICT = 0 measured).  Lanes 0-31 feed
// k = 8g+e, lanes 32-63 k = 8g+4+e of MFMA e of group g (any pairing of the two k slices is a valid contraction as long
// as A and W use the same one).
struct SkArgs {
  GemmArgs g;
  float* slabs;  // [2 * G][128 * 128] partial tiles (slot 0: the workgroup's segment that does not start a tile; 1: the one that does)
  int* cnt;      // [tiles] arrival tickets, zero between launches
  int tiles_n, tiles, KT;
};

// BT: tile edge (128: four waves of 64x64, the encoder shapes; 64: four waves of 32x32, 32 KB of LDS -- the decode-step
// shapes of 640-1 280 rows, where 128-wide tiles leave most CUs without one)
// IL: the LDS-DMA pieces of the next K tile are issued BETWEEN the MFMA groups of the current one (one A piece and one W
// piece per group of 4 k) instead of in one block in front of them: a piece keeps its wave's issue port for ~60-150
// cycles, which a 64-cycle fp32 MFMA in flight covers -- in one block the eight pieces leave the matrix pipe of that
// wave empty for ~1 000 cycles per K tile (the kernel's time was the SUM of its no-load and load-only times).
//
// X3: the fp32 contraction on the bf16 matrix pipe (sbk_gemm_nt_f32x3).  v_mfma_f32_32x32x16_bf16 delivers 16x the
// flops of v_mfma_f32_32x32x2_f32 per cycle, and an fp32 number is EXACTLY the sum of three bf16 numbers (hi = bf16(x),
// mid = bf16(x - hi), lo = x - hi - mid, round to nearest even: the remainders are exact in fp32 and the last one has
// at most 7 significant bits).  a.w = sum of nine partial products; the six of relative size >= 2^-17 are kept (hi.hi,
// hi.mid, mid.hi, hi.lo, lo.hi, mid.mid: exact products, fp32 accumulation on the matrix core), the three dropped ones
// are <= 2^-26 |a||w| each and of either sign (rounded pieces: truncated ones would all carry the product's sign and
// add up), i.e. below the rounding error of ONE fp32 multiply-add -- measured against fp64 the result
// is as close as the fp32 MFMA chain's (tests/test_kernels.py::test_gemm_f32x3).  Six bf16 MFMAs replace sixteen
// fp32-MFMA-equivalents: a 2.67x higher ceiling (2.5 PF/s / 6 = 417 TF/s fp32-equivalent) for the same fp32 result.
// A stays fp32 in HBM and in LDS (same LDS-DMA image as the fp32 kernel) and is cut into its three pieces in registers
// after the operand fetch (two ds_read_b128 = 8 consecutive k of a row = one MFMA operand per piece; ~5 VALU
// instructions per element, which the other wave of the SIMD runs under this wave's MFMAs).  W arrives pre-split
// (sbk_split_bf16x3: [N][K/32][3 pieces][32 k] bf16, 192 contiguous bytes per row and K tile) and lands in LDS as
// [128 rows][3 pieces][4 slots of 8 k]; row pitch 192 B, slot s of row r at s ^ ((r>>2)&3) => the 16-lane groups of a
// ds_read_b128 touch 16 distinct 4-bank groups.
// MEAS (measurement builds of the X3 loop, wrong results): 2 = no operand split, 4 = the hi.hi products only
// (Tried and removed: the X3 panels through registers -- global_load_dwordx4 in front of the MFMA loop, ds_write_b128 behind it --
// instead of by LDS-DMA, whose pieces keep a wave's issue port for 60-180 cycles each.  It needs 40 staging registers: at the
// 256-register budget of two workgroups per CU hipcc spills them, at 512 it parks the accumulators in AGPRs and copies them per
// K tile: 415 us where the LDS-DMA kernel takes 170, profiles/r03_f32x3_sweep.log "g1256".)
template <int BT, bool X3>
__global__ void __launch_bounds__(256, 2) gemm_nt_sk_kernel(SkArgs s) {
  static_assert(BT == 128, "128-wide tiles (the 64-wide instantiation of round 3 -- knobs 25 / 26 -- never won a shape and is gone)");
  constexpr int BK = 32, PANEL = BT * BK, WPITCH = X3 ? 48 : BK, WPANEL = BT * WPITCH, STAGE = PANEL + WPANEL;  // floats
  constexpr int TS = BT / 64, WT = BT / 2, LI = BT / 32;      // 32x32 sub-tiles per wave and dimension, wave tile edge, loader instructions per wave and panel
  // the wave's sub-tiles: TM x TN of 32 x 32.  X3: one row block x four column blocks (a wave = 32 rows of the tile, all
  // its 128 columns): every A element is fetched and split by exactly ONE wave (2 x 2 sub-tiles split it twice)
  constexpr int TM = X3 ? 1 : TS, TN = X3 ? 4 : TS;
  constexpr int LIW = X3 ? BT * 3 / 64 : LI;                  // ... of the W panel (X3: 12 slots of 16 B per row)
  SBK_DYN_LDS(float, lds);  // [2][STAGE] + the ticket word (ONE LDS object: a second one de-pipelines the LDS-DMA loop)
  // kernel arguments into registers (a by-value struct whose address is taken is copied to scratch)
  const float* const gA = s.g.A;
  const float* const gW = s.g.W;
  const float* const gbias = s.g.bias;
  const float* const gR = s.g.R;
  float* const gC = s.g.C;
  const int lda = s.g.lda, ldw = s.g.ldw, ldr = s.g.ldr, ldc = s.g.ldc, M = s.g.M, N = s.g.N, act = s.g.act;
  const float alpha = s.g.alpha;
  const int32_t* const seq_len = s.g.seq_len;
  const int rows_per_seq = s.g.rows_per_seq;
  float* const slabs = s.slabs;
  int* const cnt = s.cnt;
  const int tiles_n = s.tiles_n, KT = s.KT;

  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int wm0 = X3 ? wave * 32 : (wave >> 1) * WT, wn0 = X3 ? 0 : (wave & 1) * WT;
  const int lrow = lane & 31, half = lane >> 5, sw = (lrow >> 1) & 7;
  // ---- this workgroup's segments: whole tiles of the XCD's range, then (or first) its share of the leftover tiles
  const int W = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;  // gridDim.x is a multiple of 8
  const int t0 = (int)((long)s.tiles * x / 8), t1 = (int)((long)s.tiles * (x + 1) / 8);
  const int nfull = sbk::uniform((t1 - t0) / W), R = (t1 - t0) - nfull * W, tb = t0 + nfull * W;
  const int UT = R * KT, ubase = sbk::uniform(UT / W), urem = UT - ubase * W;
  const int q0 = j * ubase + min(j, urem), q1 = q0 + ubase + (j < urem ? 1 : 0);
  int nt = 0, tileA = 0, loA = 0, hiA = 0, hiB = 0;
  if (q1 > q0) {
    const int ta = sbk::uniform(q0 / KT);
    tileA = tb + ta;
    loA = q0 - ta * KT;
    hiA = min(KT, loA + (q1 - q0));
    hiB = q1 - (ta + 1) * KT;  // > 0: the range runs on into the next leftover tile
    nt = hiB > 0 ? 2 : 1;
  }
  const int nseg_wg = nfull + nt;
  if (nseg_wg == 0) return;
  const bool tail_first = j >= (W >> 1);
  auto seg_get = [&](int sidx, int& tile, int& lo, int& hi) SBK_INLINE_LAMBDA {
    const int d = tail_first ? sidx - nt : sidx;
    if (d >= 0 && d < nfull) {
      tile = t0 + d * W + j;
      lo = 0;
      hi = KT;
    } else if ((tail_first ? sidx : sidx - nfull) == 0) {
      tile = tileA;
      lo = loA;
      hi = hiA;
    } else {
      tile = tileA + 1;
      lo = 0;
      hi = hiB;
    }
  };
  // workgroup (index within the XCD) that owns leftover unit q
  auto owner = [&](int q) SBK_INLINE_LAMBDA {
    const int big = urem * (ubase + 1);
    return sbk::uniform(q < big ? q / (ubase + 1) : urem + (q - big) / max(ubase, 1));
  };
  const int p = x * W + j;  // slab owner id

  // loader geometry: wave-instruction i of this wave covers rows (wave*4+i)*8 .. +7 of a panel, 8 slots of 16 B each
  int lrw[LI], lsl[LI];
#pragma unroll
  for (int i = 0; i < LI; ++i) {
    lrw[i] = (wave * LI + i) * 8 + (lane >> 3);
    lsl[i] = ((lane & 7) ^ ((lrw[i] >> 1) & 7)) * 4;  // source k offset (floats) of the slot this lane fills
  }
  // X3: the W panel is [BT rows][12 slots]; lane-linear slot q of the image = row q / 12, position q % 12 = piece * 4 +
  // (logical slot ^ ((row>>2)&3)); the source is the pre-split matrix [N][KT][3][32 bf16] addressed in float units
  int wrw[LIW], wsl[LIW];
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < LIW; ++i) {
      const int q = (wave * LIW + i) * 64 + lane;
      wrw[i] = q / 12;
      const int pos = q - wrw[i] * 12;
      wsl[i] = (pos >> 2) * 16 + (((pos & 3) ^ ((wrw[i] >> 2) & 3)) * 4);
    }
  }
  const float* ap[LI];
  const float* wp[LIW];
  auto setup = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * BT, n0 = (tile % tiles_n) * BT;
#pragma unroll
    for (int i = 0; i < LI; ++i)  // rows past the matrix re-read its last row (their outputs are never stored)
      ap[i] = gA + (size_t)min(m0 + lrw[i], M - 1) * lda + lsl[i];
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < LIW; ++i) wp[i] = gW + (size_t)min(n0 + wrw[i], N - 1) * (KT * 48) + wsl[i];
    } else {
#pragma unroll
      for (int i = 0; i < LI; ++i) wp[i] = gW + (size_t)min(n0 + lrw[i], N - 1) * ldw + lsl[i];
    }
  };
  auto issue = [&](int kt, int stage) SBK_INLINE_LAMBDA {
    float* base = lds + stage * STAGE + (wave * LI) * 256;
#pragma unroll
    for (int i = 0; i < LI; ++i) sbk::glds16(ap[i] + kt * BK, base + i * 256);
    float* wbase = lds + stage * STAGE + PANEL + (wave * LIW) * 256;
#pragma unroll
    for (int i = 0; i < LIW; ++i) sbk::glds16(wp[i] + kt * WPITCH, wbase + i * 256);
  };

  f32x16 acc[TM][TN];
  auto zero = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };
  auto compute = [&](int stage, bool fly, int nkt) SBK_INLINE_LAMBDA {
    const float* As = lds + stage * STAGE + (wm0 + lrow) * BK;
    const float* Ws = lds + stage * STAGE + PANEL + (wn0 + lrow) * WPITCH;
    if constexpr (X3) {
      const int wsw = (lrow >> 2) & 3;
#pragma unroll
      for (int gk = 0; gk < 2; ++gk) {  // 16 k per step: lanes 0-31 supply k = 16 gk .. +7, lanes 32-63 the next eight
        sbk::bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float4 x0 = *reinterpret_cast<const float4*>(As + i * 32 * BK + ((4 * gk + 2 * half) ^ sw) * 4);
          const float4 x1 = *reinterpret_cast<const float4*>(As + i * 32 * BK + ((4 * gk + 2 * half + 1) ^ sw) * 4);
          const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          unsigned h[4], m[4], l[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {  // x = hi + mid + lo exactly: 8 significand bits each, remainders exact in fp32
            h[p] = sbk::bf16_pair(x[2 * p], x[2 * p + 1]);
            const float r0 = x[2 * p] - __uint_as_float(h[p] << 16), r1 = x[2 * p + 1] - __uint_as_float(h[p] & 0xffff0000u);
            m[p] = sbk::bf16_pair(r0, r1);
            l[p] = sbk::bf16_pair(r0 - __uint_as_float(m[p] << 16), r1 - __uint_as_float(m[p] & 0xffff0000u));
          }
          ah[i] = sbk::bf16x8_from_words(h[0], h[1], h[2], h[3]);
          am[i] = sbk::bf16x8_from_words(m[0], m[1], m[2], m[3]);
          al[i] = sbk::bf16x8_from_words(l[0], l[1], l[2], l[3]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float* wr = Ws + j * 32 * WPITCH + ((2 * gk + half) ^ wsw) * 4;
          bh[j] = *reinterpret_cast<const sbk::bf16x8*>(wr);
          bm[j] = *reinterpret_cast<const sbk::bf16x8*>(wr + 16);
          bl[j] = *reinterpret_cast<const sbk::bf16x8*>(wr + 32);
        }
        // smallest terms first; consecutive MFMAs go to different accumulators.  W is the FIRST operand: the wave
        // computes (W tile) . (A tile)^T, so a lane owns one row m of C and registers 4g .. 4g+3 hold four consecutive
        // columns n -- the epilogue loads residuals and stores results as 16-byte vectors (a quarter of the store
        // instructions of the column-per-lane layout: the store tail of a tile is issue-bound, MI355X_MICROARCH.md)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bh[j], al[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bl[j], ah[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bm[j], am[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bh[j], am[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bm[j], ah[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = sbk::mfma_32x32x16_bf16(bh[j], ah[i], acc[i][j]);
      }
      return;
    }
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      const int slot = ((2 * gk + half) ^ sw) * 4;
      float4 a[TS], b[TS];
#pragma unroll
      for (int i = 0; i < TS; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * BK + slot);
#pragma unroll
      for (int j = 0; j < TS; ++j) b[j] = *reinterpret_cast<const float4*>(Ws + j * 32 * BK + slot);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
          for (int j = 0; j < TS; ++j) {
            const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
            const float bw = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
            acc[i][j] = sbk::mfma_32x32x2(av, bw, acc[i][j]);
          }
    }
  };
  auto epilogue = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * BT, n0 = (tile % tiles_n) * BT;
    const bool interior = m0 + BT <= M && n0 + BT <= N;  // uniform: no per-element predicates
    if constexpr (X3) {  // transposed accumulators: lane = row m0 + wm0 + lrow, register 4g+e of block j = column j*32 + 8g + 4*half + e
      const int row = m0 + wm0 + lrow, rowc = min(row, M - 1);
      const bool row_ok = interior || row < M;
      const bool masked = seq_len && (rowc % rows_per_seq) >= seq_len[rowc / rows_per_seq];
      const float ra = masked ? 0.0f : alpha;
      float* crow = gC + (size_t)rowc * ldc;
      const float* rrow = gR ? gR + (size_t)rowc * ldr : nullptr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float4 bv[4], rv[4];
        bool ok[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // (N % 4 == 0: a vector is inside the matrix or outside as a whole)
          const int col = n0 + wn0 + j * 32 + 8 * g + 4 * half;
          ok[g] = row_ok && (interior || col < N);
          bv[g] = (gbias && ok[g]) ? *reinterpret_cast<const float4*>(gbias + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          rv[g] = (rrow && ok[g]) ? *reinterpret_cast<const float4*>(rrow + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[4 * g] = acc[0][j][4 * g] + bv[g].x;
          v[4 * g + 1] = acc[0][j][4 * g + 1] + bv[g].y;
          v[4 * g + 2] = acc[0][j][4 * g + 2] + bv[g].z;
          v[4 * g + 3] = acc[0][j][4 * g + 3] + bv[g].w;
        }
        switch (act) {  // uniform
          case SBK_ACT_SWISH:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            break;
          case SBK_ACT_GELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
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
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + wn0 + j * 32 + 8 * g + 4 * half;
          const float4 o = make_float4(masked ? rv[g].x : v[4 * g] * ra + rv[g].x, masked ? rv[g].y : v[4 * g + 1] * ra + rv[g].y,
                                       masked ? rv[g].z : v[4 * g + 2] * ra + rv[g].z, masked ? rv[g].w : v[4 * g + 3] * ra + rv[g].w);
          if (ok[g]) *reinterpret_cast<float4*>(crow + col) = o;
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 32 + lrow;
      const bool col_ok = interior || col < N;
      const float bv = (gbias && col_ok) ? gbias[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm0 + i * 32 + 4 * half;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
        switch (act) {  // uniform
          case SBK_ACT_SWISH:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            break;
          case SBK_ACT_GELU:
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
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
        if (seq_len) {  // uniform
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
            v[r] = (row % rows_per_seq) >= seq_len[row / rows_per_seq] ? 0.0f : v[r] * alpha;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= alpha;
        }
        if (interior) {
          if (gR) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += gR[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldr + col];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) gC[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = v[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (col_ok && row < M) {
              if (gR) v[r] += gR[(size_t)row * ldr + col];
              gC[(size_t)row * ldc + col] = v[r];
            }
          }
        }
      }
    }
  };
  // a K range [kt_lo, kt_hi) of `tile` is complete in acc.  The ticket word: behind the stages, or (X3: two workgroups
  // of 2 x 40 KB fill the CU's 160 KB) the first word of the stage that was just multiplied -- every wave is past the
  // barrier behind its last read and the next panels go into it only after this function
  auto finish = [&](int tile, int kt_lo, int kt_hi, int stage_done) SBK_INLINE_LAMBDA {
    int* ticket = reinterpret_cast<int*>(lds + (X3 ? stage_done * STAGE : 2 * STAGE));
    bool store = true;
    if (kt_lo != 0 || kt_hi != KT) {  // partial: publish the slab, take a ticket; the last ticket sums the tile's slabs
      const int p_first = owner((tile - tb) * KT), p_last = owner((tile - tb + 1) * KT - 1);
      const int nseg = p_last - p_first + 1;
      float4* mine = reinterpret_cast<float4*>(slabs + (size_t)(2 * p + (kt_lo == 0 ? 1 : 0)) * (BT * BT));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            mine[((wave * (TM * TN) + i * TN + j) * 4 + r4) * 64 + lane] =
                make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
      sbk::vm_drain();
      __syncthreads();
      if (tid == 0) {
        sbk::release_agent();
        *ticket = sbk::atomic_add_agent(cnt + tile, 1);
      }
      __syncthreads();
      store = sbk::uniform(*ticket) == nseg - 1;
      __syncthreads();  // (the ticket word is rewritten by the next partial tile)
      if (store) {
        if (tid == 0) sbk::acquire_agent();
        __syncthreads();
        zero();
        for (int sgm = 0; sgm < nseg; ++sgm) {  // segment order = K order: the sum does not depend on who arrived last
          const float4* src =
              reinterpret_cast<const float4*>(slabs + (size_t)(2 * (x * W + p_first + sgm) + (sgm == 0 ? 1 : 0)) * (BT * BT));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = src[((wave * (TM * TN) + i * TN + j) * 4 + r4) * 64 + lane];
                acc[i][j][4 * r4] += v.x;
                acc[i][j][4 * r4 + 1] += v.y;
                acc[i][j][4 * r4 + 2] += v.z;
                acc[i][j][4 * r4 + 3] += v.w;
              }
        }
        if (tid == 0) sbk::atomic_store_agent(cnt + tile, 0);  // re-armed for the next launch on this stream
      }
    }
    if (store) epilogue(tile);
  };

  int sidx = 0, tile, lo, hi, stage = 0;
  seg_get(0, tile, lo, hi);
  int kt = lo;
  setup(tile);
  issue(kt, 0);
  zero();
  sbk::vm_drain();
  __syncthreads();
  for (;;) {
    const bool seg_ends = kt + 1 == hi;
    const bool has_next = !seg_ends || sidx + 1 < nseg_wg;
    int ntile = tile, nlo = lo, nhi = hi, nkt = kt + 1;
    if (seg_ends && has_next) {
      seg_get(sidx + 1, ntile, nlo, nhi);
      nkt = nlo;
    }
    if (has_next) {  // the next unit's panels fly while this one is multiplied
      if (seg_ends) setup(ntile);
      issue(nkt, stage ^ 1);
    }
    compute(stage, has_next, nkt);
    sbk::vm_drain();   // this wave's share of the next panels has landed ...
    __syncthreads();   // ... and everybody's; every wave is done reading `stage`
    if (seg_ends) {
      finish(tile, lo, hi, stage);
      zero();
    }
    if (!has_next) break;
    if (seg_ends) {
      ++sidx;
      tile = ntile;
      lo = nlo;
      hi = nhi;
    }
    kt = nkt;
    stage ^= 1;
  }
}

// ---------------------------------------------------------------------------
// Skinny GEMM for the decoder steps (M = beams x utterances, a few hundred rows).
// With so few rows an LDS-tiled workgroup grid cannot fill 256 CUs, and the
// weights (L2/MALL resident) dominate traffic.  Here a workgroup owns one
// (TM*32) x 32 output tile and its 4 waves own 4 K slices of it: operands go
// straight from L2 into registers as 16-byte runs (lane (r, half) reads the
// `half` side of a 32-float chunk of row r; the two k-slices of the 32x32x2
// MFMA are fed from the two halves), double-buffered in registers, no LDS and
// no barrier in the main loop.  The four partial tiles meet in LDS and wave 0
// applies the epilogue (fixed summation order => run-to-run deterministic).
// For long K (FFN2) gridDim.y adds a second, global split whose partial tiles
// are combined by splitk_reduce_kernel.
template <int NCH>  // 32-float K chunks fetched per batch (all of them in flight together)
__global__ void __launch_bounds__(256, 2) gemm_skinny_kernel(GemmArgs g, float* __restrict__ partial, int kper,
                                                          int tiles_m, int tiles_n) {
  constexpr int KC = 32;  // floats per row per chunk (16 per lane half)
  __shared__ float red[3][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware tile order: workgroup id = 8*q + x runs on XCD x (observed round-robin), so give XCD x
  // the column tiles nt = x (mod 8): every weight row is then fetched into ONE XCD's L2 only.
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;  // column tiles per XCD (last group may be ragged)
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;  // uniform per workgroup: no barrier is skipped by a subset
  }
  const int r = lane & 31, half = lane >> 5;
  const int ks = blockIdx.y * 4 + wave;
  const int k_begin = min(g.K, ks * kper), k_end = min(g.K, k_begin + kper);
  const float* wrow = g.W + (size_t)min(nt * 32 + r, g.N - 1) * g.ldw + half * (KC / 2);
  const float* arow = g.A + (size_t)min(mt * 32 + r, g.M - 1) * g.lda + half * (KC / 2);

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;

  for (int k0 = k_begin; k0 < k_end; k0 += NCH * KC) {
    float4 a[NCH][4], w[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = min(k0 + c * KC, g.K - KC);  // chunks past the slice end re-read a valid chunk and are skipped below
#pragma unroll
      for (int v = 0; v < 4; ++v) w[c][v] = *reinterpret_cast<const float4*>(wrow + k + 4 * v);
#pragma unroll
      for (int v = 0; v < 4; ++v) a[c][v] = *reinterpret_cast<const float4*>(arow + k + 4 * v);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (k0 + c * KC < k_end) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc = sbk::mfma_32x32x2(a[c][v].x, w[c][v].x, acc);
          acc = sbk::mfma_32x32x2(a[c][v].y, w[c][v].y, acc);
          acc = sbk::mfma_32x32x2(a[c][v].z, w[c][v].z, acc);
          acc = sbk::mfma_32x32x2(a[c][v].w, w[c][v].w, acc);
        }
      }
    }
  }

  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave - 1][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  const int col = nt * 32 + r;
  const bool to_partial = gridDim.y > 1;
  const float bv = (!to_partial && g.bias && col < g.N) ? g.bias[col] : 0.0f;
  float* P = partial + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    const int row = mt * 32 + rr;
    const float sum = ((acc[q] + red[0][rr][r]) + red[1][rr][r]) + red[2][rr][r];
    if (row >= g.M || col >= g.N) continue;
    if (to_partial) {
      P[(size_t)row * g.N + col] = sum;
    } else {
      float v = apply_act(sum + bv, g.act) * g.alpha;
      if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
      if (g.R) v += g.R[(size_t)row * g.ldr + col];
      g.C[(size_t)row * g.ldc + col] = v;
    }
  }
}

// The same contraction when a wave's K slice is exactly ONE fetch batch (kper == NCH*32: K = 512 with the four
// waves, K = 2048 with the four-way global split, ...).  No loop and no predicates: all 8*NCH 16-byte loads of a
// lane are issued back to back (the sched_barrier keeps the compiler from sinking them next to their MFMAs --
// in the looped kernel above it pairs every four MFMAs with a fresh load round trip, which makes a 128-deep
// slice cost ~16 dependent L2 latencies), then the MFMA chain runs off registers.
template <int NCH>
__global__ void __launch_bounds__(256, 2) gemm_skinny_flat_kernel(GemmArgs g, float* __restrict__ partial, int tiles_m,
                                                               int tiles_n) {
  constexpr int KC = 32;
  __shared__ float red[3][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;
  }
  const int r = lane & 31, half = lane >> 5;
  const int k_begin = (blockIdx.y * 4 + wave) * NCH * KC;
  const float* wrow = g.W + (size_t)min(nt * 32 + r, g.N - 1) * g.ldw + half * (KC / 2) + k_begin;
  const float* arow = g.A + (size_t)min(mt * 32 + r, g.M - 1) * g.lda + half * (KC / 2) + k_begin;
  float4 a[NCH][4], w[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int v = 0; v < 4; ++v) w[c][v] = *reinterpret_cast<const float4*>(wrow + c * KC + 4 * v);
#pragma unroll
    for (int v = 0; v < 4; ++v) a[c][v] = *reinterpret_cast<const float4*>(arow + c * KC + 4 * v);
  }
  sbk::sched_fence();
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      acc = sbk::mfma_32x32x2(a[c][v].x, w[c][v].x, acc);
      acc = sbk::mfma_32x32x2(a[c][v].y, w[c][v].y, acc);
      acc = sbk::mfma_32x32x2(a[c][v].z, w[c][v].z, acc);
      acc = sbk::mfma_32x32x2(a[c][v].w, w[c][v].w, acc);
    }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave - 1][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    v[q] = ((acc[q] + red[0][rr][r]) + red[1][rr][r]) + red[2][rr][r];
  }
  if (gridDim.y > 1) {  // partial tile of a global K split: combined by splitk_reduce_kernel
    float* P = partial + (size_t)blockIdx.y * g.M * g.N;
    const int col = nt * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      if (row < g.M && col < g.N) P[(size_t)row * g.N + col] = v[q];
    }
    return;
  }
  tile_epilogue_32x32(g, v, mt, nt, r, half);
}

// 64 x 64 tile of the same scheme for M >= ~600 rows (a 128-utterance batch, or a grouped search over several
// recipe-sized batches): the four waves still split K four ways and hold their whole slice in registers, but each
// owns a 2 x 2 block of 32x32 accumulators.  Per MFMA that halves the bytes pulled from L2 (the 32x32 kernel moves
// 128 KB per workgroup for 64 MFMAs per wave and is bound by that traffic from ~500 workgroups on: measured 2.2 us
// per extra 100 workgroups), and the four accumulator chains are independent, so no MFMA waits on its predecessor.
// LayerNorm fused into the skinny GEMM:  C = epilogue( LN(A) . W^T ) for K = NCH*128 (one fetch batch
// per wave, so the workgroup's four waves hold complete rows of A in registers).  gamma/beta are
// pre-folded into the operands by the caller:  Wf[n,k] = W[n,k]*gamma[k],  bf[n] = b[n] + sum_k W[n,k]*beta[k],
// hence  LN(x).W^T + b = rstd * ((x - mean) . Wf^T) + bf.  Row statistics are the two-pass form of
// csrc/norm.hip (mean, then sum of squared deviations), reduced across the waves through LDS.
template <int NCH>
__global__ void __launch_bounds__(256, 2) gemm_skinny_ln_kernel(GemmArgs g, float eps, int tiles_m, int tiles_n) {
  constexpr int KC = 32;
  __shared__ float red[3][32][33];
  __shared__ float stat[2][4][32];
  __shared__ float rstd_s[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nt, mt;
  {
    const int id = blockIdx.x, x = id & 7, q = id >> 3;
    const int nt8 = (tiles_n + 7) / 8;
    mt = q % tiles_m;
    nt = x + 8 * (q / tiles_m);
    if (q / tiles_m >= nt8 || nt >= tiles_n) return;
  }
  const int r = lane & 31, half = lane >> 5;
  const int k_begin = wave * NCH * KC;  // K == 4 * NCH * KC
  const float* wrow = g.W + (size_t)min(nt * 32 + r, g.N - 1) * g.ldw + half * (KC / 2);
  const float* arow = g.A + (size_t)min(mt * 32 + r, g.M - 1) * g.lda + half * (KC / 2);
  float4 a[NCH][4], w[NCH][4];
  // the activation rows first: the row statistics below wait for them only, the weight loads stay in flight
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) a[c][v] = *reinterpret_cast<const float4*>(arow + k_begin + c * KC + 4 * v);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) w[c][v] = *reinterpret_cast<const float4*>(wrow + k_begin + c * KC + 4 * v);
  sbk::sched_fence();
  // mean over the full row: lane partial -> both halves -> the four waves
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) s += (a[c][v].x + a[c][v].y) + (a[c][v].z + a[c][v].w);
  s += sbk::shfl_xor(s, 32);
  if (half == 0) stat[0][wave][r] = s;
  __syncthreads();
  const float mean = ((stat[0][0][r] + stat[0][1][r]) + (stat[0][2][r] + stat[0][3][r])) / (float)g.K;
  float q2 = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      a[c][v].x -= mean; a[c][v].y -= mean; a[c][v].z -= mean; a[c][v].w -= mean;
      q2 += (a[c][v].x * a[c][v].x + a[c][v].y * a[c][v].y) + (a[c][v].z * a[c][v].z + a[c][v].w * a[c][v].w);
    }
  q2 += sbk::shfl_xor(q2, 32);
  if (half == 0) stat[1][wave][r] = q2;

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      acc = sbk::mfma_32x32x2(a[c][v].x, w[c][v].x, acc);
      acc = sbk::mfma_32x32x2(a[c][v].y, w[c][v].y, acc);
      acc = sbk::mfma_32x32x2(a[c][v].z, w[c][v].z, acc);
      acc = sbk::mfma_32x32x2(a[c][v].w, w[c][v].w, acc);
    }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave - 1][(q & 3) + 8 * (q >> 2) + 4 * half][r] = acc[q];
  }
  __syncthreads();
  if (wave > 0) return;
  if (half == 0) {
    const float var = ((stat[1][0][r] + stat[1][1][r]) + (stat[1][2][r] + stat[1][3][r])) / (float)g.K;
    rstd_s[r] = rsqrtf(var + eps);
  }
  sbk::wave_sync();
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rr = (q & 3) + 8 * (q >> 2) + 4 * half;
    v[q] = (((acc[q] + red[0][rr][r]) + red[1][rr][r]) + red[2][rr][r]) * rstd_s[rr];
  }
  tile_epilogue_32x32(g, v, mt, nt, r, half);
}

// C = epilogue(sum_ks partial[ks]) ; fixed summation order => run-to-run deterministic.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs g, const float* __restrict__ partial, int SK) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / g.N), col = (int)(i % g.N);
    float acc = partial[i];
    for (int ks = 1; ks < SK; ++ks) acc += partial[(size_t)ks * total + i];
    float v = apply_act(acc + (g.bias ? g.bias[col] : 0.0f), g.act) * g.alpha;
    if (g.seq_len && (row % g.rows_per_seq) >= g.seq_len[row / g.rows_per_seq]) v = 0.0f;
    if (g.R) v += g.R[(size_t)row * g.ldr + col];
    g.C[(size_t)row * g.ldc + col] = v;
  }
}

}  // namespace
namespace {
template <int BM, int BN, int BK, int WM, int WN>
int launch_gemm(const GemmArgs& g, bool vec, hipStream_t st) {
  dim3 grid(sbk::cdiv(g.N, BN), sbk::cdiv(g.M, BM));
  dim3 block((BM / WM) * (BN / WN) * 64);
  static const char* kName = (BM == 256 && BN == 128) ? (WM == 128 ? "gemm_nt_256x128w" : "gemm_nt_256x128")
                             : (BM == 128 && BN == 256) ? "gemm_nt_128x256"
                             : BM == 128 ? "gemm_nt_128x128" : (BM == 64 ? "gemm_nt_64x64" : "gemm_nt_32x64");
  sbk::ProfScope prof(kName, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N), st);
  if (vec) {  // scalar LDS operand reads at pitch BK+1 (a 16-byte LDS operand variant was knob 9 in rounds 1-4: measured slower, removed)
    SBK_LAUNCH((gemm_nt_kernel<BM, BN, BK, WM, WN, true>), grid, block, 0, st, g);
  } else {
    SBK_LAUNCH((gemm_nt_kernel<BM, BN, BK, WM, WN, false>), grid, block, 0, st, g);
  }
  return sbk::launch_status("sbk_gemm_nt_f32");
}

}  // namespace

namespace sbk {
int g_skinny_off = 0;  // tuning knob: 1 = route few-row GEMMs to the LDS-tiled kernels
int g_tiled_splitk = 256;  // tuning knob (key 14): from this many rows on, K >= 2048 shapes take 64x64 LDS tiles with a
                           // 4-way K split instead of the register-operand path (0 = off).  Measured (tools/microbench.py
                           // --ffn2, N = 512, K = 2048): 160 rows 17.9 -> 21.6 us, 320: 24.2 -> 20.1, 1280: 53.0 -> 37.8,
                           // 2560: 94.4 -> 57.9; N = 768, K = 3072 at 1280 rows: 115.7 -> 63.5
// (Removed in round 5 with their kernels -- measured and lost, logs under profiles/: the reduction of the tiled split-K GEMM in
// its last-arriving workgroup (knob 36: 9 815 vs 10 416 audio-s/s, r03_last_arriver_reductions_ab.log -- an agent-scope
// release is an L2 write-back on an 8-XCD part), 64 x 64 register-operand tiles (knob 11: 29 vs 23 us in situ), the looped
// skinny schedule of round 1 (knob 10), the wider reach of the register-operand path (knob 12), other K splits (knob 15).)
int gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
            int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st);
int* tile_tickets(hipStream_t st, long tiles);  // this stream's zeroed arrival counters (nullptr: no workspace registered for the stream, or too many tiles)
int sk_route(int M, int N, int K, int* bt = nullptr);  // workgroups (and tile edge) of the persistent kernel for this shape (0: tile-grid / register-operand paths)
// Internal C++ entry shared with the fused pipelines (decoder step, encoder).
// Skinny path: M <= 512 rows, K a multiple of 64, 16-byte aligned rows.  `ws` (optional) holds the
// split-K partials: SK * M * N floats.
int gemm_nt_ws(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
               int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq,
               float* ws, size_t ws_floats, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  // (measured, tools/microbench.py --attn --gemm: from ~1.9 M outputs with a short K the LDS-tiled kernels win:
  //  M=1280 N=1536 41.6 -> 27.6 us, M=640 N=5000 63 -> 46 us; a long K still needs the split of the skinny path)
  const bool big_short = (long)M * N >= 1900000 && K <= 1024;  // (K = 768: the TransformerLM scorer's projections)
  const bool skinny_ok = !big_short && (M <= 512 || (long)cdiv(M, 128) * cdiv(N, 128) < 256) && M <= 4096 && K % 32 == 0 && lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) && aligned16(W);
  if (!skinny_ok || g_skinny_off || (aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && sk_route(M, N, K) > 0))
    return gemm_nt(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq, st);
  GemmArgs g{A, W, bias, R, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, seq_len, rows_per_seq > 0 ? rows_per_seq : 1};
  // (K = 768, the TransformerLM scorer's projections: 2-way split, 34.4 -> 24.6 us at 1280 rows; K = 512: no gain)
  const int short_sk = K < 2048 && (long)M * N < 1900000 && K >= 768 && M >= 1024 ? 2 : 0;
  if (g_tiled_splitk && ws && M >= g_tiled_splitk && (K >= 2048 || short_sk) && K % 128 == 0) {  // long K at ~1 K rows: 64x64 LDS tiles, K split 4-way
    const int SK = short_sk ? short_sk : 4, kper = K / SK;
    if ((size_t)SK * M * N <= ws_floats) {
      ProfScope prof("gemm_skinny", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
      dim3 grid(cdiv(N, 64), cdiv(M, 64), SK), block(256);
      SBK_LAUNCH((gemm_nt_splitk_kernel<64, 64, 32, 32, 32, true>), grid, block, 0, st, g, ws, kper);
      int rc = launch_status("gemm_splitk_tiled");
      if (rc) return rc;
      const size_t total = (size_t)M * N;
      SBK_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, g, (const float*)ws, SK);
      return launch_status("splitk_reduce");
    }
  }
  const int tiles_m = cdiv(M, 32), tiles_n = cdiv(N, 32);
  // a second, global K split when the tile grid alone leaves SIMDs idle (needs `ws` for the partial tiles)
  int SKg = 1;
  if (ws) {
    // (only worth the extra reduce launch for long K: K = 512 slices are 128 deep already)
    while (K / (4 * SKg) > 128 && K % (4 * SKg * 2 * 32) == 0 && tiles_m * tiles_n * SKg < 1024 &&
           (size_t)(SKg * 2) * M * N <= ws_floats)
      SKg *= 2;
  }
  const int kper = cdiv(cdiv(K, 4 * SKg), 32) * 32;
  ProfScope prof("gemm_skinny", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
  dim3 grid(8 * tiles_m * cdiv(tiles_n, 8), SKg), block(256);
  const int nch = kper >= 128 ? 4 : (kper >= 64 ? 2 : 1);
  const bool flat = K == 4 * SKg * kper && (kper == 128 || kper == 64 || kper == 32);
  if (flat) {  // one fetch batch per wave: every load in flight before the first MFMA
    if (kper == 128) {
      SBK_LAUNCH((gemm_skinny_flat_kernel<4>), grid, block, 0, st, g, ws, tiles_m, tiles_n);
    } else if (kper == 64) {
      SBK_LAUNCH((gemm_skinny_flat_kernel<2>), grid, block, 0, st, g, ws, tiles_m, tiles_n);
    } else {
      SBK_LAUNCH((gemm_skinny_flat_kernel<1>), grid, block, 0, st, g, ws, tiles_m, tiles_n);
    }
  } else if (nch >= 4) {
    SBK_LAUNCH((gemm_skinny_kernel<4>), grid, block, 0, st, g, ws, kper, tiles_m, tiles_n);
  } else if (nch >= 2) {
    SBK_LAUNCH((gemm_skinny_kernel<2>), grid, block, 0, st, g, ws, kper, tiles_m, tiles_n);
  } else {
    SBK_LAUNCH((gemm_skinny_kernel<1>), grid, block, 0, st, g, ws, kper, tiles_m, tiles_n);
  }
  int rc = launch_status("gemm_skinny");
  if (rc || SKg == 1) return rc;
  const size_t total = (size_t)M * N;
  SBK_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, g, (const float*)ws, SKg);
  return launch_status("splitk_reduce");
}

// C = epilogue(LN(A) . Wf^T + bf) with gamma/beta pre-folded into (Wf, bf); returns -1 when the shape is
// not eligible (the caller then runs LayerNorm + gemm_nt_ws with the unfolded weights).
int gemm_ln_nt(const float* A, int lda, const float* Wf, int ldw, const float* bf, const float* R, int ldr, float* C,
               int ldc, int M, int N, int K, float eps, int act, float alpha, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  const bool ok = (K == 512 || K == 256 || K == 128) && (M <= 512 || (long)cdiv(M, 128) * cdiv(N, 128) < 256) &&
                  M <= 4096 && (long)M * N < 1900000 && lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) && aligned16(Wf) &&
                  !g_skinny_off;
  if (!ok) return -1;
  GemmArgs g{A, Wf, bf, R, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, nullptr, 1};
  const int tiles_m = cdiv(M, 32), tiles_n = cdiv(N, 32);
  ProfScope prof("gemm_skinny_ln", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
  dim3 grid(8 * tiles_m * cdiv(tiles_n, 8)), block(256);
  if (K == 512) {
    SBK_LAUNCH((gemm_skinny_ln_kernel<4>), grid, block, 0, st, g, eps, tiles_m, tiles_n);
  } else if (K == 256) {
    SBK_LAUNCH((gemm_skinny_ln_kernel<2>), grid, block, 0, st, g, eps, tiles_m, tiles_n);
  } else {
    SBK_LAUNCH((gemm_skinny_ln_kernel<1>), grid, block, 0, st, g, eps, tiles_m, tiles_n);
  }
  return launch_status("gemm_skinny_ln");
}

// ---- stream-K launch: per-stream workspace (slabs + tile tickets), registered by the caller (sbk_stream_workspace_set)
int g_sk_mode = 1;        // tuning knob (key 18): 0 = tile-grid kernels only, 1 = routed by shape (sk_route), 2 = always (tests), 3 = always from 8 tiles on (A/B)
int g_sk_min_rows = 2048;  // tuning knob (key 24): fewer rows than this never take the persistent kernel in routed mode
// (Knobs 19 / 21 / 22 / 23 / 25 / 26 / 30 / 31 of rounds 3-4 -- grid size, units per workgroup, the measurement builds, the
// tail-first stagger switch, 64-wide persistent tiles, panel pieces interleaved with the MFMA groups, the split-operand
// kernel's grid -- are gone: their measured values are the constants below, their logs profiles/r03_*.)
namespace {
constexpr int kSkMaxGrid = 512, kSkMaxTiles = 1 << 16;
int sk_cus();
}
// Workgroups of the persistent kernel for this shape, 0 = the tile-grid kernels.  Measured on MI355X (tools/microbench.py
// --sk, profiles/r03_gemm_persistent_sweep.log): the persistent kernel wins once every workgroup gets about a tile's
// worth of units (T >= W per XCD keeps the leftover share small); two workgroups per CU from ~500 tiles on, one below;
// narrow short-K shapes (N <= 512, K <= 512: four column tiles, 16 K steps per tile -- the epilogue and the partial
// tiles weigh most there) stay on the tile grid below ~400 tiles (in situ -- tools/microbench.py --enc-layer,
// profiles/r03_encoder_in_situ_gemm_routing.log -- the persistent kernel already wins there from 500 tiles on).
int sk_route(int M, int N, int K, int* bt) {
  int unused;
  if (!bt) bt = &unused;
  *bt = 128;
  if (!g_sk_mode || K % 32 != 0 || K < 64) return 0;
  const long T = (long)cdiv(M, 128) * cdiv(N, 128), U = T * (K / 32);
  const int cus = sk_cus();
  int G = 0;
  if (g_sk_mode == 1 && M < g_sk_min_rows) return 0;  // (decode-step shapes: the register-operand / split-operand few-row kernels)
  if (g_sk_mode == 1) {
    const bool narrow_short = N <= 512 && K <= 512;
    if (narrow_short ? 2 * T < 3L * cus : U < 16L * cus) return 0;
    G = U >= 32L * cus ? 2 * cus : cus;
  } else {
    if (g_sk_mode == 3 && T < 8) return 0;
    G = 2 * cus;
    if (U / 4 < G) G = (int)(U / 4);  // short launches: fewer, longer ranges (>= 4 units per workgroup)
  }
  if (G > kSkMaxGrid) G = kSkMaxGrid;
  return G >= 8 ? (G / 8) * 8 : 8;  // W workgroups on each of the 8 XCDs
}
namespace {
struct SkWorkspace {
  float* slabs;
  int* cnt;
};
// Stream workspaces are CALLER-OWNED device memory (sbk_stream_workspace_set, include/sbk.h): the library allocates
// nothing.  One per (device, stream): launches of one stream are ordered, so they can share the slabs and the tickets.
std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, SkWorkspace> g_sk_ws;
int g_sk_cus[64] = {};

int cur_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}

// two partial-tile slabs per workgroup: 512 workgroups x 128 x 128 (this file's kernels) or 256 x 256 x 256 (gemm_x3p.hip)
constexpr size_t kSkSlabBytes = (size_t)2 * 256 * 256 * 256 * sizeof(float);
static_assert(kSkSlabBytes >= (size_t)2 * kSkMaxGrid * 128 * 128 * sizeof(float), "slab area");
constexpr size_t kSkTicketBytes = (size_t)kSkMaxTiles * sizeof(int);

// the stream's workspace; false when the caller has registered none for it (the tile-grid kernels run instead)
bool sk_workspace(hipStream_t st, SkWorkspace* out) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  auto it = g_sk_ws.find(std::make_pair(cur_device(), st));
  if (it == g_sk_ws.end()) return false;
  *out = it->second;
  return true;
}

}  // namespace
int* tile_tickets(hipStream_t st, long tiles) {
  SkWorkspace w;
  if (tiles > kSkMaxTiles || !sk_workspace(st, &w)) return nullptr;
  return w.cnt;
}
bool stream_ws(hipStream_t st, float** slabs, int** cnt) {  // (gemm_x3p.hip)
  SkWorkspace w;
  if (!sk_workspace(st, &w)) return false;
  *slabs = w.slabs;
  *cnt = w.cnt;
  return true;
}
namespace {

int sk_cus() {
  const int dev = cur_device() & 63;
  if (!g_sk_cus[dev]) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    g_sk_cus[dev] = cus > 0 ? cus : 256;
  }
  return g_sk_cus[dev];
}

}  // namespace
int device_cus() { return sk_cus(); }
namespace {
int launch_sk(const GemmArgs& g, int G, int bt, hipStream_t st, bool x3 = false) {
  SkArgs s;
  s.g = g;
  s.tiles_n = cdiv(g.N, bt);
  s.tiles = cdiv(g.M, bt) * s.tiles_n;
  s.KT = g.K / 32;
  if (s.tiles > kSkMaxTiles || (long)s.tiles * s.KT > (1L << 30)) return -1;
  SkWorkspace w;
  if (!sk_workspace(st, &w)) return -1;
  s.slabs = w.slabs;
  s.cnt = w.cnt;
  const size_t lds = x3 ? (size_t)2 * (128 * 32 + 128 * 48) * sizeof(float) : (size_t)(2 * 2 * bt * 32 + 4) * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, false>), (size_t)(2 * 2 * 128 * 32 + 4) * sizeof(float));
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_sk_kernel<128, true>), (size_t)2 * (128 * 32 + 128 * 48) * sizeof(float));
    once = true;
  }
  const double flops = 2.0 * g.M * g.N * g.K, bytes = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
  if (x3) {
    // (fp32-equivalent flops: the six bf16 partial products of an element pair count as ONE multiply-add)
    ProfScope prof("gemm_nt_f32x3", flops, bytes + 2.0 * (double)g.N * g.K, st);
    SBK_LAUNCH((gemm_nt_sk_kernel<128, true>), dim3((unsigned)G), dim3(256), lds, st, s);
  } else {
    ProfScope prof("gemm_nt_persistent", flops, bytes, st);
    SBK_LAUNCH((gemm_nt_sk_kernel<128, false>), dim3((unsigned)G), dim3(256), lds, st, s);
  }
  return launch_status("sbk_gemm_nt_f32 (stream-K)");
}
}  // namespace

// internal callers (the search's memory / CTC / vocabulary projections): from 1 024 rows and 192 tiles on (M = 1 280,
// N = 5 000: 64.7 vs 87.9 us; M = 640: 44.9 vs 47.4, left on the tile grid)
int g_x3_route_rows = 1024, g_x3_route_tiles = 192;  // knobs 34 / 35 (tests lower them to reach these calls with small models)
bool x3_routed(int M, int N, int K) {
  return M >= g_x3_route_rows && K % 32 == 0 && K >= 64 && N % 4 == 0 && (long)cdiv(M, 128) * cdiv(N, 128) >= g_x3_route_tiles;
}

// fp32 contraction with a pre-split W (sbk_split_bf16x3) on the bf16 matrix pipe; -1: no workspace for this stream
int gemm_nt_x3(const float* A, int lda, const uint16_t* W3, const float* bias, const float* R, int ldr, float* C, int ldc,
               int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  GemmArgs g{A, reinterpret_cast<const float*>(W3), bias, R, C, lda, 0, ldr, ldc, M, N, K, act, alpha, seq_len,
             rows_per_seq > 0 ? rows_per_seq : 1};
  const long T = (long)cdiv(M, 128) * cdiv(N, 128), U = T * (K / 32);
  const int cus = sk_cus();
  // Measured on MI355X (tools/microbench.py --x3, profiles/r03_f32x3_sweep.log): two workgroups per CU from two tiles per
  // CU on, one below (M = 4 032: N = 1 024 31 vs 50 us, N = 1 536 59 vs 69 us with one)
  int G = T >= 2L * cus ? 2 * cus : cus;
  if (U / 4 < G) G = (int)(U / 4);  // short launches: fewer, longer ranges
  if (G > kSkMaxGrid) G = kSkMaxGrid;
  G = G >= 8 ? (G / 8) * 8 : 8;
  return launch_sk(g, G, 128, st, true);
}

int gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
            int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  GemmArgs g{A, W, bias, R, C, lda, ldw, ldr, ldc, M, N, K, act, alpha, seq_len, rows_per_seq > 0 ? rows_per_seq : 1};
  const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && aligned16(A) && aligned16(W);
  // Tile choice: keep >= ~1 workgroup per CU where the problem allows it.
  const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128);
  const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
  if (vec) {
    int bt = 128;
    const int G = sk_route(M, N, K, &bt);
    if (G > 0) {
      const int rc = launch_sk(g, G, bt, st);
      if (rc != -1) return rc;  // -1: the caller registered no workspace for this stream
    }
  }
  // (256 x 128, 128 x 256 and 64-deep-K variants of the 128 x 128 tile -- knob 6 of rounds 1-2 -- never beat it: removed in round 5)
  if (tiles128 >= 384) return launch_gemm<128, 128, 32, 64, 64>(g, vec, st);
  if (tiles64 >= 256 || M > 256) return launch_gemm<64, 64, 32, 32, 32>(g, vec, st);
  return launch_gemm<32, 64, 32, 32, 32>(g, vec, st);
}
}  // namespace sbk

extern "C" size_t sbk_stream_workspace_bytes(void) { return sbk::kSkSlabBytes + sbk::kSkTicketBytes; }

extern "C" int sbk_stream_workspace_set(sbk_stream_t stream, void* workspace, size_t workspace_bytes) {
  SBK_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "stream workspace: null or not 256-byte aligned");
  SBK_REQUIRE(workspace_bytes >= sbk_stream_workspace_bytes(), "stream workspace: %zu bytes given, %zu needed", workspace_bytes,
              sbk_stream_workspace_bytes());
  hipStream_t st = sbk::as_stream(stream);
  sbk::SkWorkspace w{reinterpret_cast<float*>(workspace),
                     reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + sbk::kSkSlabBytes)};
  // tickets start at zero (ordered before the first launch on this stream); every launch leaves them at zero
  const hipError_t e = hipMemsetAsync(w.cnt, 0, sbk::kSkTicketBytes, st);
  if (e != hipSuccess) return sbk::fail((int)e, "stream workspace: memset: %s", hipGetErrorString(e));
  std::lock_guard<std::mutex> lk(sbk::g_sk_mu);
  sbk::g_sk_ws[std::make_pair(sbk::cur_device(), st)] = w;
  return 0;
}

extern "C" int sbk_stream_workspace_release(sbk_stream_t stream) {
  std::lock_guard<std::mutex> lk(sbk::g_sk_mu);
  sbk::g_sk_ws.erase(std::make_pair(sbk::cur_device(), sbk::as_stream(stream)));
  return 0;
}

extern "C" int sbk_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                               const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                               float alpha, const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(A && W && C, "gemm: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  // lda < K is allowed: rows of A then overlap (a strided window over a time-major signal = a 1-D convolution read in place)
  SBK_REQUIRE(lda > 0 && ldw >= K && ldc >= N, "gemm: leading dimension smaller than the row");
  SBK_REQUIRE(!residual || ldr >= N, "gemm: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm: seq_len given without rows_per_seq");
  return sbk::gemm_nt_ws(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq,
                         nullptr, 0, sbk::as_stream(stream));
}

extern "C" int sbk_gemm_ln_nt_f32(const float* A, int lda, const float* Wf, int ldw, const float* bf,
                                  const float* residual, int ldr, float* C, int ldc, int M, int N, int K, float eps,
                                  int act, float alpha, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(A && Wf && C, "gemm_ln: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "gemm_ln: bad shape");
  const int rc = sbk::gemm_ln_nt(A, lda, Wf, ldw, bf, residual, ldr, C, ldc, M, N, K, eps, act, alpha,
                                 sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_ln: shape M=%d N=%d K=%d not eligible for the fused kernel", M, N, K);
  return rc;
}

extern "C" int sbk_gemm_nt_splitk_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                                      const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                                      float alpha, float* workspace, size_t workspace_floats, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(A && W && C, "gemm: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  SBK_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gemm: leading dimension smaller than the row");
  SBK_REQUIRE(!residual || ldr >= N, "gemm: residual stride");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm: unknown activation %d", act);
  return sbk::gemm_nt_ws(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, alpha, nullptr, 0, workspace,
                         workspace ? workspace_floats : 0, sbk::as_stream(stream));
}


// Measurement helper: `iters` back-to-back launches of the same contraction between two events on
// `stream` (host launch overhead amortised); returns the mean time per launch in microseconds.
extern "C" int sbk_prof_gemm_repeat_f32(const float* A, const float* W, float* C, int M, int N, int K, float* workspace,
                                        size_t workspace_floats, int iters, float* us_per_launch,
                                        sbk_stream_t stream) {
  SBK_REQUIRE(A && W && C && us_per_launch && iters > 0, "gemm_repeat: bad arguments");
  hipStream_t st = sbk::as_stream(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return sbk::fail(1, "event create");
  int rc = 0;
  for (int i = 0; i < 3 && !rc; ++i)
    rc = sbk::gemm_nt_ws(A, K, W, K, nullptr, nullptr, 0, C, N, M, N, K, SBK_ACT_NONE, 1.0f, nullptr, 0, workspace,
                         workspace_floats, st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters && !rc; ++i)
    rc = sbk::gemm_nt_ws(A, K, W, K, nullptr, nullptr, 0, C, N, M, N, K, SBK_ACT_NONE, 1.0f, nullptr, 0, workspace,
                         workspace_floats, st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = ms * 1000.0f / iters;
  return rc;
}

// the tuning / measurement switch behind a key (nullptr: no such key)
static int* knob_slot(int key) {
  switch (key) {
    case 2: return &sbk::g_skinny_off;
    case 4: return &sbk::g_cross_rows;
    case 8: return &sbk::g_cross_fc256;
    case 14: return &sbk::g_tiled_splitk;
    case 18: return &sbk::g_sk_mode;
    case 24: return &sbk::g_sk_min_rows;
    case 34: return &sbk::g_x3_route_rows;
    case 35: return &sbk::g_x3_route_tiles;
    case 40: return &sbk::g_score_fused;
    case 41: return &sbk::g_x3r_mode;
    case 42: return &sbk::g_x3r_min_rows;
    case 45: return &sbk::g_x3r_ln;
    case 47: return &sbk::g_persist;
    case 48: return &sbk::g_persist_grid;
    case 49: return &sbk::g_persist_stamps;
    case 51: return &sbk::g_x3r_xc;
    case 53: return &sbk::g_nt_mask;
    case 54: return &sbk::g_x3r_probe;
    case 55: return &sbk::g_self_anc;
    case 58: return &sbk::g_x3r_pair;
    case 59: return &sbk::g_persist_tree;
    case 60: return &sbk::g_attn_exp2;
    case 61: return &sbk::g_lp256;
    case 62: return &sbk::g_lp256_mode;
    case 63: return &sbk::g_x3p_fast_epi;
    case 64: return &sbk::g_x3p_mode;
    default: return nullptr;
  }
}
extern "C" void sbk_prof_set_knob(int key, int value) {
  if (int* p = knob_slot(key)) *p = value;
}
// current value of a switch (a test restores what it found); INT_MIN for an unknown key
extern "C" int sbk_prof_get_knob(int key) {
  const int* p = knob_slot(key);
  return p ? *p : (-2147483647 - 1);
}


// ---- fp32 contraction on the bf16 matrix pipe (exact three-way operand split) ----------------------------------
extern "C" int sbk_split_bf16x3(const float* W, int ldw, uint16_t* W3, int N, int K, sbk_stream_t stream) {
  if (N == 0) return 0;
  SBK_REQUIRE(W && W3 && N > 0 && K > 0 && K % 32 == 0 && ldw >= K, "split_bf16x3: bad arguments (K must be a multiple of 32)");
  const long n = (long)N * K, blocks = (n + 255) / 256;
  SBK_LAUNCH(split_bf16x3_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, sbk::as_stream(stream), W, ldw,
             reinterpret_cast<unsigned short*>(W3), n, K);
  return sbk::launch_status("sbk_split_bf16x3");
}

extern "C" int sbk_gemm_nt_f32x3(const float* A, int lda, const uint16_t* W3, const float* bias, const float* residual,
                                 int ldr, float* C, int ldc, int M, int N, int K, int act, float alpha,
                                 const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(A && W3 && C, "gemm_f32x3: null operand");
  SBK_REQUIRE(M >= 0 && N >= 0 && K >= 64 && K % 32 == 0, "gemm_f32x3: bad shape M=%d N=%d K=%d (K: a multiple of 32, >= 64)", M, N, K);
  SBK_REQUIRE(lda > 0 && lda % 4 == 0 && ldc >= N && sbk::aligned16(A) && sbk::aligned16(W3),
              "gemm_f32x3: operand rows must be 16-byte aligned (lda=%d)", lda);
  SBK_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && sbk::aligned16(C) && (!bias || sbk::aligned16(bias)),
              "gemm_f32x3: N and ldc must be multiples of 4, C / bias 16-byte aligned (rows are stored as 16-byte vectors)");
  SBK_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && sbk::aligned16(residual)), "gemm_f32x3: residual stride / alignment");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_f32x3: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm_f32x3: seq_len given without rows_per_seq");
  const int rc = sbk::gemm_nt_x3(A, lda, W3, bias, residual, ldr, C, ldc, M, N, K, act, alpha, seq_len, rows_per_seq,
                                 sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_f32x3: no workspace registered for this stream (sbk_stream_workspace_set) or too many tiles");
  return rc;
}
