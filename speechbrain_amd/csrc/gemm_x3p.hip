// fp32 contractions on the bf16 matrix pipe with BOTH operands pre-split ("x3p"): the large contractions of the Conformer
// encoder (nnet/attention.py:623,735 in_proj / out_proj, Conformer.py:129,155 macaron feed-forward, :310-330 pointwise
// convolutions) at 2 000 - 24 000 rows.
//
// Arithmetic: exactly sbk_gemm_nt_f32x3's (csrc/gemm.hip) -- x = hi + mid + lo with three bf16 pieces (exact), six of the
// nine partial products per element pair on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest first.  What changes
// is WHERE the pieces are made and HOW the tile is fed.  Measured on the f32x3 kernel (profiles/r03_f32x3_measurement_
// builds.log, M 12 800 x N 2 048 x K 512): 179 us in all, 120 us with five sixths of the MFMAs removed, 64 us of pure MFMA
// time -- the kernel's time was the SUM of its matrix time and of everything else (LDS-DMA issue 31 us, operand split 27 us,
// operand fetch + barriers 42 us, epilogue 19 us): with 128 x 128 tiles and 24 MFMAs per wave between barriers nothing
// overlapped.  Here:
//   * the A operand arrives pre-split too (sbk_split_x3p, or written in that form by the producing kernel's epilogue): no
//     VALU work in the loop, and an A element is split ONCE instead of once per column tile (16 x at N = 2 048);
//   * operands live in HBM in PANEL layout [rows/64][K/16][3 pieces][2 k-halves][64 rows][8 k] bf16: every 1 KB chunk is
//     one global_load_lds_dwordx4 wave-instruction (contiguous source, lane-linear LDS image) AND is read back as MFMA
//     fragments (lane = row, 16 bytes = the lane's 8 k) by ds_read_b128 with consecutive rows in consecutive 16-byte slots:
//     conflict-free without a swizzle;
//   * 256 x 256 (or 256 x 128) tiles, eight waves of 64 x 128 (64 x 64), one 16-deep K step per stage: 48 (24) MFMAs per
//     wave and stage against 18 (12) fragment fetches and 6 (4.5) LDS-DMA pieces -- 0.6 x the DMA issues and 0.64 x the
//     fetches per MFMA of the 128 x 128 kernel;
//   * a three-slot LDS ring (3 x 48 KB) with the DMA two stages ahead, counted vmcnt waits and raw s_barrier (MI355X guide,
//     "Pipelining across barriers": a __syncthreads() would drain the DMA queue at every barrier);
//   * the two waves of every SIMD (wave w and w + 4) run HALF A STAGE APART: between two barriers one of them issues its
//     DMA pieces and fetches its fragments while the other owns the matrix pipe with 48 back-to-back MFMAs
//     (MI355X_MICROARCH.md, "Two waves per SIMD": complementary phases, separated by s_barrier).
// Tiles are scheduled like gemm_nt_sk_kernel's: a fixed grid, XCD x owns a contiguous range of tiles, its workgroups take
// whole tiles round-robin and share the leftover tiles stream-K style (partial tiles through slabs, last arriver sums in K
// order: run-to-run deterministic, nobody waits).  (Round 5 measured the tile space cut into 2 / 4 column groups, so that an
// XCD's eighth of the tile indices is a block of rows x one column group and its L2 keeps that group's W panels: 155.6 / 160.9 us
// against 156.3 at (12 800, 2 048, 512), 135.5 / 136.0 against 135.0 at (14 016, 1 536, 512), 101.3 / 102.2 against 99.6 at
// (14 016, 1 024, 512), profiles/r05_q_*: the kernel is not bound by what crosses the fabric.  Not kept.)
#include "common.h"
#include "internal.h"

#include <algorithm>
#include <type_traits>

using sbk::cdiv;
using sbk::f32x16;

namespace sbk {
int g_x3p_fast_epi = 1;  // key 63
int g_x3p_mode = 0;  // key 64
// csrc/gemm.hip: the caller-registered stream workspace (slabs of kX3pSlabFloats floats each, tile tickets)
bool stream_ws(hipStream_t st, float** slabs, int** cnt);
int device_cus();
}  // namespace sbk

namespace {

constexpr int kChunk = 1024;          // bytes: 64 rows x 8 bf16
constexpr int kChunkFloats = 256;     // the same in floats (LDS offsets are kept in floats)
constexpr int kKbBytes = 6 * kChunk;  // one row block's chunks of one 16-deep K step: 3 pieces x 2 halves

// ---------------------------------------------------------------------------------------------------------------
// X [rows, K] fp32 (row stride ldx floats; ldx < K allowed: a sliding window) -> panel image.  One thread = one row and
// one half k-step (8 consecutive k): two 16-byte loads, three 16-byte stores; a wave = 64 rows of one row block, so every
// store instruction writes one whole 1 KB chunk.  Rows past `rows` (the padding up to a multiple of 64) are written as
// zeros: the contraction may read them, and 0 x anything finite must stay 0.
__global__ void __launch_bounds__(256) split_x3p_kernel(const float* __restrict__ X, int ldx, uint4* __restrict__ P, int rows,
                                                        int KB, int vec) {
  const int tid = threadIdx.x, r = tid & 63;
  const int rb = blockIdx.x;
  const int row = rb * 64 + r;
  const bool live = row < rows;
  const float* xr = X + (size_t)(live ? row : 0) * ldx;
  for (int u = blockIdx.y * 4 + (tid >> 6); u < 2 * KB; u += gridDim.y * 4) {  // u = kb * 2 + half
    const int kb = u >> 1, h = u & 1;
    float x[8];
    if (live) {
      if (vec) {  // (uniform: 16-byte aligned rows)
        const float4 a = *reinterpret_cast<const float4*>(xr + u * 8), b = *reinterpret_cast<const float4*>(xr + u * 8 + 4);
        x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = xr[u * 8 + e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.0f;
    }
    unsigned hi[4], mi[4], lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // x = hi + mid + lo exactly: 8 significand bits each, remainders exact in fp32
      hi[p] = sbk::bf16_pair(x[2 * p], x[2 * p + 1]);
      const float r0 = x[2 * p] - __uint_as_float(hi[p] << 16), r1 = x[2 * p + 1] - __uint_as_float(hi[p] & 0xffff0000u);
      mi[p] = sbk::bf16_pair(r0, r1);
      lo[p] = sbk::bf16_pair(r0 - __uint_as_float(mi[p] << 16), r1 - __uint_as_float(mi[p] & 0xffff0000u));
    }
    uint4* dst = P + ((size_t)(rb * KB + kb) * 6 + h) * 64 + r;  // chunk (rb, kb, piece 0, h), slot r; pieces are 2 chunks apart
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[128] = make_uint4(mi[0], mi[1], mi[2], mi[3]);
    dst[256] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

struct X3pArgs {
  const float* PA;  // panel image of A [M, K] (addressed in floats: a chunk is 256 floats)
  const float* PW;  // panel image of W [N, K]
  const float* bias;
  const float* R;
  float* C;          // fp32 result [M, N] (may be null when PC is given)
  uint2* PC;         // optional: the result as the panel image of a [M, N] matrix (the next contraction's A operand)
  int ldr, ldc, M, N, K, act;
  float alpha;
  const int32_t* seq_len;
  int rows_per_seq;
  float* slabs;
  int* cnt;
  int tiles_n, tiles, KT;  // KT = K / 16 stages per tile
  int whole;               // 1: gridDim.x == tiles, workgroup b runs tile b (no stream-K)
  int fast_epi;            // key 63: 1 (default) = the hot epilogue forms as straight-line code, 0 = the generic form always
};

// WM x WN waves (= 8), each TM x TN sub-tiles of 32 x 32.
// MODE: 0 = the kernel; measurement builds (key 64, tools/microbench.py --x3p-modes; garbage results): bit 0 = no LDS-DMA after a
// segment's first two stages, 1 = no MFMAs, 2 = no epilogue, 3 = no fragment fetches.
template <int WM, int WN, int TM, int TN, int MODE>
__global__ void __launch_bounds__(512, 2) gemm_nt_x3p_kernel(X3pArgs s) {
  constexpr bool kNoDma = MODE & 1, kNoMfma = MODE & 2, kNoEpi = MODE & 4, kNoFetch = MODE & 8;
  static_assert(WM * WN == 8, "eight waves: two per SIMD, half a stage apart");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RBA = BM / 64, RBW = BN / 64;            // row blocks (chunks of 64 rows) of the A / W panels of a tile
  constexpr int NPA = RBA * 6, NP = NPA + RBW * 6;       // 1 KB pieces per stage
  constexpr int STAGE = NP * kChunkFloats;               // floats
  constexpr int NPW = (NP + 7) / 8;                      // pieces per wave and stage (the last one may be missing)
  SBK_DYN_LDS(float, lds);                               // [3][STAGE] (ONE LDS object)
  const float* const PA = s.PA;
  const float* const PW = s.PW;
  const float* const gbias = s.bias;
  const float* const gR = s.R;
  float* const gC = s.C;
  uint2* const gPC = s.PC;
  const int ldr = s.ldr, ldc = s.ldc, M = s.M, N = s.N, act = s.act;
  const float alpha = s.alpha;
  const int32_t* const seq_len = s.seq_len;
  const int rows_per_seq = s.rows_per_seq;
  float* const slabs = s.slabs;
  int* const cnt = s.cnt;
  const int tiles_n = s.tiles_n, KT = s.KT;
  const int rbA_max = (M + 63) / 64 - 1, rbW_max = (N + 63) / 64 - 1;

  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int group = wave >> 2;  // waves w and w + 4 share a SIMD: group 1 runs half a stage behind group 0
  const int wm = wave % WM, wn = wave / WM;
  const int wrow0 = wm * TM * 32, wcol0 = wn * TN * 32;
  const int lrow = lane & 31, half = lane >> 5;

  // ---- this workgroup's segments (gemm_nt_sk_kernel's scheme with KT = K / 16 units per tile)
  int nfull = 0, nt = 0, tileA = 0, loA = 0, hiA = 0, hiB = 0, t0 = 0, W = 1, j = 0, x = 0, tb = 0, ubase = 0, urem = 0;
  if (s.whole) {
    nfull = 1;
    t0 = blockIdx.x;
  } else {
    W = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    t0 = (int)((long)s.tiles * x / 8);
    const int t1 = (int)((long)s.tiles * (x + 1) / 8);
    nfull = sbk::uniform((t1 - t0) / W);
    const int Rl = (t1 - t0) - nfull * W;
    tb = t0 + nfull * W;
    const int UT = Rl * KT;
    ubase = sbk::uniform(UT / W), urem = UT - ubase * W;
    const int q0 = j * ubase + min(j, urem), q1 = q0 + ubase + (j < urem ? 1 : 0);
    if (q1 > q0) {
      const int ta = sbk::uniform(q0 / KT);
      tileA = tb + ta;
      loA = q0 - ta * KT;
      hiA = min(KT, loA + (q1 - q0));
      hiB = q1 - (ta + 1) * KT;
      nt = hiB > 0 ? 2 : 1;
    }
  }
  const int nseg = nfull + nt;
  if (nseg == 0) return;
  auto seg_get = [&](int sidx, int& tile, int& lo, int& hi) SBK_INLINE_LAMBDA {
    if (sidx < nfull) {
      tile = s.whole ? t0 : t0 + sidx * W + j;
      lo = 0;
      hi = KT;
    } else if (sidx == nfull) {
      tile = tileA;
      lo = loA;
      hi = hiA;
    } else {
      tile = tileA + 1;
      lo = 0;
      hi = hiB;
    }
  };
  auto owner = [&](int q) SBK_INLINE_LAMBDA {  // workgroup (index within the XCD) that owns leftover unit q
    const int big = urem * (ubase + 1);
    return sbk::uniform(q < big ? q / (ubase + 1) : urem + (q - big) / max(ubase, 1));
  };
  const int pid = x * W + j;  // slab owner id

  // ---- loader: piece q = wave + 8 i of a stage; q < NPA: A chunk (row block q / 6, piece-half q % 6), else W.  A chunk's
  // address is wave-uniform (scalar registers, scalar arithmetic); the lanes share one offset register
  const float* src[NPW];
  const unsigned lane16 = (unsigned)lane * 16u;
  auto setup = [&](int tile) SBK_INLINE_LAMBDA {
    const int rb0 = (tile / tiles_n) * RBA, cb0 = (tile % tiles_n) * RBW;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = wave + 8 * i;
      if (q < NPA) {  // (row blocks past the matrix re-read its last one: their outputs are never stored)
        const int rb = min(rb0 + q / 6, rbA_max);
        src[i] = PA + ((size_t)rb * KT * 6 + (q % 6)) * kChunkFloats;
      } else {
        const int qq = min(q - NPA, RBW * 6 - 1), cb = min(cb0 + qq / 6, rbW_max);
        src[i] = PW + ((size_t)cb * KT * 6 + (qq % 6)) * kChunkFloats;
      }
    }
  };
  const int npw = sbk::uniform((NP - wave + 7) / 8);  // pieces THIS wave issues per stage
  auto issue = [&](int kt, int slot) SBK_INLINE_LAMBDA {
    float* base = lds + slot * STAGE + wave * kChunkFloats;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (i < NPW - 1 || npw == NPW) sbk::glds16_uniform(src[i] + (size_t)kt * (6 * kChunkFloats), lane16, base + 8 * i * kChunkFloats);
  };
  auto wait_keep_one_stage = [&]() SBK_INLINE_LAMBDA {  // all of this wave's pieces but the newest stage's have landed
    if (npw == NPW) {
      sbk::vm_wait<NPW>();
    } else {
      sbk::vm_wait<(NPW > 1 ? NPW - 1 : 0)>();
    }
  };

  f32x16 acc[TM][TN];
  auto zero = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;
  };

  // fragment addresses within a stage: ONE base per operand and compile-time offsets (A sub-tile i, piece p -> chunk (row
  // block, p, half), slot = row in block).  The wave's 64 rows are one row block, its columns TN / 2 column blocks.
  static_assert(TM == 2 && TN % 2 == 0, "a wave owns whole 64-row blocks of both operands");
  const int abase = ((wrow0 >> 6) * 6 + half) * kChunkFloats + lrow * 4;
  const int wbase = (NPA + (wcol0 >> 6) * 6 + half) * kChunkFloats + lrow * 4;

  sbk::bf16x8 af[3][TM], wf[3][TN];
  if constexpr (kNoFetch) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[p][i] = sbk::opaque_zero<sbk::bf16x8>();
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) wf[p][jj] = sbk::opaque_zero<sbk::bf16x8>();
    }
  }
  auto fetch = [&](int slot) SBK_INLINE_LAMBDA {
    const float* sa = lds + slot * STAGE + abase;
    const float* sw = lds + slot * STAGE + wbase;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
        wf[p][jj] = *reinterpret_cast<const sbk::bf16x8*>(sw + ((jj >> 1) * 6 + p * 2) * kChunkFloats + (jj & 1) * 128);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[p][i] = *reinterpret_cast<const sbk::bf16x8*>(sa + p * 2 * kChunkFloats + i * 128);
    }
  };
  auto multiply = [&]() SBK_INLINE_LAMBDA {
    // smallest terms first; consecutive MFMAs go to different accumulators.  W is the FIRST operand: a lane owns one row m
    // of C and registers 4g .. 4g+3 hold four consecutive columns (16-byte epilogue vectors), as in gemm_nt_sk_kernel<X3>
    constexpr int PW_[6] = {0, 2, 1, 0, 1, 0}, PA_[6] = {2, 0, 1, 1, 0, 0};  // (W piece, A piece): hi.lo lo.hi mid.mid hi.mid mid.hi hi.hi
    if constexpr (kNoMfma) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) sbk::keep(af[p][i]);
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) sbk::keep(wf[p][jj]);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) acc[i][jj] = sbk::mfma_32x32x16_bf16(wf[PW_[t]][jj], af[PA_[t]][i], acc[i][jj]);
  };

  auto epilogue_generic = [&](int tile) SBK_INLINE_LAMBDA {
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const bool interior = m0 + BM <= M && n0 + BN <= N;  // uniform: no per-element predicates
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = m0 + wrow0 + i * 32 + lrow, rowc = min(row, M - 1);
      const bool row_ok = interior || row < M;
      const bool masked = seq_len && (rowc % rows_per_seq) >= seq_len[rowc / rows_per_seq];
      const float ra = masked ? 0.0f : alpha;
      float* crow = gC ? gC + (size_t)rowc * ldc : nullptr;
      const float* rrow = gR ? gR + (size_t)rowc * ldr : nullptr;
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) {
        float4 bv[4], rv[4];
        bool ok[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // (N % 4 == 0: a vector is inside the matrix or outside as a whole)
          const int col = n0 + wcol0 + jj * 32 + 8 * g + 4 * half;
          ok[g] = row_ok && (interior || col < N);
          bv[g] = (gbias && ok[g]) ? *reinterpret_cast<const float4*>(gbias + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          rv[g] = (rrow && ok[g]) ? *reinterpret_cast<const float4*>(rrow + col) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[4 * g] = acc[i][jj][4 * g] + bv[g].x;
          v[4 * g + 1] = acc[i][jj][4 * g + 1] + bv[g].y;
          v[4 * g + 2] = acc[i][jj][4 * g + 2] + bv[g].z;
          v[4 * g + 3] = acc[i][jj][4 * g + 3] + bv[g].w;
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
          const int col = n0 + wcol0 + jj * 32 + 8 * g + 4 * half;
          const float o0 = masked ? rv[g].x : v[4 * g] * ra + rv[g].x, o1 = masked ? rv[g].y : v[4 * g + 1] * ra + rv[g].y;
          const float o2 = masked ? rv[g].z : v[4 * g + 2] * ra + rv[g].z, o3 = masked ? rv[g].w : v[4 * g + 3] * ra + rv[g].w;
          if (ok[g] && crow) *reinterpret_cast<float4*>(crow + col) = make_float4(o0, o1, o2, o3);
          if (gPC && row_ok && (interior || col < N)) {
            // the result as the NEXT contraction's A operand (its K = this N): columns col .. col+3 are k = col .. col+3 of row
            // `row`: chunk (row / 64, col / 16, piece, (col % 16) / 8), slot row % 64, bytes (col % 8) * 2 .. +8
            const unsigned h0 = sbk::bf16_pair(o0, o1), h1 = sbk::bf16_pair(o2, o3);
            const float r0 = o0 - __uint_as_float(h0 << 16), r1 = o1 - __uint_as_float(h0 & 0xffff0000u);
            const float r2 = o2 - __uint_as_float(h1 << 16), r3 = o3 - __uint_as_float(h1 & 0xffff0000u);
            const unsigned m0_ = sbk::bf16_pair(r0, r1), m1_ = sbk::bf16_pair(r2, r3);
            const unsigned l0 = sbk::bf16_pair(r0 - __uint_as_float(m0_ << 16), r1 - __uint_as_float(m0_ & 0xffff0000u));
            const unsigned l1 = sbk::bf16_pair(r2 - __uint_as_float(m1_ << 16), r3 - __uint_as_float(m1_ & 0xffff0000u));
            const int KBn = N >> 4;
            uint2* d = gPC + (((size_t)(row >> 6) * KBn + (col >> 4)) * 6 + ((col >> 3) & 1)) * 128 + (row & 63) * 2 + ((col >> 2) & 1);
            d[0] = make_uint2(h0, h1);
            d[256] = make_uint2(m0_, m1_);
            d[512] = make_uint2(l0, l1);
          }
        }
      }
    }
  };

  // The hot forms of the epilogue as straight-line code (round 6): activation, residual, outputs and the edge predicates are compile-time,
  // the bias vectors and the wave's whole residual block are requested before the first sub-tile is touched.  The generic form above has a
  // run-time switch and exec-mask branches around every load and store; the compiler serialises it into dependent request -> wait ->
  // store round trips (189 branches, 13 full s_waitcnt vmcnt(0) per tile): 28-36 % of the launch at K = 512 (profiles/r06_ae_*).
  // The arithmetic per element is the generic form's, operation for operation (unmasked rows: v * alpha + residual, residual 0 when absent).
  auto epilogue_fast = [&](int tile, auto act_c, auto r_c, auto c_c, auto pc_c, auto edge_c) SBK_INLINE_LAMBDA {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool HAS_R = decltype(r_c)::value, HAS_C = decltype(c_c)::value, HAS_PC = decltype(pc_c)::value, EDGE = decltype(edge_c)::value;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    float4 bv[TN][4];
#pragma unroll
    for (int jj = 0; jj < TN; ++jj)
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[jj][g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (gbias) {  // (uniform; the vectors are used unconditionally)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + wcol0 + jj * 32 + 8 * g + 4 * half;
          bv[jj][g] = *reinterpret_cast<const float4*>(gbias + (EDGE ? min(col, N - 4) : col));
        }
    }
    float4 rv[TM][TN][4];
    if constexpr (HAS_R) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = m0 + wrow0 + i * 32 + lrow;
        const float* rrow = gR + (size_t)(EDGE ? min(row, M - 1) : row) * ldr;
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = n0 + wcol0 + jj * 32 + 8 * g + 4 * half;
            rv[i][jj][g] = *reinterpret_cast<const float4*>(rrow + (EDGE ? min(col, N - 4) : col));
          }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = m0 + wrow0 + i * 32 + lrow;
      const bool row_ok = !EDGE || row < M;
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) {
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[4 * g] = acc[i][jj][4 * g] + bv[jj][g].x;
          v[4 * g + 1] = acc[i][jj][4 * g + 1] + bv[jj][g].y;
          v[4 * g + 2] = acc[i][jj][4 * g + 2] + bv[jj][g].z;
          v[4 * g + 3] = acc[i][jj][4 * g + 3] + bv[jj][g].w;
        }
        if constexpr (ACT == SBK_ACT_SWISH) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
        } else {
          static_assert(ACT == SBK_ACT_NONE, "x3p: activation not instantiated in the straight-line epilogue");
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + wcol0 + jj * 32 + 8 * g + 4 * half;
          float4 r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if constexpr (HAS_R) r4 = rv[i][jj][g];
          const float o0 = v[4 * g] * alpha + r4.x, o1 = v[4 * g + 1] * alpha + r4.y;
          const float o2 = v[4 * g + 2] * alpha + r4.z, o3 = v[4 * g + 3] * alpha + r4.w;
          if (!EDGE || (row_ok && col < N)) {
            if constexpr (HAS_C) *reinterpret_cast<float4*>(gC + (size_t)row * ldc + col) = make_float4(o0, o1, o2, o3);
            if constexpr (HAS_PC) {  // (the panel image of the result: see the generic form)
              const unsigned h0 = sbk::bf16_pair(o0, o1), h1 = sbk::bf16_pair(o2, o3);
              const float r0 = o0 - __uint_as_float(h0 << 16), r1 = o1 - __uint_as_float(h0 & 0xffff0000u);
              const float r2 = o2 - __uint_as_float(h1 << 16), r3 = o3 - __uint_as_float(h1 & 0xffff0000u);
              const unsigned m0_ = sbk::bf16_pair(r0, r1), m1_ = sbk::bf16_pair(r2, r3);
              const unsigned l0 = sbk::bf16_pair(r0 - __uint_as_float(m0_ << 16), r1 - __uint_as_float(m0_ & 0xffff0000u));
              const unsigned l1 = sbk::bf16_pair(r2 - __uint_as_float(m1_ << 16), r3 - __uint_as_float(m1_ & 0xffff0000u));
              const int KBn = N >> 4;
              uint2* d = gPC + (((size_t)(row >> 6) * KBn + (col >> 4)) * 6 + ((col >> 3) & 1)) * 128 + (row & 63) * 2 + ((col >> 2) & 1);
              d[0] = make_uint2(h0, h1);
              d[256] = make_uint2(m0_, m1_);
              d[512] = make_uint2(l0, l1);
            }
          }
        }
      }
    }
  };
  auto epilogue = [&](int tile) SBK_INLINE_LAMBDA {
    using std::integral_constant;
    using T = std::true_type;
    using F = std::false_type;
    if (s.fast_epi && !seq_len) {  // (uniform)
      const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
      const bool edge = !(m0 + BM <= M && n0 + BN <= N);
      constexpr integral_constant<int, SBK_ACT_NONE> kNone{};
      constexpr integral_constant<int, SBK_ACT_SWISH> kSwish{};
      if (act == SBK_ACT_SWISH && !gR && !gC && gPC) {  // first feed-forward projection -> the second one's panel operand
        if (edge) epilogue_fast(tile, kSwish, F{}, F{}, T{}, T{}); else epilogue_fast(tile, kSwish, F{}, F{}, T{}, F{});
        return;
      }
      if (act == SBK_ACT_NONE && gR && gC && !gPC) {  // out-projection, second feed-forward projection, second pointwise convolution
        if (edge) epilogue_fast(tile, kNone, T{}, T{}, F{}, T{}); else epilogue_fast(tile, kNone, T{}, T{}, F{}, F{});
        return;
      }
      if (act == SBK_ACT_NONE && !gR && gC && !gPC) {  // q/k/v projection, first pointwise convolution
        if (edge) epilogue_fast(tile, kNone, F{}, T{}, F{}, T{}); else epilogue_fast(tile, kNone, F{}, T{}, F{}, F{});
        return;
      }
      if (act == SBK_ACT_SWISH && !gR && gC && !gPC) {
        if (edge) epilogue_fast(tile, kSwish, F{}, T{}, F{}, T{}); else epilogue_fast(tile, kSwish, F{}, T{}, F{}, F{});
        return;
      }
    }
    epilogue_generic(tile);
  };

  // a K range [lo, hi) of `tile` is complete in acc (every wave of the workgroup is here, both groups aligned)
  auto finish = [&](int tile, int lo, int hi) SBK_INLINE_LAMBDA {
    int* ticket = reinterpret_cast<int*>(lds + 2 * STAGE);  // slot 2 is idle between two segments
    bool store = true;
    if (lo != 0 || hi != KT) {  // partial: publish the slab, take a ticket; the last ticket sums the tile's slabs
      const int p_first = owner((tile - tb) * KT), p_last = owner((tile - tb + 1) * KT - 1);
      const int nsegs = p_last - p_first + 1;
      float4* mine = reinterpret_cast<float4*>(slabs + (size_t)(2 * pid + (lo == 0 ? 1 : 0)) * (BM * BN));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            mine[((wave * (TM * TN) + i * TN + jj) * 4 + r4) * 64 + lane] =
                make_float4(acc[i][jj][4 * r4], acc[i][jj][4 * r4 + 1], acc[i][jj][4 * r4 + 2], acc[i][jj][4 * r4 + 3]);
      sbk::vm_drain();
      __syncthreads();
      if (tid == 0) {
        sbk::release_agent();
        *ticket = sbk::atomic_add_agent(cnt + tile, 1);
      }
      __syncthreads();
      store = sbk::uniform(*ticket) == nsegs - 1;
      __syncthreads();
      if (store) {
        if (tid == 0) sbk::acquire_agent();
        __syncthreads();
        zero();
        for (int sg = 0; sg < nsegs; ++sg) {  // segment order = K order: the sum does not depend on who arrived last
          const float4* sp =
              reinterpret_cast<const float4*>(slabs + (size_t)(2 * (x * W + p_first + sg) + (sg == 0 ? 1 : 0)) * (BM * BN));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = sp[((wave * (TM * TN) + i * TN + jj) * 4 + r4) * 64 + lane];
                acc[i][jj][4 * r4] += v.x;
                acc[i][jj][4 * r4 + 1] += v.y;
                acc[i][jj][4 * r4 + 2] += v.z;
                acc[i][jj][4 * r4 + 3] += v.w;
              }
        }
        if (tid == 0) sbk::atomic_store_agent(cnt + tile, 0);  // re-armed for the next launch on this stream
      }
    }
    if constexpr (kNoEpi) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) sbk::pin(acc[i][jj]);
    } else {
      if (store) epilogue(tile);
      // (the epilogue's stores retired in the compiler's books as well: with stores "in flight" it put an s_waitcnt vmcnt(0) of its own
      // behind the LDS-DMA issue of every stage -- sbk::vm_drain_visible; the segment start waits for everything anyway)
      sbk::vm_drain_visible();
    }
  };

  // the barrier between two phases.  The scheduling fences pin it: MFMAs touch no memory, so the scheduler would otherwise
  // hoist the barrier that ENDS a matrix phase above the 48 MFMAs (seen in the ISA: two barriers back to back, then the
  // MFMAs and the next fetch phase in one interval -- the two groups then take turns instead of overlapping)
  auto phase_barrier = [&]() SBK_INLINE_LAMBDA {
    sbk::sched_fence();
    sbk::block_barrier_raw();
    sbk::sched_fence();
  };
  // ---- main loop: per segment, a two-stage-ahead LDS-DMA pipeline; group 1 one barrier behind group 0
  int tile, lo, hi;
  seg_get(0, tile, lo, hi);
  setup(tile);
  issue(lo, 0);
  if (hi - lo > 1) issue(lo + 1, 1);
  zero();
  for (int sidx = 0; sidx < nseg; ++sidx) {
    const int ns = hi - lo;
    sbk::vm_drain();          // stages 0 and 1 of this segment (and the previous tile's stores)
    sbk::block_barrier_raw();  // ... everybody's
    if (group == 1) sbk::block_barrier_raw();
#pragma unroll 1
    for (int n = 0; n < ns; ++n) {
      // ---- fetch phase (the SIMD's other wave multiplies meanwhile)
      const int slot = n % 3;
      if constexpr (!kNoDma) {
        if (n + 2 < ns) issue(lo + n + 2, (n + 2) % 3);  // its slot was last read in stage n - 1: two barriers ago for both groups
      }
      if constexpr (!kNoFetch) fetch(slot);
      sbk::lds_drain();
      if (!kNoDma && n + 2 < ns) {
        wait_keep_one_stage();  // this wave's share of stage n + 1 has landed (stage n + 2 may fly on)
      } else {
        sbk::vm_drain();
      }
      phase_barrier();
      // ---- matrix phase
      multiply();
      phase_barrier();
    }
    if (group == 0) sbk::block_barrier_raw();
    // both groups aligned, nobody reads LDS: the next segment's first two stages fly during this one's epilogue
    int ntile = tile, nlo = lo, nhi = hi;
    if (sidx + 1 < nseg) {
      seg_get(sidx + 1, ntile, nlo, nhi);
      setup(ntile);
      issue(nlo, 0);
      if (nhi - nlo > 1) issue(nlo + 1, 1);
    }
    finish(tile, lo, hi);
    zero();
    tile = ntile, lo = nlo, hi = nhi;
  }
}

template <int WM, int WN, int TM, int TN, int MODE>
int launch_x3p(const X3pArgs& a0, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (size_t)3 * ((BM / 64 + BN / 64) * 6) * kChunk;
  X3pArgs a = a0;
  a.fast_epi = sbk::g_x3p_fast_epi;
  a.tiles_n = cdiv(a.N, BN);
  a.tiles = cdiv(a.M, BM) * a.tiles_n;
  a.KT = a.K / 16;
  const int cus = sbk::device_cus();
  int G;
  if (a.tiles <= cus) {  // one whole tile per workgroup
    a.whole = 1;
    G = a.tiles;
  } else {
    a.whole = 0;
    G = (cus / 8) * 8;
    if (!sbk::stream_ws(st, &a.slabs, &a.cnt)) return -1;
    if (a.tiles > (1 << 16)) return -1;
  }
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_x3p_kernel<WM, WN, TM, TN, MODE>), lds);
    once = true;
  }
  SBK_LAUNCH((gemm_nt_x3p_kernel<WM, WN, TM, TN, MODE>), dim3((unsigned)G), dim3(512), lds, st, a);
  return sbk::launch_status("sbk_gemm_nt_x3p");
}

}  // namespace

namespace sbk {
// (256 x 256 tiles -- knob 39 of round 4 -- were slower than 256 x 128 on every encoder shape, e.g. 179 vs 155 us at N = 2 048:
// profiles/r04_d_*; instantiation and knob removed in round 5)

int gemm_nt_x3p(const uint16_t* PA, const uint16_t* PW, const float* bias, const float* R, int ldr, float* C, int ldc,
                uint16_t* PC, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  X3pArgs a{};
  a.PA = reinterpret_cast<const float*>(PA);
  a.PW = reinterpret_cast<const float*>(PW);
  a.bias = bias, a.R = R, a.C = C, a.PC = reinterpret_cast<uint2*>(PC);
  a.ldr = ldr, a.ldc = ldc, a.M = M, a.N = N, a.K = K, a.act = act, a.alpha = alpha;
  a.seq_len = seq_len, a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1;
  const double flops = 2.0 * M * N * K;
  // algorithmic bytes: both operand images once (6 B per element) + the result (+ the residual)
  const double bytes = 6.0 * ((double)M * K + (double)N * K) + (C ? 4.0 : 0.0) * M * N + (PC ? 6.0 : 0.0) * M * N + (R ? 4.0 : 0.0) * M * N;
  ProfScope prof("gemm_nt_x3p", flops, bytes, st);
  switch (g_x3p_mode) {  // key 64: measurement builds
    case 1: return launch_x3p<4, 2, 2, 2, 1>(a, st);
    case 2: return launch_x3p<4, 2, 2, 2, 2>(a, st);
    case 4: return launch_x3p<4, 2, 2, 2, 4>(a, st);
    case 8: return launch_x3p<4, 2, 2, 2, 8>(a, st);
    default: return launch_x3p<4, 2, 2, 2, 0>(a, st);
  }
}
}  // namespace sbk

extern "C" size_t sbk_x3p_panel_bytes(int rows, int K) {
  if (rows <= 0 || K <= 0) return 0;
  return (size_t)((rows + 63) / 64) * 64 * (size_t)K * 6;
}

extern "C" int sbk_split_x3p(const float* X, int ldx, uint16_t* P, int rows, int K, sbk_stream_t stream) {
  if (rows == 0) return 0;
  SBK_REQUIRE(X && P && rows > 0 && K >= 16 && K % 16 == 0 && ldx > 0, "split_x3p: bad arguments (K: a multiple of 16)");
  SBK_REQUIRE(sbk::aligned16(P), "split_x3p: the panel image must be 16-byte aligned");
  const int RB = (rows + 63) / 64, KB = K / 16;
  const int gy = std::min(std::max(1, (2 * KB + 3) / 4), std::max(1, 2048 / RB));
  SBK_LAUNCH(split_x3p_kernel, dim3((unsigned)RB, (unsigned)gy), dim3(256), 0, sbk::as_stream(stream), X, ldx,
             reinterpret_cast<uint4*>(P), rows, KB, (ldx % 4 == 0 && sbk::aligned16(X)) ? 1 : 0);
  return sbk::launch_status("sbk_split_x3p");
}

extern "C" int sbk_gemm_nt_x3p(const uint16_t* PA, const uint16_t* PW, const float* bias, const float* residual, int ldr,
                               float* C, int ldc, uint16_t* PC, int M, int N, int K, int act, float alpha,
                               const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream) {
  if (M == 0 || N == 0) return 0;
  SBK_REQUIRE(PA && PW && (C || PC), "gemm_x3p: null operand");
  SBK_REQUIRE(M > 0 && N > 0 && K >= 32 && K % 16 == 0, "gemm_x3p: bad shape M=%d N=%d K=%d (K: a multiple of 16, >= 32)", M, N, K);
  SBK_REQUIRE(sbk::aligned16(PA) && sbk::aligned16(PW), "gemm_x3p: panel images must be 16-byte aligned");
  SBK_REQUIRE(N % 4 == 0 && (!C || (ldc >= N && ldc % 4 == 0 && sbk::aligned16(C))) && (!bias || sbk::aligned16(bias)),
              "gemm_x3p: N and ldc must be multiples of 4, C / bias 16-byte aligned (rows are stored as 16-byte vectors)");
  SBK_REQUIRE(!PC || (N % 16 == 0 && sbk::aligned16(PC)), "gemm_x3p: a panel result needs N %% 16 == 0");
  SBK_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && sbk::aligned16(residual)), "gemm_x3p: residual stride / alignment");
  SBK_REQUIRE(act >= SBK_ACT_NONE && act <= SBK_ACT_LEAKY_RELU, "gemm_x3p: unknown activation %d", act);
  SBK_REQUIRE(!seq_len || rows_per_seq > 0, "gemm_x3p: seq_len given without rows_per_seq");
  const int rc = sbk::gemm_nt_x3p(PA, PW, bias, residual, ldr, C, ldc, PC, M, N, K, act, alpha, seq_len, rows_per_seq,
                                  sbk::as_stream(stream));
  if (rc == -1) return sbk::fail(SBK_EINVAL, "gemm_x3p: no workspace registered for this stream (sbk_stream_workspace_set) or too many tiles");
  return rc;
}
