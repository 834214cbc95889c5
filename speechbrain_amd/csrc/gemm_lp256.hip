// The large reduced-precision contractions (bf16 x bf16 and e4m3 x e4m3 activations x weights, both row-major in HBM) on 256 x 256
// tiles: the Whisper large-v3 encoder's four projections per layer at 12 000 rows (BASELINE configs[4]; reference: the Linear
// layers of integrations/huggingface/whisper.py:318-353 under the inference autocast of inference/interfaces.py:295-298).
//
// Why a second kernel beside gemm_lp.hip's 128 x 128 one (round 6).  That kernel gives every wave a 64 x 64 sub-tile: per 128-byte
// K tile a wave fetches 16 KB of fragments from LDS for 512 cycles of MFMA work, the four waves of the two resident workgroups
// 128 KB per 1 024 cycles of each SIMD's matrix time -- exactly the 128 bytes per cycle the LDS delivers, before the 64 KB the
// LDS-DMA writes into it: the kernel sits on the LDS port (614 TF/s bf16, 940 TF/s fp8: 0.25 / 0.19 of the matrix peaks), and every
// operand byte crosses the L2 -> LDS path once per 128 rows of the other operand.  Here:
//   * a wave owns 128 x 64 (4 x 2 sub-tiles of 32 x 32): 12 KB of fragments per 512 MFMA cycles -- 0.75 of the LDS reads per
//     MFMA; a workgroup tile of 256 x 256 halves the bytes every operand element costs on the way into LDS;
//   * eight waves in the two groups of gemm_nt_x3p_kernel (wave w and w + 4 share a SIMD and run HALF A STEP APART: between two
//     barriers one issues its LDS-DMA pieces and fetches its fragments while the other owns the matrix pipe with 16 (bf16) / 8
//     (fp8) back-to-back MFMAs; MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md "The 256^2 8-phase template");
//   * a step = half a K tile (64 bytes of every row); a K tile (128 bytes per row: whole cache lines of both operands) is one
//     LDS-DMA batch of eight pieces per wave into a ring of two 64 KB slots, issued the moment its slot's previous tenant has
//     been fetched by both groups (one K tile = four phases ahead of its first use), waited for with the counted
//     vmcnt two barriers before the first read (the one-barrier-more rule for groups a barrier apart);
//   * the LDS image is gemm_lp.hip's: 128-byte rows, the 16-byte slot index XORed with (row >> 1) & 7 on the SOURCE side of the
//     DMA, so that a ds_read_b128 of 32 consecutive rows is conflict-free;
//   * tiles in bands of eight tile rows, column-major inside a band, XCD x owns a contiguous range and its workgroups take
//     consecutive tiles round by round: the 32 tiles an XCD works on at a time are 8 x 4 -- 12 operand blocks in its L2 instead
//     of the 2 + tiles_n of a row-major order;
//   * persistent over whole tiles, the K tiles of consecutive output tiles one stream through the ring (the next tile's first K tile
//     lands under this tile's last steps); the epilogue turns the
//     accumulators through LDS and writes whole lines (see there).
// Arithmetic, epilogue and outputs are gemm_nt_bf16dma_kernel's / gemm_nt_fp8dma_kernel's (fp32 accumulation; per-row scales of
// both fp8 operands applied to the accumulators; bias, activation, alpha, fp32 residual; fp32 / bf16 / e4m3 outputs): the same sums
// in the same order per output element (K ascending, one accumulator), so the two kernels agree bit for bit.
#include "common.h"
#include "internal.h"

#include <type_traits>

using sbk::f32x16;

namespace sbk {
int device_cus();
int g_lp256 = 1;  // key 61: 1 (default) = shapes with >= 128 (96: fp32 result + residual) tiles of 256 x 256 take this kernel, 0 = never, 2 = always (tests)
}  // namespace sbk

namespace {

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

constexpr int kRowF = 32;                 // floats per 128-byte LDS row
constexpr int kPanelF = 256 * kRowF;      // one operand's 256 rows of a K tile
constexpr int kSlotF = 2 * kPanelF;       // A panel | W panel: 64 KB
constexpr int kStgPitch = 72, kStgF = 16 * kStgPitch;  // a wave's epilogue staging: 16 rows x 64 columns fp32, pitch 72 floats (8 x 4.5 KB: inside slot 1)
static_assert(8 * kStgF <= kSlotF, "the staging area is one slot");
constexpr size_t kLdsBytes = (size_t)2 * kSlotF * sizeof(float);

// MODE: 0 = the kernel.  Measurement builds (key 62, tools/microbench.py --lp256-modes; results are garbage, timings are what they are
// for): bit 0 = no LDS-DMA after a tile's first two K tiles, 1 = no MFMAs, 2 = no epilogue, 3 = no fragment fetches.  Schedule variants
// (same results): bit 4 = the wave raises its priority for its MFMA phase, bit 5 = the LDS reads are awaited AFTER the phase barrier
// (group 1 in the second half of a K tile keeps the early wait: those reads are the last ones of the slot the other group refills
// behind that barrier).
template <bool FP8, int MODE, int ACT>
__global__ void __launch_bounds__(512, 2) gemm_nt_lp256_kernel(sbk::Lp256Args s) {
  constexpr bool kNoDma = MODE & 1, kNoMfma = MODE & 2, kNoEpi = MODE & 4, kNoFetch = MODE & 8, kPrio = MODE & 16, kLateDrain = MODE & 32;
  SBK_DYN_LDS(float, lds);  // [2 slots][A 256 rows | W 256 rows][128 bytes]  (ONE LDS object)
  const unsigned char* const gA = s.A;
  const unsigned char* const gW = s.W;
  const float* const gsa = s.sa;
  const float* const gsw = s.sw;
  const float* const gbias = s.bias;
  const float* const gR = s.R;
  float* const gC = s.C;
  unsigned short* const gCb = s.Cb;
  unsigned char* const gC8 = s.C8;
  const int ldr = s.ldr, ldc = s.ldc, ldcb = s.ldcb, ldc8 = s.ldc8, M = s.M, N = s.N;
  const long lda = s.lda, ldw = s.ldw;  // bytes
  const float alpha = s.alpha, c8_inv = 1.0f / s.c8_scale;
  const int tiles_m = s.tiles_m, tiles_n = s.tiles_n, KT = s.KT;

  const int tid = threadIdx.x, lane = tid & 63, wave = sbk::uniform(tid >> 6);
  const int group = wave >> 2;  // waves w and w + 4 share a SIMD: group 1 runs half a step (one barrier) behind group 0
  const int wrow0 = (wave & 1) * 128, wcol0 = (wave >> 1) * 64;
  const int lrow = lane & 31, half = lane >> 5, sw = (lrow >> 1) & 7;

  // ---- this workgroup's tiles: every Wx-th tile of the XCD's contiguous range (band order, see tile_origin)
  int t_first, t_stride, ntile;
  if (s.whole) {
    t_first = blockIdx.x, t_stride = 0, ntile = 1;
  } else {
    const int Wx = gridDim.x >> 3, x = blockIdx.x & 7, j = blockIdx.x >> 3;  // gridDim.x is a multiple of 8
    const int t0 = (int)((long)s.tiles * x / 8), t1 = (int)((long)s.tiles * (x + 1) / 8);
    t_first = t0 + j, t_stride = Wx;
    ntile = sbk::uniform(t0 + j < t1 ? (t1 - t0 - j + Wx - 1) / Wx : 0);
  }
  if (ntile == 0) return;
  auto tile_origin = [&](int t, int& m0, int& n0) SBK_INLINE_LAMBDA {  // bands of 8 tile rows, column-major inside a band
    const int per_band = 8 * tiles_n, band = t / per_band, within = t - band * per_band;
    const int bh = min(8, tiles_m - band * 8);
    m0 = sbk::uniform((band * 8 + within % bh) * 256);
    n0 = sbk::uniform((within / bh) * 256);
  };

  // ---- loader: a piece = 8 rows x 128 bytes (lane l: row l >> 3, LDS slot l & 7, which holds the row's source slot
  // (l & 7) ^ ((row >> 1) & 7)); wave w loads rows 32 w .. 32 w + 31 of both panels.  Scalar tile base + one 32-bit offset per
  // lane and piece (rows past the matrix re-read its last row: their outputs are never stored)
  unsigned aoff[4], woff[4];
  const unsigned char* abase = gA;
  const unsigned char* wbase = gW;
  auto setup = [&](int t) SBK_INLINE_LAMBDA {
    int m0, n0;
    tile_origin(t, m0, n0);
    abase = gA + (size_t)m0 * lda;
    wbase = gW + (size_t)n0 * ldw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave * 4 + i) * 8 + (lane >> 3);
      const unsigned so = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
      aoff[i] = (unsigned)(min(r, M - 1 - m0) * lda) + so;
      woff[i] = (unsigned)(min(r, N - 1 - n0) * ldw) + so;
    }
  };
  auto issue = [&](int kt, int slot) SBK_INLINE_LAMBDA {
    float* dst = lds + slot * kSlotF + wave * (4 * 8 * kRowF);
    const float* ab = reinterpret_cast<const float*>(abase + (size_t)kt * 128);
    const float* wb = reinterpret_cast<const float*>(wbase + (size_t)kt * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) sbk::glds16_uniform(ab, aoff[i], dst + i * (8 * kRowF));
#pragma unroll
    for (int i = 0; i < 4; ++i) sbk::glds16_uniform(wb, woff[i], dst + kPanelF + i * (8 * kRowF));
  };

  f32x16 acc[4][2];
  auto zero = [&]() SBK_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;
  };

  // ---- fragments of one step (64 bytes of every row of the wave's 128 + 64 rows): 12 ds_read_b128, 48 registers
  using Frag = std::conditional_t<FP8, uint4, sbk::bf16x8>;
  Frag fa[2][4], fw[2][2];  // [16-byte piece][sub-tile]: bf16 = k step (8 k per lane), fp8 = low / high half of the lane's 32 bytes
  if constexpr (kNoFetch) {  // (defined operands the optimiser cannot see through)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[p][i] = sbk::opaque_zero<Frag>();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) fw[p][jj] = sbk::opaque_zero<Frag>();
    }
  }
  auto fetch = [&](int slot, int h) SBK_INLINE_LAMBDA {
    const float* sa = lds + slot * kSlotF + (wrow0 + lrow) * kRowF;
    const float* sb = lds + slot * kSlotF + kPanelF + (wcol0 + lrow) * kRowF;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      // bf16: k step g = 2 h + p, the lane's 8 k are source slot 2 g + half; fp8: the lane's 32 bytes are source slots 4 h + 2 half + p
      const int src = FP8 ? 4 * h + 2 * half + p : 2 * (2 * h + p) + half;
      const int so = (src ^ sw) * 4;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) fw[p][jj] = *reinterpret_cast<const Frag*>(sb + jj * 32 * kRowF + so);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[p][i] = *reinterpret_cast<const Frag*>(sa + i * 32 * kRowF + so);
    }
  };
  auto multiply = [&]() SBK_INLINE_LAMBDA {
    if constexpr (kNoMfma) {  // (the fetched fragments stay live)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sbk::keep(fa[p][i]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) sbk::keep(fw[p][jj]);
      }
    } else if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[i][jj] = sbk::mfma_32x32x64_fp8(sbk::i32x8_from_u4(fa[0][i], fa[1][i]), sbk::i32x8_from_u4(fw[0][jj], fw[1][jj]), acc[i][jj]);
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = sbk::mfma_32x32x16_bf16(fa[p][i], fw[p][jj], acc[i][jj]);
    }
  };

  // acc[i][jj][r]: row m0 + wrow0 + 32 i + (r & 3) + 8 (r >> 2) + 4 half, column n0 + wcol0 + 32 jj + lrow.  Written from that layout
  // (a lane = a column) the result leaves as 4-byte / 2-byte stores and the residual arrives as 4-byte loads, eight-odd dependent
  // batches per tile with the matrix pipe idle: measured 41-48 us of a 138 / 80-us launch (profiles/r06_z_*).  So a wave turns each
  // 16 x 64 block through its own 4.5 KB of LDS (the slot of the tile's last K tile; pitch 72 floats: the two
  // half-waves of a ds_write_b32 land 32 banks apart) and handles it by rows: a lane owns four consecutive columns -- 16-byte
  // residual loads, one block ahead of the block being written, and 16 / 8 / 4-byte stores that complete whole 128-byte lines.
  // The arithmetic per element is the 128 x 128 kernels': (acc * (row scale * column scale) + bias) -> activation -> * alpha -> + residual.
  // Straight-line code: the activation is a template parameter, interior tiles carry no per-lane predicates, the residual's loads are
  // unconditional on clamped addresses and the four LDS reads of a unit are issued together -- the first form of this epilogue had a
  // run-time switch and exec-mask branches around every load and store, and the compiler serialised it into 32 dependent
  // read -> wait -> store round trips per wave (12 us per tile).
  auto epilogue_tile = [&](int t, int stg_slot, auto edge_c, auto res_c) SBK_INLINE_LAMBDA {
    constexpr bool EDGE = decltype(edge_c)::value, HAS_R = decltype(res_c)::value;
    int m0, n0;
    tile_origin(t, m0, n0);
    float* const stg = lds + stg_slot * kSlotF + wave * kStgF;
    const int rq = lane >> 4, c4 = (lane & 15) * 4;
    const int col = n0 + wcol0 + c4;
    const bool col_ok = !EDGE || col < N;        // (N % 4 == 0: a vector is inside the matrix or outside as a whole)
    const int colc = EDGE ? min(col, N - 4) : col;  // an address inside the matrix for the lanes past its edge
    float4 bv4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cs4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (gbias) bv4 = *reinterpret_cast<const float4*>(gbias + colc);
    if constexpr (FP8) {
      if (gsw) cs4 = *reinterpret_cast<const float4*>(gsw + colc);
    }
    float4 rv[2][4];
    auto load_r = [&](int u, float4 (&dst)[4]) SBK_INLINE_LAMBDA {  // unit u: rows 16 u .. 16 u + 15 of the wave's 128
      if constexpr (HAS_R) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = m0 + wrow0 + 16 * u + 4 * it + rq;
          dst[it] = *reinterpret_cast<const float4*>(gR + (size_t)(EDGE ? min(row, M - 1) : row) * ldr + colc);
        }
      }
    };
    load_r(0, rv[0]);
    static_for<0, 8>([&](auto uc) SBK_INLINE_LAMBDA {  // (compile-time indices: the accumulators stay in registers)
      constexpr int u = decltype(uc)::value, i = u >> 1, hh = u & 1;
      if constexpr (u + 1 < 8) load_r(u + 1, rv[(u + 1) & 1]);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8)  // registers 8 hh .. 8 hh + 7: rows 16 hh + (r & 3) + 8 ((r >> 2) & 1) + 4 half of the block
          stg[((r8 & 3) + 8 * (r8 >> 2) + 4 * half) * kStgPitch + jj * 32 + lrow] = acc[i][jj][8 * hh + r8];
      sbk::wave_sync();  // (a wave's LDS operations execute in order)
      float4 a4[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) a4[it] = *reinterpret_cast<const float4*>(stg + (4 * it + rq) * kStgPitch + c4);
      sbk::wave_sync();
      float rs4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if constexpr (FP8) {
        if (gsa) {  // (uniform)
#pragma unroll
          for (int it = 0; it < 4; ++it) rs4[it] = gsa[min(m0 + wrow0 + 16 * u + 4 * it + rq, M - 1)];
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = m0 + wrow0 + 16 * u + 4 * it + rq;
        const float4 r4 = rv[u & 1][it];
        const float av[4] = {a4[it].x, a4[it].y, a4[it].z, a4[it].w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
        (void)rr;
        const float bvv[4] = {bv4.x, bv4.y, bv4.z, bv4.w}, csv[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
        float v[4];
        if constexpr (FP8) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = av[e] * (rs4[it] * csv[e]) + bvv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = av[e] + bvv[e];
        }
        if constexpr (ACT == SBK_ACT_SWISH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
        } else if constexpr (ACT == SBK_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = sbk::gelu_erfc(v[e]);
        } else {
          static_assert(ACT == SBK_ACT_NONE, "lp256: activation not instantiated");
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = v[e] * alpha;
          if constexpr (HAS_R) o[e] += rr[e];
        }
        if (!EDGE || (col_ok && row < M)) {
          if (gC) *reinterpret_cast<float4*>(gC + (size_t)row * ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
          if (gCb) *reinterpret_cast<uint2*>(gCb + (size_t)row * ldcb + col) = make_uint2(sbk::bf16_pair(o[0], o[1]), sbk::bf16_pair(o[2], o[3]));
          if constexpr (FP8) {
            if (gC8)
              *reinterpret_cast<unsigned*>(gC8 + (size_t)row * ldc8 + col) =
                  (unsigned)sbk::f32x2_to_fp8(o[0] * c8_inv, o[1] * c8_inv) | ((unsigned)sbk::f32x2_to_fp8(o[2] * c8_inv, o[3] * c8_inv) << 16);
          }
        }
      }
    });
  };
  auto epilogue = [&](int t, int stg_slot) SBK_INLINE_LAMBDA {
    int m0, n0;
    tile_origin(t, m0, n0);
    // (four straight-line variants.  With the residual behind a run-time `if (gR)` the compiler saw a path on which its loads are
    // requested and never consumed, and -- their registers being the main loop's fragment registers -- put an s_waitcnt vmcnt(0) in
    // front of the loop's first ds_read: every K tile then waited for the LDS-DMA issued just before it; bf16 1 290 -> 820 TF/s at
    // 8 192^3, profiles/r06_ac_*)
    const bool interior = m0 + 256 <= M && n0 + 256 <= N;  // (uniform)
    if (gR) {
      if (interior) {
        epilogue_tile(t, stg_slot, std::false_type{}, std::true_type{});
      } else {
        epilogue_tile(t, stg_slot, std::true_type{}, std::true_type{});
      }
    } else {
      if (interior) {
        epilogue_tile(t, stg_slot, std::false_type{}, std::false_type{});
      } else {
        epilogue_tile(t, stg_slot, std::true_type{}, std::false_type{});
      }
    }
  };

  // the barrier between two phases, pinned by scheduling fences (gemm_nt_x3p_kernel: MFMAs touch no memory, the scheduler would
  // hoist the barrier that ENDS a matrix phase above its MFMAs and the two groups would take turns instead of overlapping)
  auto phase_barrier = [&]() SBK_INLINE_LAMBDA {
    sbk::sched_fence();
    sbk::block_barrier_raw();
    sbk::sched_fence();
  };

  // ---- main loop.  The K tiles of this workgroup's tiles form ONE stream n = 0, 1, ... through the two slots (slot n & 1): the first K
  // tile of the next output tile is just the next K tile.  Step (n, h): [at h = 0 issue K tile n + 1 into the other slot -- its previous
  // tenant n - 1 was last fetched in step (n - 1, 1), which ended two physical barriers ago for this group and one for the other; behind
  // an output tile's end the slot was that tile's epilogue staging area, released by the barrier after the epilogue] fetch; drain
  // the LDS reads; [h = 1: this wave's share of K tile n + 1 has landed -- read two program barriers later, i.e. at least one physical
  // barrier after the OTHER group's wait]; barrier; MFMAs; barrier.
  setup(t_first);
  issue(0, 0);
  zero();
  sbk::vm_drain();           // K tile 0
  sbk::block_barrier_raw();  // ... everybody's share of it
  if (group == 1) sbk::block_barrier_raw();
  int n = 0;  // K tiles consumed so far (uniform)
  for (int ord = 0; ord < ntile; ++ord) {
    const int t = t_first + ord * t_stride;
#pragma unroll 1
    for (int m = 0; m < KT; ++m, ++n) {
      const int slot = n & 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (!kNoDma) {
          if (h == 0) {
            if (m + 1 < KT) {
              issue(m + 1, slot ^ 1);
            } else if (ord + 1 < ntile) {  // (this tile's last K tile was issued a K tile ago: the loader moves on to the next tile)
              setup(t + t_stride);
              issue(0, slot ^ 1);
            }
          }
        }
        if constexpr (!kNoFetch) fetch(slot, h);
        const bool early = !kLateDrain || (h == 1 && group == 1);  // (uniform)
        if (early) sbk::lds_drain();
        if (h == 1) sbk::vm_drain();
        phase_barrier();
        if (!early) sbk::lds_drain();
        if constexpr (kPrio) sbk::set_prio<1>();
        multiply();
        if constexpr (kPrio) sbk::set_prio<0>();
        phase_barrier();
      }
    }
    if (group == 0) sbk::block_barrier_raw();
    // both groups aligned, nobody reads the slot of this tile's last K tile any more: it is the epilogue's staging area (the other slot
    // holds the next tile's first K tile, landed and published above)
    if constexpr (kNoEpi) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) sbk::pin(acc[i][jj]);
    } else {
      epilogue(t, (n - 1) & 1);
      // (the epilogue's stores retired in the compiler's books too: see vm_drain_visible.  The next tile's first K tile landed before
      // the epilogue; what this waits for is the stores' acknowledgement -- measured neutral against not waiting, visits AA / AB)
      sbk::vm_drain_visible();
    }
    zero();
    if (ord + 1 < ntile) {
      sbk::lds_drain();
      sbk::block_barrier_raw();  // every wave is done with the staging area: the next tile's second K tile may land there
      if (group == 1) sbk::block_barrier_raw();
    }
  }
}

template <bool FP8, int MODE, int ACT>
int launch_lp256(const sbk::Lp256Args& a0, hipStream_t st) {
  sbk::Lp256Args a = a0;
  a.tiles_m = sbk::cdiv(a.M, 256);
  a.tiles_n = sbk::cdiv(a.N, 256);
  a.tiles = a.tiles_m * a.tiles_n;
  const int cus = sbk::device_cus();
  int G;
  if (a.tiles <= cus) {  // one tile per workgroup
    a.whole = 1;
    G = a.tiles;
  } else {
    a.whole = 0;
    G = (cus / 8) * 8;
  }
  static bool once = false;
  if (!once) {
    (void)SBK_ALLOW_DYN_LDS((gemm_nt_lp256_kernel<FP8, MODE, ACT>), kLdsBytes);
    once = true;
  }
  SBK_LAUNCH((gemm_nt_lp256_kernel<FP8, MODE, ACT>), dim3((unsigned)G), dim3(512), kLdsBytes, st, a);
  return sbk::launch_status(FP8 ? "sbk_gemm_nt_fp8a" : "sbk_gemm_nt_bf16a");
}

}  // namespace

namespace sbk {
bool lp256_routed(const Lp256Args& a) {
  if (g_lp256 == 0 || a.KT < 2) return false;
  if (a.act != SBK_ACT_NONE && a.act != SBK_ACT_GELU && a.act != SBK_ACT_SWISH) return false;  // (the instantiated epilogues)
  // the epilogue's vectors: four consecutive columns per lane
  const auto al = [](const void* p, uintptr_t n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if (a.N % 4 != 0 || !al(a.bias, 16) || !al(a.sw, 16) || (a.R && (a.ldr % 4 != 0 || !al(a.R, 16))) || (a.C && (a.ldc % 4 != 0 || !al(a.C, 16))) ||
      (a.Cb && (a.ldcb % 4 != 0 || !al(a.Cb, 8))) || (a.C8 && (a.ldc8 % 4 != 0 || !al(a.C8, 4))))
    return false;
  if (g_lp256 == 2) return true;
  // from how many tiles on (profiles/r06_ai_*: 36 ... 480 tiles): 128 in general -- below, the 128 x 128 kernels' four times as many
  // workgroups fill the chip better --, 96 for an fp32 result with a residual, whose rows the 128 x 128 kernels move as 4-byte accesses
  // (6 000 x 1 280 x 1 280, 120 tiles: 41 against 80 us)
  return (long)cdiv(a.M, 256) * cdiv(a.N, 256) >= ((a.C && a.R) ? 96 : 128);
}
int g_lp256_mode = 0;  // key 62: MODE of gemm_nt_lp256_kernel (bf16 operands only; measurement builds 1 / 2 / 4 / 8, schedule variants 16 / 32 / 48)
int gemm_nt_lp256(const Lp256Args& a, bool fp8, hipStream_t st) {
  if (fp8) {
    switch (a.act) {
      case SBK_ACT_GELU: return launch_lp256<true, 0, SBK_ACT_GELU>(a, st);
      case SBK_ACT_SWISH: return launch_lp256<true, 0, SBK_ACT_SWISH>(a, st);
      default: return launch_lp256<true, 0, SBK_ACT_NONE>(a, st);
    }
  }
  if (a.act == SBK_ACT_GELU) return launch_lp256<false, 0, SBK_ACT_GELU>(a, st);
  if (a.act == SBK_ACT_SWISH) return launch_lp256<false, 0, SBK_ACT_SWISH>(a, st);
  switch (g_lp256_mode) {
    case 1: return launch_lp256<false, 1, SBK_ACT_NONE>(a, st);
    case 2: return launch_lp256<false, 2, SBK_ACT_NONE>(a, st);
    case 4: return launch_lp256<false, 4, SBK_ACT_NONE>(a, st);
    case 8: return launch_lp256<false, 8, SBK_ACT_NONE>(a, st);
    case 16: return launch_lp256<false, 16, SBK_ACT_NONE>(a, st);
    case 32: return launch_lp256<false, 32, SBK_ACT_NONE>(a, st);
    case 48: return launch_lp256<false, 48, SBK_ACT_NONE>(a, st);
    default: return launch_lp256<false, 0, SBK_ACT_NONE>(a, st);
  }
}
}  // namespace sbk
