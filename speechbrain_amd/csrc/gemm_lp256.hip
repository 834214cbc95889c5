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
//   * persistent over whole tiles, the next tile's first two K tiles issued before this tile's epilogue.
// Arithmetic, epilogue and outputs are gemm_nt_bf16dma_kernel's / gemm_nt_fp8dma_kernel's (fp32 accumulation; per-row scales of
// both fp8 operands applied to the accumulators; bias, activation, alpha, fp32 residual; fp32 / bf16 / e4m3 outputs): the same sums
// in the same order per output element (K ascending, one accumulator), so the two kernels agree bit for bit.
#include "common.h"
#include "internal.h"

#include <type_traits>

using sbk::f32x16;

namespace sbk {
int device_cus();
int g_lp256 = 1;  // key 61: 1 (default) = shapes with >= 128 tiles of 256 x 256 take this kernel, 0 = never, 2 = always (tests)
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
constexpr size_t kLdsBytes = (size_t)2 * kSlotF * sizeof(float);

template <bool FP8>
__global__ void __launch_bounds__(512, 2) gemm_nt_lp256_kernel(sbk::Lp256Args s) {
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
  const int ldr = s.ldr, ldc = s.ldc, ldcb = s.ldcb, ldc8 = s.ldc8, M = s.M, N = s.N, act = s.act;
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
    if constexpr (FP8) {
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

  // acc[i][jj][r]: row m0 + wrow0 + 32 i + (r & 3) + 8 (r >> 2) + 4 half, column n0 + wcol0 + 32 jj + lrow
  auto epilogue = [&](int t) SBK_INLINE_LAMBDA {
    int m0, n0;
    tile_origin(t, m0, n0);
    const bool interior = m0 + 256 <= M && n0 + 256 <= N;  // uniform: no per-element predicates
    static_for<0, 4>([&](auto ic) SBK_INLINE_LAMBDA {  // (a compile-time i: left to `#pragma unroll` the compiler kept this loop rolled
      constexpr int i = decltype(ic)::value;           //  in the fp8 instantiation and moved the accumulators to scratch memory)
      const int rbase = m0 + wrow0 + i * 32 + 4 * half;
      float rs[16];
      if constexpr (FP8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = gsa ? gsa[min(rbase + (r & 3) + 8 * (r >> 2), M - 1)] : 1.0f;
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int col = n0 + wcol0 + jj * 32 + lrow;
        const bool col_ok = interior || col < N;
        const float bv = (gbias && col_ok) ? gbias[col] : 0.0f;
        float v[16];
        if constexpr (FP8) {
          const float cs = (gsw && col_ok) ? gsw[col] : 1.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[i][jj][r] * (rs[r] * cs) + bv;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[i][jj][r] + bv;
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
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (interior || (col_ok && row < M)) {
            float o = v[r] * alpha;
            if (gR) o += gR[(size_t)row * ldr + col];
            if (gC) gC[(size_t)row * ldc + col] = o;
            if (gCb) gCb[(size_t)row * ldcb + col] = sbk::f32_to_bf16(o);
            if constexpr (FP8) {
              if (gC8) gC8[(size_t)row * ldc8 + col] = (unsigned char)(sbk::f32x2_to_fp8(o * c8_inv, 0.0f) & 0xff);
            }
          }
        }
      }
    });
  };

  // the barrier between two phases, pinned by scheduling fences (gemm_nt_x3p_kernel: MFMAs touch no memory, the scheduler would
  // hoist the barrier that ENDS a matrix phase above its MFMAs and the two groups would take turns instead of overlapping)
  auto phase_barrier = [&]() SBK_INLINE_LAMBDA {
    sbk::sched_fence();
    sbk::block_barrier_raw();
    sbk::sched_fence();
  };

  // ---- main loop.  Per tile: K tiles 0 and 1 are in flight on entry.  Step (m, h): [issue K tile m + 1 at h = 0, m >= 1 -- its slot's
  // previous tenant m - 1 was last fetched in step (m - 1, 1), which ended two physical barriers ago for this group and one for the
  // other] fetch; drain the LDS reads; [h = 1: this wave's share of K tile m + 1 has landed -- read two program barriers later, i.e. at
  // least one physical barrier after the OTHER group's wait]; barrier; MFMAs; barrier.
  setup(t_first);
  issue(0, 0);
  if (KT > 1) issue(1, 1);
  zero();
  for (int ord = 0; ord < ntile; ++ord) {
    const int t = t_first + ord * t_stride;
    sbk::vm_drain();           // K tiles 0 and 1 of this tile (and the previous tile's stores)
    sbk::block_barrier_raw();  // ... everybody's
    if (group == 1) sbk::block_barrier_raw();
#pragma unroll 1
    for (int m = 0; m < KT; ++m) {
      const int slot = m & 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 0 && m >= 1 && m + 1 < KT) issue(m + 1, slot ^ 1);
        fetch(slot, h);
        sbk::lds_drain();
        if (h == 1) sbk::vm_drain();
        phase_barrier();
        multiply();
        phase_barrier();
      }
    }
    if (group == 0) sbk::block_barrier_raw();
    // both groups aligned, nobody reads LDS: the next tile's first two K tiles fly during this one's epilogue
    if (ord + 1 < ntile) {
      setup(t + t_stride);
      issue(0, 0);
      if (KT > 1) issue(1, 1);
    }
    epilogue(t);
    zero();
  }
}

template <bool FP8>
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
    (void)SBK_ALLOW_DYN_LDS(gemm_nt_lp256_kernel<FP8>, kLdsBytes);
    once = true;
  }
  SBK_LAUNCH(gemm_nt_lp256_kernel<FP8>, dim3((unsigned)G), dim3(512), kLdsBytes, st, a);
  return sbk::launch_status(FP8 ? "sbk_gemm_nt_fp8a" : "sbk_gemm_nt_bf16a");
}

}  // namespace

namespace sbk {
bool lp256_routed(int M, int N, long k_bytes) {
  if (g_lp256 == 0 || k_bytes < 256 || k_bytes % 128 != 0) return false;
  if (g_lp256 == 2) return true;
  return (long)cdiv(M, 256) * cdiv(N, 256) >= 128;
}
int gemm_nt_lp256(const Lp256Args& a, bool fp8, hipStream_t st) {
  return fp8 ? launch_lp256<true>(a, st) : launch_lp256<false>(a, st);
}
}  // namespace sbk
