// GLU + depthwise Conv1d of the Conformer ConvolutionModule      (sbk_glu_dwconv_f32)
//
// Roofline: HBM (12*d bytes per frame: read [T,2d], write [T,d]; 31 taps = 62
// flop per output element).  One workgroup = (batch, 32-frame tile, 64-channel
// tile): the gated activations of the tile plus a (ksize-1)-frame halo are
// computed once into LDS ([frame][channel], channel fastest => conflict-free
// across a wave), each thread keeps the ksize taps of its channel in registers
// and slides over 8 output frames.  Frames outside [0,T) contribute zeros
// (Conv1d zero padding); frames past an utterance's true length are NOT
// masked here, exactly like the reference (Conformer.py:315-328 masks only
// the module output).
#include "common.h"

namespace {

constexpr int kTT = 32;    // output frames per workgroup
constexpr int kCT = 64;    // channels per workgroup
constexpr int kMaxK = 63;  // largest odd kernel size kept in registers

template <int KS>
__global__ void __launch_bounds__(256) glu_dwconv_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int T, int d, int chunk) {
  constexpr int HALO = (KS - 1) / 2;
  constexpr int ROWS = kTT + KS - 1;
  __shared__ float g[ROWS][kCT];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * kTT, c0 = blockIdx.y * kCT, b = blockIdx.z;
  const int c = tid & 63, tq = tid >> 6;
  const int ch = c0 + c;
  const bool ch_ok = ch < d;
  const float* hb = h + (size_t)b * T * 2 * d;
  for (int r = tq; r < ROWS; r += 4) {
    const int t = t0 + r - HALO;
    float v = 0.0f;
    if (ch_ok && t >= 0 && t < T) {
      const float a = hb[(size_t)t * 2 * d + ch];
      const float gate = hb[(size_t)t * 2 * d + d + ch];
      v = a * (1.0f / (1.0f + expf(-gate)));
    }
    g[r][c] = v;
  }
  float wk[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) wk[k] = ch_ok ? w[(size_t)ch * KS + k] : 0.0f;
  const float bv = ch_ok ? bias[ch] : 0.0f;
  __syncthreads();
#pragma unroll
  for (int o = 0; o < kTT / 4; ++o) {
    const int tl = tq * (kTT / 4) + o;
    const int t = t0 + tl;
    // Dynamic Chunk Convolution (Conformer.py:190-313): an output frame sees its past normally but nothing beyond
    // the end of its own chunk -- taps k with t + k - HALO >= chunk_end read zeros
    const int kmax = chunk > 0 ? min(KS, (t / chunk + 1) * chunk - t + HALO) : KS;
    float acc = bv;
#pragma unroll
    for (int k = 0; k < KS; ++k) acc = fmaf(k < kmax ? wk[k] : 0.0f, g[tl + k][c], acc);
    if (ch_ok && t < T) y[((size_t)b * T + t) * d + ch] = acc;
  }
}

}  // namespace

namespace sbk {
int glu_dwconv(const float* h, const float* w, const float* bias, float* y, int B, int T, int d, int ksize,
               hipStream_t st, int chunk) {
  if (B == 0 || T == 0) return 0;
  dim3 grid(cdiv(T, kTT), cdiv(d, kCT), B), block(256);
  ProfScope prof("glu_dwconv", (2.0 * ksize + 4.0) * B * T * d, 12.0 * B * T * d, st);
  switch (ksize) {
    case 31: SBK_LAUNCH((glu_dwconv_kernel<31>), grid, block, 0, st, h, w, bias, y, T, d, chunk); break;
    case 15: SBK_LAUNCH((glu_dwconv_kernel<15>), grid, block, 0, st, h, w, bias, y, T, d, chunk); break;
    case 7: SBK_LAUNCH((glu_dwconv_kernel<7>), grid, block, 0, st, h, w, bias, y, T, d, chunk); break;
    case 5: SBK_LAUNCH((glu_dwconv_kernel<5>), grid, block, 0, st, h, w, bias, y, T, d, chunk); break;
    case 3: SBK_LAUNCH((glu_dwconv_kernel<3>), grid, block, 0, st, h, w, bias, y, T, d, chunk); break;
    default: return fail(SBK_EINVAL, "glu_dwconv: kernel size %d not instantiated (3,5,7,15,31)", ksize);
  }
  return launch_status("sbk_glu_dwconv_f32");
}
}  // namespace sbk

extern "C" int sbk_glu_dwconv_f32(const float* h, const float* w, const float* bias, float* y, int B, int T, int d,
                                  int ksize, int chunk_size, sbk_stream_t stream) {
  if (B == 0 || T == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(h && w && bias && y, "glu_dwconv: null operand");
  SBK_REQUIRE(B >= 0 && T >= 0 && d > 0 && chunk_size >= 0, "glu_dwconv: bad shape");
  return sbk::glu_dwconv(h, w, bias, y, B, T, d, ksize, sbk::as_stream(stream), chunk_size);
}
