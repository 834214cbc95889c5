// ConvolutionFrontEnd block:  reflect-pad(1) -> Conv2d 3x3 stride 2 (+bias)
//   -> LayerNorm over (F',C) -> LeakyReLU(0.01)          (sbk_conv_block_f32)
//
// Roofline: HBM for block 0 (C_in = 1: 320 B in, 10 KB out per output frame);
// block 1 (64 -> 32 channels) is a small implicit GEMM kept on the vector ALU
// in this revision (5.9 GFLOP per 32 x 10 s batch, < 4 % of the encoder).
//
// Layout: activations are [B,T,F,C] (C fastest) on both sides, exactly the
// tensors the reference hands between blocks, so no transposes exist.  One
// workgroup produces one output frame (all F' x C' values): the 3 input frames
// it needs are staged in LDS with the reflect padding resolved at staging
// time, the 3x3 taps are read from a [C_in*9, C_out] re-laid-out weight
// (coalesced across output channels), and the (F',C') LayerNorm statistics are
// a workgroup reduction over values still in registers -- the pre-norm
// activation never goes to HBM.
#include "common.h"

namespace {

constexpr int kNPT = 10;  // outputs kept per thread (F'*C' <= 2560 with 256 threads)

struct ConvArgs {
  const float* x;      // [B,Tin,Fin,Cin]
  const float* wt;     // [Cin*9, Cout], row = (ci*3+kf)*3+kt
  const float* bias;   // [Cout]
  const float* gamma;  // [Fout*Cout]
  const float* beta;
  float* y;            // [B,Tout,Fout,Cout]
  int B, Tin, Fin, Cin, Tout, Fout, Cout;
  float eps, slope;
};

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = sbk::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) conv_block_kernel(ConvArgs a) {
  SBK_DYN_LDS(float, patch);  // [3][Fin+2][Cin]
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, to = blockIdx.x;
  const int Fp = a.Fin + 2;
  const int rowlen = Fp * a.Cin;
  for (int i = tid; i < 3 * rowlen; i += 256) {
    const int kt = i / rowlen, rem = i % rowlen;
    const int fp = rem / a.Cin, ci = rem % a.Cin;
    const int ti = reflect1(2 * to + kt - 1, a.Tin);
    const int fi = reflect1(fp - 1, a.Fin);
    patch[i] = a.x[(((size_t)b * a.Tin + ti) * a.Fin + fi) * a.Cin + ci];
  }
  __syncthreads();

  const int nout = a.Fout * a.Cout;
  float acc[kNPT];
#pragma unroll
  for (int s = 0; s < kNPT; ++s) {
    const int o = tid + s * 256;
    float v = 0.0f;
    if (o < nout) {
      const int c = o % a.Cout, fo = o / a.Cout;
      v = a.bias[c];
      for (int ci = 0; ci < a.Cin; ++ci) {
        for (int kf = 0; kf < 3; ++kf) {
          const float* prow = patch + (size_t)(2 * fo + kf) * a.Cin + ci;
          const float* wrow = a.wt + (size_t)((ci * 3 + kf) * 3) * a.Cout + c;
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) v = fmaf(prow[kt * rowlen], wrow[kt * a.Cout], v);
        }
      }
    }
    acc[s] = v;
  }

  float s1 = 0.0f;
#pragma unroll
  for (int s = 0; s < kNPT; ++s)
    if (tid + s * 256 < nout) s1 += acc[s];
  const float mean = block_sum(s1, red) / (float)nout;
  float s2 = 0.0f;
#pragma unroll
  for (int s = 0; s < kNPT; ++s)
    if (tid + s * 256 < nout) s2 += (acc[s] - mean) * (acc[s] - mean);
  const float rstd = rsqrtf(block_sum(s2, red) / (float)nout + a.eps);

  float* yo = a.y + ((size_t)b * a.Tout + to) * nout;
#pragma unroll
  for (int s = 0; s < kNPT; ++s) {
    const int o = tid + s * 256;
    if (o < nout) {
      const float v = (acc[s] - mean) * rstd * a.gamma[o] + a.beta[o];
      yo[o] = v > 0.0f ? v : a.slope * v;
    }
  }
}

}  // namespace

extern "C" int sbk_conv_block_f32(const float* x, const float* wt, const float* bias, const float* gamma,
                                  const float* beta, float* y, int B, int Tin, int Fin, int Cin, int Cout, float eps,
                                  float slope, sbk_stream_t stream) {
  if (B == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(x && wt && bias && gamma && beta && y, "conv_block: null operand");
  SBK_REQUIRE(B >= 0 && Tin >= 2 && Fin >= 2 && Cin >= 1 && Cout >= 1, "conv_block: bad shape");
  const int Tout = (Tin - 1) / 2 + 1, Fout = (Fin - 1) / 2 + 1;  // floor((n + 2 - 3) / 2) + 1
  SBK_REQUIRE(Fout * Cout <= kNPT * 256, "conv_block: F'*C' = %d exceeds %d", Fout * Cout, kNPT * 256);
  const size_t lds = (size_t)3 * (Fin + 2) * Cin * sizeof(float);
  SBK_REQUIRE(lds <= 64 * 1024, "conv_block: input patch of %zu B does not fit the LDS window", lds);
  if (B == 0) return 0;
  ConvArgs a{x, wt, bias, gamma, beta, y, B, Tin, Fin, Cin, Tout, Fout, Cout, eps, slope};
  sbk::ProfScope prof(Cin == 1 ? "conv_block_cin1" : "conv_block", 2.0 * 9 * Cin * (double)B * Tout * Fout * Cout,
                      4.0 * B * ((double)Tin * Fin * Cin + (double)Tout * Fout * Cout), sbk::as_stream(stream));
  SBK_LAUNCH(conv_block_kernel, dim3(Tout, B), dim3(256), lds, sbk::as_stream(stream), a);
  return sbk::launch_status("sbk_conv_block_f32");
}
