// HBM calibration kernels (sbk_prof_stream_f32): what a plain float4 streaming read / copy reaches on THIS chip.
// "measure achievable with a rocprof copy kernel first" -- the HBM-bound kernels of the path quote their
// rate against these numbers (MI355X_MICROARCH.md measured 6.29 TB/s for a float4 copy; the spec peak is 8 TB/s).
// Grid-stride, 16 B per lane, 2048 workgroups (8 per CU), four independent loads in flight per lane.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n4) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n4) dst[i + u * 256] = v[u];
  }
}

__global__ void __launch_bounds__(256) stream_read_kernel(const float4* __restrict__ src, float* __restrict__ sink, long n4) {
  const long stride = (long)gridDim.x * 256 * 4;
  float acc = 0.0f;
  for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < n4 ? src[i + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  if (acc == 12345.678f) sink[blockIdx.x] = acc;  // keeps the loads alive; practically never true
}

// f32 MFMA ceiling under the chip's power management: every SIMD issues v_mfma_f32_32x32x2_f32 back to back from
// registers (4 independent accumulators per wave, no memory traffic).  `seed_scale` 0 = zero operands, else uniform
// pseudo-random operands in [-1,1) (the clock the chip sustains depends on the data: MI355X_MICROARCH.md, DVFS).
__global__ void __launch_bounds__(256) mfma_peak_kernel(float* __restrict__ sink, int iters, float seed_scale) {
  const unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  float a[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = seed_scale * ((float)((h >> (e * 3)) & 0xffff) / 32768.0f - 1.0f);
    b[e] = seed_scale * ((float)((h >> (e * 2 + 5)) & 0xffff) / 32768.0f - 1.0f);
  }
  sbk::f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = sbk::mfma_32x32x2(a[e], b[(e + i) & 3], acc[i]);
  }
  float t = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  if (t == 12345.678f) sink[blockIdx.x] = t;
}

}  // namespace

// `iters` x 16 MFMAs per wave, `wgs` workgroups of 4 waves; *tflops = 2*32*32*2 flops per MFMA / time (HOST out)
extern "C" int sbk_prof_mfma_peak_f32(float* sink, int wgs, int iters, int random_data, float* tflops, sbk_stream_t stream) {
  SBK_REQUIRE(sink && tflops && wgs > 0 && iters > 0, "mfma_peak: bad arguments");
  hipStream_t st = sbk::as_stream(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return sbk::fail(1, "event create");
  SBK_LAUNCH(mfma_peak_kernel, dim3(wgs), dim3(256), 0, st, sink, iters, random_data ? 1.0f : 0.0f);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < 5; ++i) SBK_LAUNCH(mfma_peak_kernel, dim3(wgs), dim3(256), 0, st, sink, iters, random_data ? 1.0f : 0.0f);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *tflops = (float)(5.0 * wgs * 4.0 * iters * 16.0 * 4096.0 / (ms * 1e-3) / 1e12);
  return sbk::launch_status("sbk_prof_mfma_peak_f32");
}

// mode 0: copy src -> dst (n floats each, 2 * 4 * n bytes move); mode 1: read src (4 * n bytes; dst = a sink of >= 2048 floats).
// `iters` back-to-back launches between two events on `stream`; *us_per_launch = mean time of one.
extern "C" int sbk_prof_stream_f32(const float* src, float* dst, long n, int mode, int iters, float* us_per_launch,
                                   sbk_stream_t stream) {
  SBK_REQUIRE(src && dst && us_per_launch && n > 0 && n % 4 == 0 && iters > 0, "stream: bad arguments");
  SBK_REQUIRE(sbk::aligned16(src) && sbk::aligned16(dst), "stream: operands must be 16-byte aligned");
  SBK_REQUIRE(mode == 0 || mode == 1, "stream: mode %d", mode);
  hipStream_t st = sbk::as_stream(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return sbk::fail(1, "event create");
  const long n4 = n / 4;
  const long want = (n4 + 1023) / 1024;
  const dim3 grid((unsigned)(want < 2048 ? want : 2048)), block(256);
  auto go = [&]() {
    if (mode == 0) {
      SBK_LAUNCH(stream_copy_kernel, grid, block, 0, st, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4);
    } else {
      SBK_LAUNCH(stream_read_kernel, grid, block, 0, st, reinterpret_cast<const float4*>(src), dst, n4);
    }
  };
  for (int i = 0; i < 3; ++i) go();
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) go();
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = ms * 1000.0f / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return sbk::launch_status("sbk_prof_stream_f32");
}
