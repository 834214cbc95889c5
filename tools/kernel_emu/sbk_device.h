// Host-side stand-in for csrc/hip/sbk_device.h  --  TEST TOOLING ONLY.
//
// Gives the device vocabulary (threadIdx, __shared__, __syncthreads, wave
// shuffles, f32 MFMA, SBK_LAUNCH ...) host semantics so the very same kernel
// sources can be compiled with g++ and single-stepped / unit-tested on a CPU
// box without a GPU.  Each workgroup runs as a set of cooperative fibers on
// one OS thread (so `static thread_local` storage behaves like LDS); waves are
// groups of 64 consecutive fibers that rendezvous for shuffles and MFMA.
//
// This is NOT a product path: the shipped library (libsbk_hip.so) is built from
// csrc/hip/sbk_device.h by hipcc only, and speechbrain_amd.native refuses to
// run without it.  The emulator library is loaded only by tests that ask for it
// explicitly (tests/emu_utils.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyDeviceToDevice = 3, hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return 0; }
enum { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 16; return 0; }
struct hipDeviceProp_t { char gcnArchName[256]; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "gfx950:emu"); return 0; }
typedef void* hipEvent_t;
// hipGraph: not emulated -- capture reports failure, the callers fall back to plain launches
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = nullptr; return 1; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)2; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 2; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local

namespace sbk_emu {
struct ThreadCtx {
  dim3 tid, bid, bdim, gdim;
  int lin, lane, wave;
};
ThreadCtx& cur();
void block_barrier();
void wave_barrier();
float* wave_buf(int which);  // 64-float exchange buffers of the current wave (which = 0,1)
void* dyn_lds();
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
// every workgroup of the grid on an OS thread of its own, all running at once (a cooperative launch: grid barriers work)
void launch_coop(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
void grid_barrier_wait(int* ctr, int target);
}  // namespace sbk_emu

#define threadIdx (sbk_emu::cur().tid)
#define blockIdx (sbk_emu::cur().bid)
#define blockDim (sbk_emu::cur().bdim)
#define gridDim (sbk_emu::cur().gdim)
static inline void __syncthreads() { sbk_emu::block_barrier(); }

using std::max;
using std::min;
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int atomicMax(int* p, int v) {
  int o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return o;
}
static inline int atomicMin(int* p, int v) {
  int o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return o;
}
static inline int atomicMaxUnused_(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }

namespace sbk {
constexpr int kWave = 64;
struct f32x16 {
  float v[16];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
struct f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};

static inline f32x16 mfma_32x32x2(float a, float b, f32x16 acc) {
  const int l = sbk_emu::cur().lane;
  float* A = sbk_emu::wave_buf(0);
  float* B = sbk_emu::wave_buf(1);
  A[l] = a;
  B[l] = b;
  sbk_emu::wave_barrier();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float c = acc[r];
    for (int k = 0; k < 2; ++k) c = fmaf(A[row + 32 * k], B[col + 32 * k], c);
    acc[r] = c;
  }
  sbk_emu::wave_barrier();
  return acc;
}
struct __attribute__((aligned(16))) bf16x8 {
  unsigned short v[8];
};
static inline unsigned short f32_to_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline bf16x8 pack_bf16x8(const unsigned short (&h)[8]) {
  bf16x8 r;
  memcpy(r.v, h, sizeof(r.v));
  return r;
}
static inline bf16x8 cvt_bf16x8(const float (&x)[8]) {
  bf16x8 r;
  for (int e = 0; e < 8; ++e) r.v[e] = f32_to_bf16(x[e]);
  return r;
}
static inline bf16x8 bf16x8_from_words(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  const unsigned w[4] = {w0, w1, w2, w3};
  bf16x8 r;
  memcpy(r.v, w, sizeof(r.v));
  return r;
}
static inline unsigned bf16_pair(float x0, float x1) { return (unsigned)f32_to_bf16(x0) | ((unsigned)f32_to_bf16(x1) << 16); }
static inline float bf16_to_f32_(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 acc) {
  const int l = sbk_emu::cur().lane;
  float* A = sbk_emu::wave_buf(0);  // [lane][8]
  float* B = sbk_emu::wave_buf(1);
  for (int e = 0; e < 8; ++e) {
    A[l * 8 + e] = bf16_to_f32_(a.v[e]);
    B[l * 8 + e] = bf16_to_f32_(b.v[e]);
  }
  sbk_emu::wave_barrier();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float c = acc[r];
    for (int kb = 0; kb < 2; ++kb)
      for (int e = 0; e < 8; ++e) c = fmaf(A[(row + 32 * kb) * 8 + e], B[(col + 32 * kb) * 8 + e], c);
    acc[r] = c;
  }
  sbk_emu::wave_barrier();
  return acc;
}
// fp16 / fp8 (e4m3, OCP) operand forms: host conversions + the same contraction as the bf16 form
static inline unsigned short f32_to_f16(float f) {  // round to nearest even
  unsigned u;
  memcpy(&u, &f, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);
  if (u >= 0x47800000u) return (unsigned short)(sign | 0x7c00u);  // overflow -> inf
  if (u < 0x38800000u) {  // subnormal half (or zero)
    if (u < 0x33000000u) return (unsigned short)sign;
    const int e = (int)(u >> 23);
    unsigned m = (u & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e;  // 14 .. 24
    const unsigned half_m = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
    unsigned r = half_m + ((rem > mid || (rem == mid && (half_m & 1))) ? 1 : 0);
    return (unsigned short)(sign | r);
  }
  unsigned r = u - 0x38000000u;
  const unsigned rem = r & 0x1fffu;
  r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
  return (unsigned short)(sign | r);
}
static inline float f16_to_f32_(unsigned short h) {
  const unsigned sign = ((unsigned)h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = ldexpf((float)m, -24);
  } else if (e == 31) {
    f = m ? NAN : INFINITY;
  } else {
    f = ldexpf((float)(m | 0x400u), (int)e - 25);
  }
  unsigned u;
  memcpy(&u, &f, 4);
  u |= sign;
  memcpy(&f, &u, 4);
  return f;
}
static inline float fp8_to_f32_(unsigned char b) {  // e4m3fn: bias 7, no inf, 0x7f = NaN
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ((e == 15 && m == 7) ? NAN : ldexpf((float)(8 + m), e - 10));
  return s ? -f : f;
}
static inline unsigned char f32_to_fp8_(float x) {  // nearest even, saturating at 448
  if (x != x) return 0x7f;
  const unsigned char s = std::signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (a > 448.0f) a = 448.0f;
  unsigned char best = 0;
  float bd = INFINITY;
  for (int c = 0; c < 0x7f; ++c) {  // 127 finite magnitudes: exhaustive nearest (ties to the even code)
    const float d = fabsf(fp8_to_f32_((unsigned char)c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) {
      bd = d;
      best = (unsigned char)c;
    }
  }
  return s | best;
}
static inline unsigned short f32x2_to_fp8(float a, float b) {
  return (unsigned short)(f32_to_fp8_(a) | ((unsigned)f32_to_fp8_(b) << 8));
}
struct __attribute__((aligned(16))) f16x8 {
  unsigned short v[8];
};
typedef long fp8x8;
template <typename F>
static inline f32x16 mfma_32x32x16_generic_(const float (&av)[8], const float (&bv)[8], f32x16 acc, F) {
  const int l = sbk_emu::cur().lane;
  float* A = sbk_emu::wave_buf(0);
  float* B = sbk_emu::wave_buf(1);
  for (int e = 0; e < 8; ++e) {
    A[l * 8 + e] = av[e];
    B[l * 8 + e] = bv[e];
  }
  sbk_emu::wave_barrier();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float c = acc[r];
    for (int kb = 0; kb < 2; ++kb)
      for (int e = 0; e < 8; ++e) c = fmaf(A[(row + 32 * kb) * 8 + e], B[(col + 32 * kb) * 8 + e], c);
    acc[r] = c;
  }
  sbk_emu::wave_barrier();
  return acc;
}
static inline f32x16 mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 acc) {
  float av[8], bv[8];
  for (int e = 0; e < 8; ++e) {
    av[e] = f16_to_f32_(a.v[e]);
    bv[e] = f16_to_f32_(b.v[e]);
  }
  return mfma_32x32x16_generic_(av, bv, acc, 0);
}
static inline f32x16 mfma_32x32x16_fp8(fp8x8 a, fp8x8 b, f32x16 acc) {
  float av[8], bv[8];
  for (int e = 0; e < 8; ++e) {
    av[e] = fp8_to_f32_((unsigned char)((unsigned long)a >> (8 * e)));
    bv[e] = fp8_to_f32_((unsigned char)((unsigned long)b >> (8 * e)));
  }
  return mfma_32x32x16_generic_(av, bv, acc, 0);
}
struct i32x8 {
  int v[8];
  int& operator[](int i) { return v[i]; }
  const int& operator[](int i) const { return v[i]; }
};
static inline i32x8 i32x8_from_u4(uint4 lo, uint4 hi) {
  return i32x8{{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w}};
}
// e4m3 operands, 32 per lane: four passes of the 8-per-lane contraction over the lane's bytes 8q .. 8q+7
static inline f32x16 mfma_32x32x64_fp8(i32x8 a, i32x8 b, f32x16 acc) {
  for (int q = 0; q < 4; ++q) {
    float av[8], bv[8];
    for (int e = 0; e < 8; ++e) {
      av[e] = fp8_to_f32_((unsigned char)((unsigned)a.v[2 * q + e / 4] >> (8 * (e % 4))));
      bv[e] = fp8_to_f32_((unsigned char)((unsigned)b.v[2 * q + e / 4] >> (8 * (e % 4))));
    }
    acc = mfma_32x32x16_generic_(av, bv, acc, 0);
  }
  return acc;
}
static inline f32x4 mfma_16x16x4(float a, float b, f32x4 acc) {
  const int l = sbk_emu::cur().lane;
  float* A = sbk_emu::wave_buf(0);
  float* B = sbk_emu::wave_buf(1);
  A[l] = a;
  B[l] = b;
  sbk_emu::wave_barrier();
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r, col = l & 15;
    float c = acc[r];
    for (int k = 0; k < 4; ++k) c = fmaf(A[row + 16 * k], B[col + 16 * k], c);
    acc[r] = c;
  }
  sbk_emu::wave_barrier();
  return acc;
}
template <typename T>
static inline T shfl_idx_(T v, int src) {
  const int l = sbk_emu::cur().lane;
  float* X = sbk_emu::wave_buf(0);
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  memcpy(&X[l], &v, 4);
  sbk_emu::wave_barrier();
  T r;
  memcpy(&r, &X[src & 63], 4);
  sbk_emu::wave_barrier();
  return r;
}
static inline float shfl_xor(float v, int m) { return shfl_idx_(v, sbk_emu::cur().lane ^ m); }
static inline int shfl_xor(int v, int m) { return shfl_idx_(v, sbk_emu::cur().lane ^ m); }
static inline float shfl(float v, int lane) { return shfl_idx_(v, lane); }
static inline int shfl(int v, int lane) { return shfl_idx_(v, lane); }
static inline void wave_sync() { sbk_emu::wave_barrier(); }
static inline void sched_fence() {}
static inline float fast_ldexp(float x, int e) {
  if (e < -400) return x * 0.0f;
  if (e > 400) e = 400;
  return ldexpf(x, e);
}
static inline int frexp_exp(float x) { return (x != 0.0f && std::isfinite(x)) ? ilogbf(x) + 1 : 0; }
template <int G>
static inline float group_sum(float v) {  // same lane selects as the DPP sequence of the device header
  const int lane = sbk_emu::cur().lane;
  if (G >= 2) v += shfl_idx_(v, lane ^ 1);
  if (G >= 4) v += shfl_idx_(v, lane ^ 2);
  if (G >= 8) v += shfl_idx_(v, (lane & ~7) | (7 - (lane & 7)));
  if (G >= 16) v += shfl_idx_(v, (lane & ~15) | (15 - (lane & 15)));
  return v;
}

// LDS-DMA / inter-workgroup hand-off vocabulary (host semantics: workgroups run on OS threads of one process)
static inline void glds16(const float* src, float* lds_wave_base) {
  memcpy(reinterpret_cast<char*>(lds_wave_base) + 16 * sbk_emu::cur().lane, src, 16);
}
static inline void glds16_uniform(const float* base, unsigned lane_byte_offset, float* lds_wave_base) {
  memcpy(reinterpret_cast<char*>(lds_wave_base) + 16 * sbk_emu::cur().lane, reinterpret_cast<const char*>(base) + lane_byte_offset, 16);
}
static inline int uniform(int v) { return v; }
static inline float gelu_erfc(float x) {  // (the device version's formula; its v_rcp_f32 / v_exp_f32 are a true quotient / exp2f here)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = 1.0f / fmaf(0.3275911f, z, 1.0f);
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float q = (p * t) * exp2f(-1.44269504088896340736f * (z * z));
  return (0.5f * x) * (x >= 0.0f ? 2.0f - q : q);
}
template <class T>
static inline void keep(const T&) {}
template <class T>
static inline T opaque_zero() { T t; memset(&t, 0, sizeof(T)); return t; }
static inline float exp2_raw(float x) { return exp2f(x); }
static inline bool wave_any(bool p) {  // (every fiber of the wave calls it)
  int v = p ? 1 : 0;
  for (int k = 32; k >= 1; k >>= 1) v |= shfl_idx_(v, sbk_emu::cur().lane ^ k);
  return v != 0;
}
template <bool NT>
static inline float4 ld16(const float* p) { return *reinterpret_cast<const float4*>(p); }  // (the cache policy has no host meaning)
template <bool NT>
static inline float ld4(const float* p) { return *p; }
template <int MASK, int SIZE>
static inline void sched_group() {}
template <int P>
static inline void set_prio() {}
static inline void pin(unsigned&) {}
static inline void pin(float&) {}
static inline void pin(f32x16&) {}
static inline void vm_drain() {}
static inline void vm_drain_visible() {}
static inline void lds_drain() {}
static inline void block_barrier_raw() { sbk_emu::block_barrier(); }
template <int N>
static inline void vm_wait() {}
static inline void release_agent() { __atomic_thread_fence(__ATOMIC_RELEASE); }
static inline void acquire_agent() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
static inline int atomic_add_agent(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline void atomic_store_agent(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline int atomic_load_agent(const int* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline float ld_agent(const float* p) { unsigned u = __atomic_load_n(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED); float f; memcpy(&f, &u, 4); return f; }
static inline float2 ld_agent2(const float* p) { return float2{ld_agent(p), ld_agent(p + 1)}; }
static inline void st_agent(float* p, float v) { unsigned u; memcpy(&u, &v, 4); __atomic_store_n(reinterpret_cast<unsigned*>(p), u, __ATOMIC_RELAXED); }
static inline long long wall_clock() { return 0; }
static inline void grid_arrive(int* ctr) {  // (every fiber calls it; fiber 0 of the workgroup takes the ticket)
  sbk_emu::block_barrier();
  if (sbk_emu::cur().lin == 0) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    __atomic_fetch_add(ctr, 1, __ATOMIC_SEQ_CST);
  }
}
static inline void grid_arrive_tree(int* sub, int sub_target, int* top) {
  sbk_emu::block_barrier();
  if (sbk_emu::cur().lin == 0) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    const int old = __atomic_fetch_add(sub, 1, __ATOMIC_SEQ_CST);
    if (old + 1 == sub_target) __atomic_fetch_add(top, 1, __ATOMIC_SEQ_CST);
  }
}
static inline void grid_wait(int* ctr, int target) {
  if (sbk_emu::cur().lin == 0) sbk_emu::grid_barrier_wait(ctr, target);
  sbk_emu::block_barrier();
}
static inline float mul_rn(float a, float b) { return a * b; }  // (g++ on x86-64 does not contract without -mfma)
static inline float add_rn(float a, float b) { return a + b; }
static inline float sub_rn(float a, float b) { return a - b; }
static inline void wave_argmax(float& v, int& i) {  // largest v; among the lanes that hold it, the smallest i
  float m = v;
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, shfl_xor(m, k));
  int c = v == m ? i : 0x7fffffff;
  for (int k = 32; k >= 1; k >>= 1) c = std::min(c, shfl_xor(c, k));
  v = m;
  i = c;
}
static inline float wave_sum(float v) {
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}
static inline float wave_max(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
  return v;
}
}  // namespace sbk

// lambdas of a kernel body must be inlined (a called lambda takes its captures through scratch memory)
#define SBK_INLINE_LAMBDA
#define SBK_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(sbk_emu::dyn_lds())
#define SBK_ALLOW_DYN_LDS(kernel, bytes) ((void)(bytes), 0)
#define SBK_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
  sbk_emu::launch([=]() { kernel(__VA_ARGS__); }, grid, block, lds_bytes)
#define SBK_LAUNCH_COOP(kernel, grid, block, lds_bytes, stream, arg_struct) \
  (sbk_emu::launch_coop([=]() { kernel(arg_struct); }, grid, block, lds_bytes), 0)
#define SBK_COOP_MAX_GRID(kernel, block_threads, lds_bytes, out_int) ((out_int) = 6, 0)
