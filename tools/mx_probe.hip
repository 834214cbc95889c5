// Probe (not product code; built and run by tools/gpu_r4_m2.sh): what v_mfma_scale_f32_32x32x64_f8f6f4 computes and how fast.
//  1. operand layout: lane l supplies row (l & 31) and the 32 consecutive k of block (l >> 5) as 32 fp8 bytes (8 VGPRs);
//     C/D as every 32x32 MFMA.  Checked with A = e4m3 ramp values, B = asymmetric pattern against a host product.
//  2. scale operands: E8M0 byte per LANE (its row, its 32-k block); value 127 = x1, 128 = x2; which byte opsel selects.
//  3. rate: back-to-back MFMAs from registers on every CU, fp8 (scaled, K = 64) against bf16 (32x32x16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static float e4m3_to_float(uint8_t b) {  // OCP e4m3fn
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m / 8.0f, -6);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return s ? -v : v;
}

// out[mode][32][32]; A, B: [32 rows][64 k] bytes
__global__ void probe_kernel(const uint8_t* A, const uint8_t* B, float* out, int sa_val, int sb_val) {
  const int l = threadIdx.x, row = l & 31, kb = l >> 5;
  v8i a, b;
  const int* ap = reinterpret_cast<const int*>(A + row * 64 + kb * 32);
  const int* bp = reinterpret_cast<const int*>(B + row * 64 + kb * 32);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  for (int mode = 0; mode < 6; ++mode) {
    v16f c = {0};
    int sa = 0, sb = 0;
    // modes: 0: scales 0 / 0 (what does "no scale" mean?)  1: 127 / 127 (x1)  2: 128 / 127 (A x2)  3: byte 1 of the scale
    // register = 128 with opsel 1 on A  4: per-lane scale: lanes with kb = 1 get 128 on A  5: per-row: rows >= 16 get 128 on B
    if (mode == 1) { sa = 127; sb = 127; }
    if (mode == 2) { sa = 128; sb = 127; }
    if (mode == 3) { sa = 127 | (128 << 8); sb = 127; }
    if (mode == 4) { sa = kb ? 128 : 127; sb = 127; }
    if (mode == 5) { sa = 127; sb = row >= 16 ? 128 : 127; }
    if (mode == 3)
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa, 0, sb);
    else
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) out[(mode * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kb)) * 32 + row] = c[r];  // C[i][j]: i from (r, kb), j = lane & 31
  }
  (void)sa_val; (void)sb_val;
}

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* sink, int iters) {
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  if (KIND == 0) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x3a3a3a3a ^ (threadIdx.x << 3); }
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
  } else {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.01f * threadIdx.x); b[i] = (__bf16)(0.5f - 0.001f * threadIdx.x); }
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  float s = 0.0f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 123.456f) sink[0] = s;
}

int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  for (int i = 0; i < 32; ++i)
    for (int k = 0; k < 64; ++k) {
      hA[i * 64 + k] = (uint8_t)(0x30 + ((i * 3 + k) % 24));        // positive normals 0.5 .. ~7
      hB[i * 64 + k] = (uint8_t)((0x28 + ((i * 7 + 2 * k) % 20)) | ((k & 3) == 1 ? 0x80 : 0));  // asymmetric, some negative
    }
  static float ref[32][32], refk[2][32][32];
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s[2] = {0, 0};
      for (int k = 0; k < 64; ++k) s[k / 32] += (double)e4m3_to_float(hA[i * 64 + k]) * (double)e4m3_to_float(hB[j * 64 + k]);
      ref[i][j] = (float)(s[0] + s[1]);
      refk[0][i][j] = (float)s[0];
      refk[1][i][j] = (float)s[1];
    }
  uint8_t *dA, *dB;
  float* dO;
  hipMalloc((void**)&dA, sizeof(hA)); hipMalloc((void**)&dB, sizeof(hB)); hipMalloc((void**)&dO, 6 * 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO, 0, 0);
  static float out[6][32][32];
  hipMemcpy(out, dO, sizeof(out), hipMemcpyDeviceToHost);
  const char* names[6] = {"scales 0/0", "scales 127/127", "A scale 128", "A scale reg = 127|128<<8, opsel_a 1", "A scale 128 on lanes of k block 1", "B scale 128 on rows >= 16"};
  for (int m = 0; m < 6; ++m) {
    double e1 = 0, e2 = 0, eT = 0, ek = 0, er = 0, ratio = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        e1 = fmax(e1, fabs(out[m][i][j] - ref[i][j]));
        e2 = fmax(e2, fabs(out[m][i][j] - 2 * ref[i][j]));
        eT = fmax(eT, fabs(out[m][j][i] - ref[i][j]));
        ek = fmax(ek, fabs(out[m][i][j] - (refk[0][i][j] + 2 * refk[1][i][j])));
        er = fmax(er, fabs(out[m][i][j] - (j >= 16 ? 2.0 : 1.0) * ref[i][j]));
        ratio = fmax(ratio, fabs(ref[i][j]) > 1 ? fabs(out[m][i][j] / ref[i][j]) : 0);
      }
    printf("mx probe mode %d (%s): max|out - ref| %.4g, |out - 2 ref| %.4g, transposed %.4g, |out - (k0 + 2 k1)| %.4g, |out - ref * (col >= 16 ? 2 : 1)| %.4g, max ratio %.4g\n",
           m, names[m], e1, e2, eT, ek, er, ratio);
  }
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 20000;
  for (int kind = 0; kind < 2; ++kind) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(cus * 2), dim3(256), 0, 0, dO, iters);
      else hipLaunchKernelGGL(rate_kernel<1>, dim3(cus * 2), dim3(256), 0, 0, dO, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * (kind == 0 ? 64 : 16) * 4.0 * iters * (double)(cus * 2) * 4;
    printf("mx probe rate: %s: %.1f TF/s (%d CUs, %.2f ms)\n", kind == 0 ? "fp8 scaled 32x32x64" : "bf16 32x32x16", flops / (ms * 1e-3) / 1e12, cus, ms);
  }
  return 0;
}
