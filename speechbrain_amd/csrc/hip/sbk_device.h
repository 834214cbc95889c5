// Device-side vocabulary for the gfx950 kernels (wave64, MFMA, LDS).
//
// Every kernel in csrc/ is written against the handful of names declared here
// so that the source reads as plain CDNA4 HIP.  (tools/kernel_emu/ ships a
// header of the same name that gives these names host semantics, so kernels
// can be single-stepped on a CPU during development; it is test tooling and is
// never on the include path of the shipped library.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbk {

constexpr int kWave = 64;  // CDNA wavefront width

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// D(32x32) += A(32x2) * B(2x32).  Lane l supplies A[l&31][l>>5], B[l>>5][l&31];
// acc[r] is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].  Exact f32 fma chain.
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}
// D(16x16) += A(16x4) * B(4x16).  Lane l supplies A[l&15][l>>4], B[l>>4][l&15];
// acc[r] is D[(l>>4)*4 + r][l&15].
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

// bf16 operands: 8 per lane (4 VGPRs).  D(32x32) += A(32x16) * B(16x32), fp32 accumulate; lane l supplies row / column
// l&31 and the k block l>>5 (8 consecutive k each); acc layout as for mfma_32x32x2.  2.5 PFLOP/s dense on MI355X.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
// fp32 -> bf16, round to nearest even: gfx950 converts in hardware (v_cvt_pk_bf16_f32, two values per instruction)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ bf16x8 cvt_bf16x8(const float (&x)[8]) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (__bf16)x[e];
  return r;
}

// four 32-bit words = eight bf16 (element e in the low / high half of word e/2) -> one MFMA operand
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ bf16x8 bf16x8_from_words(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  u32x4 u = {w0, w1, w2, w3};
  return __builtin_bit_cast(bf16x8, u);
}
// two floats -> two bf16 (round to nearest even) in one word, x0 in the low half: one v_cvt_pk_bf16_f32
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned bf16_pair(float x0, float x1) {
  bf16x2 v = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(unsigned, v);
}

// fp16 operands (same shape and lane mapping as the bf16 form), fp32 accumulate
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }  // RNE
// OCP fp8 e4m3 operands: 8 per lane in one 64-bit register, fp32 accumulate (the non-scaled fp8 MFMA runs at the bf16 rate)
using fp8x8 = long;
__device__ __forceinline__ f32x16 mfma_32x32x16_fp8(fp8x8 a, fp8x8 b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, acc, 0, 0, 0);
}
// OCP e4m3 operands, 32 per lane (8 VGPRs): D(32x32) += A(32x64) * B(64x32), fp32 accumulate, on
// v_mfma_scale_f32_32x32x64_f8f6f4 with UNIT block scales (scale operands 0 = E8M0 127: measured exact,
// tools/mx_probe.hip).  Lane l supplies row / column l & 31 and 32 bytes of k block l >> 5 (the order of the k inside
// the instruction does not matter to a contraction as long as A and B use the same one).  Measured from registers:
// 4 267 TF/s, 2.06 x v_mfma_f32_32x32x16_bf16 (profiles/r04_m2_*).
using i32x8 = __attribute__((ext_vector_type(8))) int;
__device__ __forceinline__ f32x16 mfma_32x32x64_fp8(i32x8 a, i32x8 b, f32x16 acc) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0, 0, 0);
}
__device__ __forceinline__ i32x8 i32x8_from_u4(uint4 lo, uint4 hi) {
  i32x8 r = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
  return r;
}
// two floats -> two e4m3 bytes (low byte = a), round to nearest even, saturating at +-448 (v_cvt_pk_fp8_f32)
__device__ __forceinline__ unsigned short f32x2_to_fp8(float a, float b) {
  a = fminf(fmaxf(a, -448.0f), 448.0f);
  b = fminf(fmaxf(b, -448.0f), 448.0f);
  return (unsigned short)(__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffff);
}

// eight bf16 bit patterns -> one MFMA operand
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
__device__ __forceinline__ bf16x8 pack_bf16x8(const unsigned short (&h)[8]) {
  u16x8 u;
#pragma unroll
  for (int e = 0; e < 8; ++e) u[e] = h[e];
  return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ float shfl(float v, int lane) { return __shfl(v, lane, kWave); }
__device__ __forceinline__ int shfl(int v, int lane) { return __shfl(v, lane, kWave); }

// Orders this wave's LDS traffic in program order (the hardware already runs a
// wave's DS instructions in order; this pins the compiler's schedule).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// Scheduling fence for the compiler: nothing is moved across it (keeps a block of loads issued ahead of
// the arithmetic that consumes them instead of sunk next to each use).
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// One entry of an instruction-interleave recipe: the scheduler places `SIZE` instructions of class MASK (0x008 MFMA,
// 0x002 VALU, 0x100 LDS read, 0x020 VMEM read ...) next; a sequence of these pins the order in which the instructions
// of a basic block are issued (an in-order wave hides a VALU instruction under its own MFMAs only if it stands between
// them in program order).
template <int MASK, int SIZE>
__device__ __forceinline__ void sched_group() { __builtin_amdgcn_sched_group_barrier(MASK, SIZE, 0); }

// The value is materialised in a register HERE: keeps the optimiser from sinking the arithmetic that produces it into a
// later basic block (next to its use), i.e. out of the MFMA stream it was written to be issued under.
__device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(f32x16& x) { asm volatile("" : "+v"(x)); }  // (an MFMA result: keeps the MFMA at this point of the stream)

// 2^x as ONE v_exp_f32 (no range fix-ups: fine for softmax weights, x <= 0 or -inf; libm's exp2f / expf add a scaling and its undo);
// wave_any: the predicate holds in at least one lane (wave-uniform result)
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// measurement builds: a 16-byte value stays live up to this point / an all-zero 16-byte value the optimiser cannot see through
template <class T>
__device__ __forceinline__ void keep(const T& x) {
  static_assert(sizeof(T) == 16, "keep: one register quad");
  const u32x4 u = __builtin_bit_cast(u32x4, x);
  asm volatile("" ::"v"(u));
}
template <class T>
__device__ __forceinline__ T opaque_zero() {
  static_assert(sizeof(T) == 16, "opaque_zero: one register quad");
  u32x4 u = {0u, 0u, 0u, 0u};
  asm volatile("" : "+v"(u));
  return __builtin_bit_cast(T, u);
}
// GELU (erf form) for the epilogues of the REDUCED-PRECISION contractions (bf16 / e4m3 activations: csrc/gemm_lp.hip, gemm_lp256.hip;
// the fp32 parity path keeps libm's erff).  1 + erf(x / sqrt 2) = 2 - erfc(z) for x >= 0 and erfc(z) for x < 0, z = |x| / sqrt 2, with
// erfc by Abramowitz & Stegun 7.1.26 (absolute error <= 1.5e-7: below half an ulp of the bf16 / e4m3 value it is rounded to, and the
// negative tail keeps its RELATIVE accuracy because erfc is formed directly, not as 1 - erf) on one v_rcp_f32 and one v_exp_f32: about 17
// instructions where libm's erff is ~35 with both of its branches taken in a wave -- the first feed-forward projection of a Whisper
// large-v3 layer evaluates it 61 M times per launch (54 of its 146 us, profiles/r06_ad_*).
// Compiled under `fp contract(off)` with its fused operations written out: two kernels that must agree bit for bit (the 128 x 128 and
// the 256 x 256 contraction, tests/test_kernels.py) inline it, and what each kernel's optimiser would contract differs (seen on the GPU).
__device__ __forceinline__ float gelu_erfc(float x) {
#pragma clang fp contract(off)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float q = (p * t) * exp2_raw(-1.44269504088896340736f * (z * z));  // erfc(z)
  return (0.5f * x) * (x >= 0.0f ? 2.0f - q : q);
}
// wave priority for the instruction arbiter of the SIMD (0 = default .. 3)
template <int P>
__device__ __forceinline__ void set_prio() { __builtin_amdgcn_s_setprio(P); }

// x * 2^e and the exponent k of x = f * 2^k, f in [0.5,1) (0 for x = 0): single VALU instructions
// (v_ldexp_f32 / v_frexp_exp_i32_f32) without the libm special-case wrappers.
__device__ __forceinline__ float fast_ldexp(float x, int e) { return __builtin_amdgcn_ldexpf(x, e); }
__device__ __forceinline__ int frexp_exp(float x) { return __builtin_amdgcn_frexp_expf(x); }

// Sum over aligned groups of G lanes (G = 2, 4, 8, 16), result in every lane of the group, on the VALU
// through DPP lane selects (quad_perm / row_half_mirror / row_mirror) -- unlike __shfl_xor, which goes
// through the LDS crossbar (ds_bpermute_b32) and serialises behind the CU's other LDS traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group_sum: group width");
  if constexpr (G >= 2) v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  if constexpr (G >= 4) v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v += dpp_move<0x141>(v);  // row_half_mirror
  if constexpr (G >= 16) v += dpp_move<0x140>(v); // row_mirror
  return v;
}


// ---- direct global -> LDS copies and inter-workgroup hand-offs (MI355X_MICROARCH.md, "Workgroup dispatch ...")
// 16 bytes per lane from `src` (per-lane global address) to `lds_wave_base + lane*16` (the LDS destination of an
// LDS-DMA is wave-uniform base + lane*size: global_load_lds_dwordx4).  Asynchronous: counted on vmcnt; the data is
// ordered for a ds_read only by the issuing wave's s_waitcnt vmcnt + a barrier the reader has passed.
__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  // Issued from inline asm on purpose: hipcc orders every ds_read behind a builtin LDS-DMA it cannot disambiguate
  // (s_waitcnt vmcnt(0) in front of the first operand fetch of each K tile, which serialises the load with the MFMAs
  // it is meant to overlap).  An asm LDS-DMA is outside the compiler's counter bookkeeping: the caller waits
  // (vm_drain) and barriers before the data is read.  M0 (LDS base of the DMA) is saved and restored.
  const unsigned dst = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(dst)
      : "memory");
}
// The same with a wave-UNIFORM source chunk: `base` lives in a scalar register pair and the lanes share one 32-bit byte
// offset register (lane * 16 for a contiguous 1 KB chunk) -- no per-piece address VGPRs in a loop that needs its vector
// registers for accumulators and operand fragments; the per-stage address arithmetic runs on the scalar unit.
__device__ __forceinline__ void glds16_uniform(const float* base, unsigned lane_byte_offset, float* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_byte_offset), "s"(base), "s"(dst)
      : "memory");
}
// Loads of data a kernel streams ONCE (the K/V memory of a cross-attention step, the CTC posteriors): NT = the non-temporal cache
// policy (global_load_dword[x4] ... nt: the line is not kept in the CU's vector L1 and is marked for early eviction in L2 / the
// Infinity Cache, so that operands other kernels re-use -- weight panels, the residual stream -- survive the stream);
// false = the default policy.  A template parameter, not a run-time branch: the policy is part of the instruction.
template <bool NT>
__device__ __forceinline__ float4 ld16(const float* p) {
  if constexpr (NT) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
template <bool NT>
__device__ __forceinline__ float ld4(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
// a value the caller knows to be the same in every lane, moved to a scalar register (uniform branches, scalar address math)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// every outstanding vector-memory operation of this wave has completed (inline asm: the compiler cannot drop it)
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same wait as an instruction the compiler's own counter bookkeeping SEES (gfx9 encoding: vmcnt 0, expcnt / lgkmcnt untouched).
// For the end of an epilogue in a kernel whose loop issues its LDS-DMA from inline asm: with stores still "in flight" in the compiler's
// books it places its own s_waitcnt vmcnt(0) in front of the first instruction of the loop that overwrites one of their registers --
// wherever that is, e.g. right behind the asm LDS-DMA of every K tile (csrc/gemm_lp256.hip, profiles/r06_ac_*).
__device__ __forceinline__ void vm_drain_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// at most N of this wave's vector-memory operations still in flight (they retire in issue order)
template <int N>
__device__ __forceinline__ void vm_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// every outstanding LDS operation of this wave has completed (reads: their data is in registers)
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// the bare workgroup barrier: no memory-counter waits are attached (a __syncthreads() drains vmcnt, i.e. every LDS-DMA in
// flight); the caller has waited for exactly what the barrier must publish (lds_drain / vm_wait<N>)
__device__ __forceinline__ void block_barrier_raw() { __builtin_amdgcn_s_barrier(); }
// publish this workgroup's earlier plain stores to every CU of the device (one lane, after a __syncthreads())
__device__ __forceinline__ void release_agent() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// drop this CU's stale L1 lines before reading what another workgroup published (one lane, then __syncthreads())
__device__ __forceinline__ void acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ int atomic_add_agent(int* p, int v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_store_agent(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int atomic_load_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- data handed from one workgroup to another INSIDE a launch (the persistent few-row decoding step, csrc/decoder_persist.hip).
// An MI355X has eight XCDs with private, mutually non-coherent L2s: a plain store may sit in the writer's L2 and a plain load
// may hit a stale line of the reader's.  Agent-scope relaxed atomics compile to sc1 accesses that are served at the
// device-coherent level, so no L2 write-back / invalidate (buffer_wbl2 / buffer_inv: 2-10 us per barrier on this part,
// tools/persist_probe.hip) is ever needed for such data; a writer only has to drain its own stores (vm_drain) before it
// signals.  Checked on the box: 0 mismatches over 500 rounds of all-to-all hand-over (profiles/r05_a_*).
__device__ __forceinline__ float ld_agent(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float2 ld_agent2(const float* p) {  // 8-byte aligned
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__builtin_bit_cast(float, (unsigned)(u & 0xffffffffull)), __builtin_bit_cast(float, (unsigned)(u >> 32)));
}
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Barrier among the G co-resident workgroups of a cooperative launch: `ctr` counts arrivals monotonically (the caller passes
// the value it must reach); every lane first drains its own sc1 stores, one ticket per workgroup.  1.3 us at G = 64, 2.0 at 128.
// 100 MHz wall clock shared by every CU (measurement only)
__device__ __forceinline__ long long wall_clock() { return (long long)wall_clock64(); }
// In two halves, so that loads which do not depend on the other workgroups (the next projection's weights) can be issued
// between the arrival and the wait and fly while the barrier completes.
__device__ __forceinline__ void grid_arrive(int* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same arrival on a two-level counter: this workgroup's sub-counter `sub` (one of up to eight, each on its own 128-byte line) and,
// from the arrival that completes the sub-counter's round (`sub_target` = rounds x its member count), the top counter `top` the
// waiters poll.  Atomics on one address are totally ordered, so the completing arrival is ordered after every other member's --
// and each of those after its workgroup's drained stores: a waiter that sees the top target sees all of them.
__device__ __forceinline__ void grid_arrive_tree(int* sub, int sub_target, int* top) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == sub_target) __hip_atomic_fetch_add(top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void grid_wait(int* ctr, int target) {
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// Products / sums / differences that the compiler must NOT contract into a fused multiply-add with a neighbouring
// operation: two kernels that are required to produce bit-identical values (the fused scoring pass and the launches it
// replaces) write the shared arithmetic with these, so the result does not depend on what each kernel's optimiser fuses.
// (HIP's __fmul_rn / __fadd_rn are plain operators the optimiser is free to contract -- measured: the per-token
// log-probs of the two paths differed in the last bit at temperature != 1; an operation compiled under
// `fp contract(off)` carries no contract flag and cannot be fused with a neighbour, inlined or not.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// Arg-max over the wave: on return every lane holds the largest v of the wave and, among the lanes that supplied it, the
// smallest i (candidate ids are unique, so exactly one lane recognises its own pair).  All 64 lanes must be active.  Two
// reductions on the VALU (DPP lane selects inside each row of 16, v_readlane across the four rows) instead of 2 x 6
// ds_bpermute round trips through the LDS crossbar per round of a selection loop.
template <int CTRL>
__device__ __forceinline__ int dpp_move_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_max_valu(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  const int b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ int wave_min_valu(int v) {
  v = min(v, dpp_move_i<0xB1>(v));
  v = min(v, dpp_move_i<0x4E>(v));
  v = min(v, dpp_move_i<0x141>(v));
  v = min(v, dpp_move_i<0x140>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
  const float m = wave_max_valu(v);
  i = wave_min_valu(v == m ? i : 0x7fffffff);
  v = m;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
  return v;
}

}  // namespace sbk

// Dynamic LDS window of the launching workgroup.
// lambdas of a kernel body must be inlined (a called lambda takes its captures through scratch memory)
#define SBK_INLINE_LAMBDA __attribute__((always_inline))
#define SBK_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]

// Raise a kernel's dynamic-LDS window above the 64 KiB default (gfx950 has 160 KiB per CU).
#define SBK_ALLOW_DYN_LDS(kernel, bytes) \
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

// Cooperative launch (every workgroup resident at once: the kernel may use sbk::grid_barrier): ONE by-value argument struct;
// evaluates to the hipError_t of the launch.  SBK_COOP_MAX_GRID: the largest grid the device can hold for this kernel.
#define SBK_LAUNCH_COOP(kernel, grid, block, lds_bytes, stream, arg_struct) \
  ([&]() -> hipError_t { void* p__[] = {(void*)&(arg_struct)};               \
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&kernel), grid, block, p__, lds_bytes, stream); }())
#define SBK_COOP_MAX_GRID(kernel, block_threads, lds_bytes, out_int)                                                   \
  ([&]() -> hipError_t { int dev__ = 0, cus__ = 0, per__ = 0; hipError_t e__ = hipGetDevice(&dev__);                    \
    if (e__ == hipSuccess) e__ = hipDeviceGetAttribute(&cus__, hipDeviceAttributeMultiprocessorCount, dev__);         \
    if (e__ == hipSuccess) e__ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per__, kernel, block_threads, lds_bytes); \
    (out_int) = cus__ * per__; return e__; }())

// kernel<<<grid, block, lds_bytes, stream>>>(args...)
#define SBK_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, __VA_ARGS__)
