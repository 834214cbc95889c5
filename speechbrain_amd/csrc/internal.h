// C++-level entry points shared between translation units (the fused pipelines
// call the same launchers the C ABI exposes, without re-validating arguments).
#pragma once
#include "common.h"

namespace sbk {
int gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
            int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq,
            hipStream_t st);
// The same contraction on the bf16 matrix pipe (csrc/gemm.hip: exact three-way operand split); W3 = sbk_split_bf16x3's
// image of W.  x3_routed: shapes it is measured to win on (tools/microbench.py --x3 --x3-decode); -1 = no workspace.
int gemm_nt_x3(const float* A, int lda, const uint16_t* W3, const float* bias, const float* R, int ldr, float* C, int ldc,
               int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st);
bool x3_routed(int M, int N, int K);
// both operands as panel images (csrc/gemm_x3p.hip); PC: optional panel image of the result; -1 = no workspace
int gemm_nt_x3p(const uint16_t* PA, const uint16_t* PW, const float* bias, const float* R, int ldr, float* C, int ldc,
                uint16_t* PC, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, hipStream_t st);
// few-row contraction on the bf16 matrix pipe (csrc/gemm_x3r.hip): A fp32 [M, K], PW = the panel image of W (sbk_split_x3p);
// -1 = shape not eligible
int gemm_nt_x3r(const float* A, int lda, const uint16_t* PW, const float* bias, const float* R, int ldr, float* C, int ldc, int M,
                int N, int K, int act, float alpha, hipStream_t st);
bool x3r_routed(int M, int N, int K);
// the same with the LayerNorm over K in its prologue (affine folded into PWf / bf; row statistics by a pre-pass over the rows);
// x3r_ln_routed: knob 45 and a K the pre-pass takes
int gemm_ln_nt_x3r(const float* A, int lda, const uint16_t* PWf, const float* bf, const float* R, int ldr, float* C, int ldc, int M,
                   int N, int K, float eps, int act, float alpha, hipStream_t st);
bool x3r_ln_routed(int K);
extern int g_x3r_mode, g_x3r_min_rows, g_x3r_ln, g_x3r_probe, g_x3r_pair;
// The decoding step of <= 16 hypothesis rows as ONE cooperative launch (csrc/decoder_persist.hip; keys 47 / 48).
// persist_eligible: shapes / weights it takes (head_dim 64, folded LayerNorm weights present, <= 16 layers); decoder_step_persist
// returns -1 when the launch cannot be made (the caller then issues the launch-per-operation step).
bool persist_eligible(const sbk_decoder_weights* W, int n, int B, int beam, int Lmax);
int persist_barriers(int n_layers);
int decoder_step_persist(const sbk_decoder_weights* W, const int32_t* tokens, const int32_t* kv_slot, const int32_t* enc_len,
                         float* x, float* qkv, float* ctx, float* q, float* ff, float* h, float* logits, float* const* kcache,
                         float* const* vcache, float* const* ckv, int* bar, int bar_seq, int* grid_io, int step, int n, int B, int T,
                         int beam, int Lmax, bool want_logits, hipStream_t st);
extern int g_persist, g_persist_grid, g_persist_stamps, g_persist_tree;
// Device-resident step counter of the search running on this host thread (nullptr: the step is the
// launch argument).  When set, every step-dependent kernel reads the step from it, so that the launches
// of one decoding step are identical for every step and can be replayed from a captured hipGraph.
extern thread_local const int32_t* g_step_ptr;
extern thread_local int g_step_min_steps;  // min_decode_steps of that search (eos floor: step < min_steps)
extern int g_cross_rows;
extern int g_nt_mask;
extern int g_self_anc;
extern int g_attn_exp2;
extern int g_cross_fc256;
// bf16 x bf16 / e4m3 x e4m3 contraction on 256 x 256 tiles (csrc/gemm_lp256.hip): the large shapes of sbk_gemm_nt_bf16a / _fp8a.
// lda / ldw in BYTES; KT = bytes of a row of K / 128; sa / sw / C8 only with fp8 operands.  lp256_routed: key 61 and the shape.
struct Lp256Args {
  const unsigned char* A;
  const unsigned char* W;
  const float* sa;
  const float* sw;
  const float* bias;
  const float* R;
  float* C;
  unsigned short* Cb;
  unsigned char* C8;
  float c8_scale;
  long lda, ldw;
  int ldr, ldc, ldcb, ldc8, M, N, act;
  float alpha;
  int KT, tiles_m, tiles_n, tiles, whole;
};
bool lp256_routed(const Lp256Args& a);
int gemm_nt_lp256(const Lp256Args& a, bool fp8, hipStream_t st);
extern int g_lp256, g_lp256_mode, g_x3p_mode, g_x3p_fast_epi;
extern int g_x3r_xc;
extern int g_score_fused;  // key 40: 1 (default) = the step's scoring as one pass per hypothesis row (csrc/search.hip)
constexpr float kCtcNeg = -1e20f;  // the CTC scorer's finite "log 0" (ctc.py:150, scorer.py:1250)
// The arithmetic of a decoding step's scoring, shared by the separate kernels (log_softmax_row / ctc_combine / am_only /
// beam_topk_stage1) and the fused pass (score_topk_row_kernel).  Written with explicitly rounded operations so that
// the two paths cannot differ by what each kernel's optimiser contracts into a fused multiply-add.
__device__ __forceinline__ float ls_logit(float x, float b1, float b2, float inv_temp) {  // (logit + masks) / temperature
  return mul_rn(add_rn(add_rn(x, b1), b2), inv_temp);
}
__device__ __forceinline__ float ls_out(float v, float lse, float w) { return mul_rn(w, sub_rn(v, lse)); }  // w * log-softmax
// log_probs += score * weight with score = psi - psi_prev (scorer.py:1248-1253, ctc.py:259-262): the reference rounds the
// difference, the product and the sum separately (three torch operations), so no fused multiply-add here
__device__ __forceinline__ float score_ctc(float v, float psi, float psi_prev, float weight) {
  return add_rn(v, mul_rn(sub_rn(psi, psi_prev), weight));
}
__device__ __forceinline__ float score_cand(float seq, float comb, float norm) {  // seq2seq.py:1225-1240
  const float x = add_rn(seq, comb);
  return norm > 0.0f ? x / norm : x;  // length normalisation divides, like seq2seq.py:1232-1233
}
int gemm_nt_ws(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
               int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len, int rows_per_seq, float* ws,
               size_t ws_floats, hipStream_t st);
int gemm_ln_nt(const float* A, int lda, const float* Wf, int ldw, const float* bf, const float* R, int ldr, float* C,
               int ldc, int M, int N, int K, float eps, int act, float alpha, hipStream_t st);
int layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int d, float eps, int act,
              hipStream_t st);
int relpos_attention(const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                     const int32_t* key_len, float* out, float* attn, int B, int T, int H, int Dh, float scale,
                     hipStream_t st, int chunk = 0, int left = -1);
int glu_dwconv(const float* h, const float* w, const float* bias, float* y, int B, int T, int d, int ksize,
               hipStream_t st, int chunk = 0);
}  // namespace sbk
