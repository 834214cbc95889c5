/* sbk.h -- C ABI of libsbk_hip.so, the MI355X (gfx950) kernels behind the
 * SpeechBrain EncoderDecoderASR hot path.
 *
 * The reference (speechbrain v1.1.0) is pure Python: it has no FFI/plugin
 * registry; each entry point below replaces the ATen call(s) made by the cited
 * reference function (paths relative to /root/reference/speechbrain).  The
 * reference-side binding (a ctypes stub per module) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is DEVICE memory owned by the caller (PyTorch-allocated);
 *    the library never allocates or frees device memory and retains no pointer past the call -- with ONE
 *    documented exception, the stream workspace below, which the caller registers and may take back;
 *  - tensors are row-major contiguous fp32 unless stated; lengths are int32;
 *  - calls are asynchronous on `stream` (a hipStream_t) and never synchronise;
 *  - return 0 on success, a negative errno-style code for bad arguments
 *    (SBK_EINVAL = -22), a positive hipError_t if a launch failed;
 *    sbk_last_error() returns the thread-local message of the last failure;
 *  - a call whose batch dimension is 0 returns 0 without touching its data pointers (they may be NULL);
 *  - re-entrant: concurrent calls from several host threads (one stream each) are supported; the
 *    searchers keep their per-call state in the caller's workspace and a per-thread helper stream.
 */
#ifndef SBK_H_
#define SBK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBK_ABI_VERSION 11

typedef void* sbk_stream_t; /* hipStream_t */

int sbk_abi_version(void);
const char* sbk_last_error(void);

/* ---- stream workspace (ABI 7) ------------------------------------------------------------------------------
 * The persistent / stream-K contraction kernels (sbk_gemm_nt_f32 from ~2 048 rows on, sbk_gemm_nt_f32x3, the searches'
 * memory / CTC / vocabulary projections) finish cut tiles through partial-tile slabs and arrival tickets in device
 * memory.  That memory is the CALLER's: `sbk_stream_workspace_bytes()` bytes, 256-byte aligned, registered once per
 * (current device, stream) with sbk_stream_workspace_set -- which zeroes the tickets on that stream -- and used by every
 * later call on that stream (launches of one stream are ordered, so they share it; two streams must not share one).
 * The pointer is retained until sbk_stream_workspace_release(stream) (or a second _set for the same stream); the caller
 * frees the memory afterwards, once the stream has drained.  Without a registered workspace sbk_gemm_nt_f32 and the searches
 * use their tile-grid kernels (same result class, another summation order) and sbk_gemm_nt_f32x3 returns SBK_EINVAL.
 * speechbrain_amd.native registers one torch-allocated workspace per stream it is called on (native._stream). */
size_t sbk_stream_workspace_bytes(void);
int sbk_stream_workspace_set(sbk_stream_t stream, void* workspace, size_t workspace_bytes);
int sbk_stream_workspace_release(sbk_stream_t stream);

/* Per-launch timing with HIP events recorded on each launch's own stream (bench.py's roofline
 * leg).  sbk_prof_report writes "name count total_ms algorithmic_flops algorithmic_bytes" lines. */
void sbk_prof_enable(int on);
void sbk_prof_reset(void);
size_t sbk_prof_report(char* buf, size_t cap);
/* mean microseconds per launch of C[M,N] = A[M,K].W[N,K]^T over `iters` back-to-back launches
 * (tools/microbench.py; synchronises the stream). us_per_launch is a HOST pointer. */
int sbk_prof_gemm_repeat_f32(const float* A, const float* W, float* C, int M, int N, int K, float* workspace,
                             size_t workspace_floats, int iters, float* us_per_launch, sbk_stream_t stream);

/* Measurement: phase time stamps (100 MHz ticks) of the most recent persistent few-row decoding step launched with knob 49 set
 * (csrc/decoder_persist.hip): [start, (end of phase, end of barrier) x barriers, end] of workgroup 0.  Synchronises the device;
 * returns the number of stamps written to the HOST array `out`.  tools/latency_probe.py --stamps prints them. */
int sbk_prof_persist_stamps(long long* out, int cap);

/* tuning knobs for experiments (key 1: K chunks per fetch batch of the skinny GEMM, 0 = automatic) */
void sbk_prof_set_knob(int key, int value);
/* ABI 11: the current value of a switch (INT_MIN for an unknown key), so that an experiment or a test can restore what it found */
int sbk_prof_get_knob(int key);
/* HBM calibration (SURVEY 8d "measure achievable with a copy kernel first"): hand-written float4 streaming kernels.
 * mode 0 = copy src -> dst (2 * 4 * n bytes per launch), mode 1 = read src (4 * n bytes; dst is a sink of >= 2048
 * floats).  n % 4 == 0, 16-byte aligned.  Mean time of `iters` back-to-back launches in *us_per_launch (HOST). */
/* f32 matrix-core ceiling under the chip's power management: `wgs` workgroups of 4 waves issue iters x 16
 * v_mfma_f32_32x32x2_f32 each from registers (zero or pseudo-random operands); *tflops is a HOST out; sink >= wgs floats. */
int sbk_prof_mfma_peak_f32(float* sink, int wgs, int iters, int random_data, float* tflops, sbk_stream_t stream);
int sbk_prof_stream_f32(const float* src, float* dst, long n, int mode, int iters, float* us_per_launch,
                        sbk_stream_t stream);
/* same for the CTC prefix-score pass (x: [B,T,V] log-softmax rows, converted in place; work: scratch) */
int sbk_prof_ctc_psi_repeat_f32(float* x, const int32_t* enc_len, const int32_t* last_tok, float* psi, float* work,
                                int B, int T, int V, int beam, int prefix_len, int iters, float* us_per_launch,
                                sbk_stream_t stream);

/* ---- activations understood by fused epilogues --------------------------- */
enum { SBK_ACT_NONE = 0, SBK_ACT_SWISH = 1, SBK_ACT_GELU = 2, SBK_ACT_RELU = 3, SBK_ACT_LEAKY_RELU = 4 };

/* ---- a1: PCM16 frames -> float32 mono.  Replaces the sample conversion of audio_io.load (dataio/audio_io.py
 * :141-209: soundfile's float32 convention, sample / 32768) and AudioNormalizer's channel mean
 * (dataio/preprocess.py:76-84) for waveforms that are kept / shipped as int16 (inference/sharded.py scatters
 * them that way: half the xGMI and PCIe bytes of fp32).  pcm [frames, channels] interleaved -> out [frames]. */
int sbk_pcm16_to_f32(const int16_t* pcm, float* out, long frames, int channels, sbk_stream_t stream);

/* ---- a2-a5: Fbank -------------------------------------------------------
 * Replaces lobes/features.py:147-169 (Fbank.forward) = processing/features.py
 * :141-188 (STFT -> torch.stft), :341-378 (spectral_magnitude, power=1),
 * :512-586 (Filterbank.forward matmul) and :736-759 (_amplitude_to_DB with the
 * per-utterance max - top_db floor).
 *   wav      [B,N]            waveforms (zero right-padded)
 *   window   [n_fft]          analysis window, already centred in n_fft
 *   twiddle  [n_fft,2]        (cos, -sin)(2 pi m / n_fft), m = 0..n_fft-1
 *   radices  [n_radix]        host int array, product = n_fft, each in {2,3,4,5}
 *   mel_w    [nnz], mel_ptr [n_mels+1], mel_bin [n_mels]: CSR-by-filter form of
 *            the [n_stft,n_mels] filter matrix (each filter = one contiguous
 *            run of bins starting at mel_bin[m], weights mel_w[mel_ptr[m]..])
 *   out      [B,T,n_mels]     T = 1 + N/hop
 *   tile_max [B,ntiles]       workspace, ntiles = ceil(T/4)
 *   norm_mean/norm_std [n_mels] or NULL: fuse InputNormalization(global, eval)
 *            (processing/features.py:1404-1455) into the dB pass.
 */
int sbk_fbank_f32(const float* wav, const float* window, const float* twiddle, const int32_t* radices,
                  int n_radix, const float* mel_w, const int32_t* mel_ptr, const int32_t* mel_bin, float* out,
                  float* tile_max, int B, int N, int n_fft, int hop, int n_mels, int nnz, float amin, float top_db,
                  const float* norm_mean, const float* norm_std, float norm_eps, sbk_stream_t stream);

/* ---- SURVEY 8f / BASELINE.json configs[4]: the log-mel front-end of WhisperASR, integrations/huggingface/whisper.py
 * :276-316 (`log_mel_spectrogram`, adapted there from openai/whisper audio.py): torch.stft(audio, n_fft = 400, hop = 160,
 * hann_window, centre + REFLECT padding) -> last frame dropped -> power -> mel filters (the HF feature extractor's
 * slaney filters, CSR as for sbk_fbank_f32) -> log10(max(., 1e-10)) -> max(., max over the WHOLE BATCH - 8) -> (. + 4)/4.
 *   wav [B,N] (pad_or_trim'ed to 480000 by the caller) -> out [B, n_mels, N/hop] (mel-major, the encoder Conv1d's layout)
 *   tmp [B, N/hop, n_mels] and tile_max [B * ceil(N/hop/4)]: workspaces. */
int sbk_whisper_log_mel_f32(const float* wav, const float* window, const float* twiddle, const int32_t* radices,
                            int n_radix, const float* mel_w, const int32_t* mel_ptr, const int32_t* mel_bin, float* tmp,
                            float* tile_max, float* out, int B, int N, int n_fft, int hop, int n_mels, int nnz,
                            sbk_stream_t stream);

/* a2: STFT.forward (processing/features.py:141-188): [B,N] -> spec [B,T,n_fft/2+1,2] (re, im);
 * window / twiddle / radices as for sbk_fbank_f32. */
int sbk_stft_f32(const float* wav, const float* window, const float* twiddle, const int32_t* radices, int n_radix,
                 float* spec, int B, int N, int n_fft, int hop, sbk_stream_t stream);

/* a3: spectral_magnitude (processing/features.py:341-378): out[i] = (re^2 + im^2)^power over n complex
 * values (eps added first when power < 1), optionally log(. + eps). */
int sbk_spectral_magnitude_f32(const float* stft, float* out, long n, float power, int take_log, float eps,
                               sbk_stream_t stream);

/* a4: Filterbank._amplitude_to_DB (processing/features.py:736-759), in place on x [B,per_utt]:
 * multiplier*log10(max(x,amin)) - db_offset, then floor at (per-utterance max - top_db).
 * tile_max: workspace [B,64]. */
int sbk_amplitude_to_db_f32(float* x, float* tile_max, int B, long per_utt, float multiplier, float amin,
                            float db_offset, float top_db, sbk_stream_t stream);

/* a6: InputNormalization.forward, eval, norm_type="global" (features.py:1404-1455):
 * y = (x - mean[c]) / max(std[c], eps), x [rows, C]. */
int sbk_input_norm_global_f32(const float* x, const float* mean, const float* std, float* y, int rows, int C,
                              float eps, sbk_stream_t stream);
/* ABI 8: the same with avoid_padding_norm=True (features.py:1447-1449): x, y [B,T,C]; the padded frames t >= n_valid[b]
 * are normalised with mean 0 / std 1, i.e. pass through unchanged. */
int sbk_input_norm_global_masked_f32(const float* x, const float* mean, const float* std, const int32_t* n_valid, float* y,
                                     int B, int T, int C, float eps, sbk_stream_t stream);

/* a6: InputNormalization.forward with norm_type="sentence" (per_batch = 0: mean / std of each utterance's own
 * valid frames, features.py:1432-1433,1482-1490) or "batch" (per_batch = 1: of all valid frames of the batch, variance
 * clamped at eps, :1434-1436; gaussian_statistics :997-1089).  x, y [B,T,C]; n_valid[b] = number of unpadded frames
 * (make_padding_mask :1535-1613: frames t < lengths[b]*T - 1e-6).  Two-pass moments as in the reference; every frame,
 * padded ones included, is normalised unless avoid_padding_norm (then padded frames pass through, :1451-1453).
 * workspace: sbk_input_norm_stats_workspace_bytes(B, C) bytes. */
size_t sbk_input_norm_stats_workspace_bytes(int B, int C);
int sbk_input_norm_stats_f32(const float* x, const int32_t* n_valid, float* y, float* workspace, int B, int T, int C,
                             int per_batch, int std_norm, float eps, int avoid_padding_norm, sbk_stream_t stream);

/* ---- dense contraction ---------------------------------------------------
 * C[M,N] = epilogue(A[M,K] . W[N,K]^T): replaces every F.linear / 1x1 Conv1d on
 * the path (nnet/linear.py:44-91, attention.py:623,915-947, Conformer.py
 * :129,155, Transformer.py decoder projections).  fp32 MFMA (exact f32 fma).
 *   v = acc + bias[n] (bias may be NULL); v = act(v);
 *   if seq_len: rows are [batch][rows_per_seq] and v = 0 for row-in-seq >= seq_len[batch]
 *               (ConvolutionModule's masked_fill_ of padded frames, Conformer.py:327-328)
 *   C = (residual ? residual[m,n] : 0) + alpha * v
 * lda / ldw / ldc / ldr are row strides in elements.  lda < K is allowed (A is read-only): consecutive rows then
 * overlap -- a window of K floats sliding by lda over a time-major signal, which is how the Conv1d layers of the
 * Whisper encoder (kernel 3, stride 1 / 2; transformers modeling_whisper.WhisperEncoder, called at
 * integrations/huggingface/whisper.py:372) run as GEMMs without an im2col copy; A must hold (M-1)*lda + K floats. */
int sbk_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* residual,
                    int ldr, float* C, int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len,
                    int rows_per_seq, sbk_stream_t stream);

/* The SAME fp32 contraction on the bf16 matrix pipe (ABI 6) -- the large encoder contractions of nnet/attention.py:623,
 * 941-945 (in_proj / out_proj), Conformer.py:129,155 (macaron feed-forward), :230-330 (pointwise convolutions) take this
 * entry.  An fp32 number is exactly the sum of three bf16 numbers (hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid,
 * round to nearest even); of the nine partial products of a.w the six of relative size >= 2^-17 are formed exactly on
 * v_mfma_f32_32x32x16_bf16 (16x the per-cycle rate of v_mfma_f32_32x32x2_f32) and accumulated in fp32, the three
 * dropped ones are <= 2^-26 |a||w| each, of either sign -- below the rounding of one fp32 multiply-add.  Same result class as
 * sbk_gemm_nt_f32 (measured against fp64: no larger error than the fp32 MFMA chain), NOT a reduced-precision path.
 *   sbk_split_bf16x3: W [N,K] fp32 (row stride ldw) -> W3 [N][K/32][3][32] bf16 bits (hi, mid, lo pieces; once per
 *   weight matrix, 6 bytes per element).  K % 32 == 0.
 *   sbk_gemm_nt_f32x3: C = epilogue(A . W^T) with A [M,K] fp32 in memory (cut into its pieces in registers), epilogue /
 *   lda < K / seq_len exactly as sbk_gemm_nt_f32.  K % 32 == 0, K >= 64, lda % 4 == 0, A and W3 16-byte aligned; rows of
 *   C (and of residual / bias) are moved as 16-byte vectors: N % 4 == 0, ldc % 4 == 0, ldr % 4 == 0, 16-byte aligned. */
int sbk_split_bf16x3(const float* W, int ldw, uint16_t* W3, int N, int K, sbk_stream_t stream);
int sbk_gemm_nt_f32x3(const float* A, int lda, const uint16_t* W3, const float* bias, const float* residual, int ldr,
                      float* C, int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len,
                      int rows_per_seq, sbk_stream_t stream);

/* The same contraction with BOTH operands pre-split and stored in PANEL layout (ABI 7; csrc/gemm_x3p.hip) -- the entry the
 * encoder's contractions take from ~2 000 rows on.  Same arithmetic and result class as sbk_gemm_nt_f32x3 (six exact bf16
 * partial products per element pair, fp32 accumulation); the A operand is no longer split inside the loop (once per element
 * instead of once per column tile), and 256 x 256 tiles fed by a three-slot LDS-DMA ring keep the matrix pipe busy.
 *   panel image of X [rows, K] (K % 16 == 0): [ceil(rows/64)][K/16][3 pieces hi|mid|lo][2 halves of 8 k][64 rows][8 k] bf16
 *   bits, sbk_x3p_panel_bytes(rows, K) = ceil(rows/64)*64*K*6 bytes, 16-byte aligned: every 1 KB chunk is one
 *   global_load_lds_dwordx4 and is read back as MFMA fragments without bank conflicts.
 *   sbk_split_x3p: X [rows, K] fp32 (row stride ldx; ldx < K allowed) -> its panel image (weights: once per matrix;
 *   activations: per call, unless the producing call wrote the image itself -- `PC` below).
 *   sbk_gemm_nt_x3p: epilogue / seq_len exactly as sbk_gemm_nt_f32; the result goes to C (fp32, may be NULL when PC is given)
 *   and / or to PC, the panel image of the [M, N] result (N % 16 == 0) = the A operand of the next contraction
 *   (PositionalwiseFeedForward, nnet/attention.py:941-945: linear -> activation -> linear without an fp32 round trip).
 *   Needs the stream workspace when the launch has more tiles than CUs. */
size_t sbk_x3p_panel_bytes(int rows, int K);
/* ABI 8: y = act(LayerNorm(x)) (nn.LayerNorm of Conformer.py:129,155,310,318 / attention.py:915) written DIRECTLY as the
 * panel image of the [rows, d] result, i.e. as the A operand of the sbk_gemm_nt_x3p that consumes it (d % 16 == 0,
 * d <= 2048; padding rows of the last 64-row block are written as zeros).  Same statistics as sbk_layernorm_f32 (two-pass
 * mean / variance in fp32); split(LayerNorm) is exact, so the contraction sees the fp32 values. */
int sbk_layernorm_x3p(const float* x, const float* gamma, const float* beta, uint16_t* P, int rows, int d, float eps,
                      int act, sbk_stream_t stream);
int sbk_split_x3p(const float* X, int ldx, uint16_t* P, int rows, int K, sbk_stream_t stream);
int sbk_gemm_nt_x3p(const uint16_t* PA, const uint16_t* PW, const float* bias, const float* residual, int ldr, float* C,
                    int ldc, uint16_t* PC, int M, int N, int K, int act, float alpha, const int32_t* seq_len,
                    int rows_per_seq, sbk_stream_t stream);

/* ABI 8 (signature of ABI 10): the same arithmetic for FEW rows -- the linear layers of a decoding step (Transformer.py:751-834:
 * self-attention in / out projections, cross-attention query / out projections, the feed-forward pair; a few hundred to a few
 * thousand hypothesis rows).  A fp32 [M, K] (row stride lda; split in registers), PW the panel image of W [N, K]
 * (sbk_split_x3p); 64 x 64 tiles whose four waves split K; N % 4 == 0, K % 256 == 0.  Epilogue: residual + alpha * act(. + bias).
 * No workspace: K is never split across workgroups (fixed summation order: the four waves' K quarters in order).
 * (ABI 8-9 also took A as a panel image and could write the result as one: measured no faster and removed in ABI 10.) */
int sbk_gemm_nt_x3r(const float* A, int lda, const uint16_t* PW, const float* bias, const float* residual, int ldr, float* C,
                    int ldc, int M, int N, int K, int act, float alpha, sbk_stream_t stream);
/* ABI 9 (signature of ABI 10): the LayerNorm in front of such a projection (Transformer.py:751-834 with normalize_before:
 * norm1 -> self-attention in_proj, norm2 -> cross-attention query rows, norm3 -> ffn.0; decoder.norm -> seq_lin) inside the
 * projection's launch: C = residual + alpha * act(LayerNorm(A) . W^T + b) with the affine folded into the operands exactly as
 * for sbk_gemm_ln_nt_f32 -- PWf = panel image (sbk_split_x3p) of Wf[n,k] = W[n,k] gamma[k], bf[n] = b[n] + sum_k W[n,k] beta[k].
 * Row statistics in the two-pass form of sbk_layernorm_f32 (biased variance, eps inside the root), computed by the
 * workgroup for its 64 rows while its first operand loads are in flight; x - mean is split, rstd scales the finished
 * tile.  A fp32 [M, K], lda % 4 == 0; K = 256, 512, 1 024 or 1 280; N % 4 == 0.
 * (ABI 9's sbk_gemm_nt_x3r_stats / sbk_row_block_stats_f32 -- row statistics handed over by the kernels that write the rows --
 * measured equal to this pre-pass on the GPU and were removed in ABI 10: profiles/r05_a_*.) */
int sbk_gemm_ln_nt_x3r(const float* A, int lda, const uint16_t* PWf, const float* bf, const float* residual, int ldr, float* C,
                       int ldc, int M, int N, int K, float eps, int act, float alpha, sbk_stream_t stream);

/* ---- bf16-operand fast entry points ("fp32 parity entry points plus bf16 ... fast entry points").
 * C = epilogue(bf16(A) . Wb^T) with fp32 accumulation on v_mfma_f32_32x32x16_bf16: A [M,K] stays fp32 in memory and is
 * rounded to bf16 (nearest even) inside the kernel, Wb [N,K] holds the weights as bf16 bits (sbk_f32_to_bf16, once per
 * model), bias / activation / residual / row mask exactly as sbk_gemm_nt_f32.  Opt-in (run_opts precision="bf16",
 * cf. utils/run_opts.py:114-115 / utils/autocast.py): NOT the parity path -- products carry 8 mantissa bits, see
 * DESIGN.md for the stated tolerance and the token-agreement rate.  K % 8 == 0, ldw % 8 == 0, lda % 4 == 0. */
int sbk_f32_to_bf16(const float* x, uint16_t* y, long n, sbk_stream_t stream);
int sbk_gemm_nt_bf16(const float* A, int lda, const uint16_t* Wb, int ldw, const float* bias, const float* residual,
                     int ldr, float* C, int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len,
                     int rows_per_seq, sbk_stream_t stream);

/* bf16 activations BETWEEN the bf16 contractions (the Whisper encoder under precision="bf16",
 * integrations/huggingface/whisper.py:190-232 <- speechbrain/integrations/huggingface/whisper.py:318-353 forward_encoder):
 * A [M,K] and Wb [N,K] both bf16 in memory (K % 64 == 0, rows 16-byte aligned), panels global -> LDS by LDS-DMA through
 * a 4-stage pipeline, fp32 accumulation, epilogue as sbk_gemm_nt_f32 (bias, activation, alpha, fp32 residual) written as
 * fp32 (C) and / or as bf16 (Cb: the next contraction's operand, rounded to nearest even -- the same value the fp32-A
 * kernel above would round on its way into LDS, so the two paths differ by summation order only).
 * sbk_layernorm_bf16o / sbk_rope_attention_bf16o are the producers of such operands: LayerNorm / attention context
 * written as bf16.  Round 6: from 128 tiles of 256 x 256 on (96 for an fp32 result with a residual; 16-byte aligned rows of
 * C / residual / bias, N % 4 == 0) this entry and sbk_gemm_nt_fp8a run csrc/gemm_lp256.hip -- the same sums in the same order,
 * bit-identical results; the GELU of these two entries is erfc by Abramowitz-Stegun 7.1.26 (|error| <= 4.2e-7 in fp32). */
int sbk_gemm_nt_bf16a(const uint16_t* A, int lda, const uint16_t* Wb, int ldw, const float* bias, const float* residual,
                      int ldr, float* C, int ldc, uint16_t* Cb, int ldcb, int M, int N, int K, int act, float alpha,
                      sbk_stream_t stream);
int sbk_layernorm_bf16o(const float* x, const float* gamma, const float* beta, uint16_t* y, int rows, int d, float eps,
                        int act, sbk_stream_t stream);
int sbk_rope_attention_bf16o(const float* qkv, const float* cosines, const float* sines, const int32_t* key_len,
                             uint16_t* out, int B, int T, int H, int Dh, int table_rows, float scale, int chunk_size,
                             int left_chunks, sbk_stream_t stream);
/* Plain scaled-dot-product attention with bf16 rows in and out (head_dim 64): qkv [B,T,H,3,64] bf16 as written by
 * sbk_gemm_nt_bf16a, out [B,T,H*64] bf16; workspace (sbk_attention_bf16io_workspace_bytes) receives the transposed
 * values.  K rows and V^T rows of a 64-key tile are shared by the 128 queries of a workgroup through LDS (LDS-DMA);
 * fp32 scores / softmax statistics / context accumulation, probabilities rounded to bf16 as in
 * sbk_rope_attention_bf16 (nnet/attention.py:941-945 scaled_dot_product_attention of the HF Whisper encoder layer). */
size_t sbk_attention_bf16io_workspace_bytes(int B, int T, int H);
int sbk_attention_bf16io(const uint16_t* qkv, const int32_t* key_len, uint16_t* out, uint16_t* workspace, int B, int T,
                         int H, int Dh, float scale, sbk_stream_t stream);

/* The same contraction with fp16 operands (v_mfma_f32_32x32x16_f16; A rounded to nearest even on its way into LDS, Wh =
 * the weights converted once by sbk_f32_to_f16).  fp16 has 3 more mantissa bits than bf16 and a narrower range
 * (|x| <= 65504): the tolerance against the fp32 product is 2^-10 relative per operand. */
int sbk_gemm_nt_f16(const float* A, int lda, const uint16_t* Wh, int ldw, const float* bias, const float* residual,
                    int ldr, float* C, int ldc, int M, int N, int K, int act, float alpha, const int32_t* seq_len,
                    int rows_per_seq, sbk_stream_t stream);
int sbk_f32_to_f16(const float* x, uint16_t* y, long n, sbk_stream_t stream);

/* fp8 (OCP e4m3fn) operands on v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulation, per-tensor scales:
 *   C = epilogue( (a_absmax/448 * w_scale) * ( e4m3(A * 448/a_absmax) . Wq^T ) )
 * Wq [N,K] = e4m3(W / w_scale) (sbk_f32_to_fp8 with mul = 1/w_scale, w_scale = max|W| / 448); a_absmax = DEVICE scalar
 * max|A| (sbk_absmax_f32: no host round trip).  K % 16 == 0, ldw % 16 == 0.  e4m3 keeps 3 mantissa bits: expect ~2-3 %
 * relative RMS error against the fp32 product (tests/test_kernels.py states the bound).  The non-scaled fp8 MFMA runs
 * at the bf16 rate, so with fp32 activations in HBM this path saves only the weights' bytes. */
int sbk_gemm_nt_fp8(const float* A, int lda, const float* a_absmax, const uint8_t* Wq, int ldw, float w_scale,
                    const float* bias, const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                    float alpha, const int32_t* seq_len, int rows_per_seq, sbk_stream_t stream);
int sbk_f32_to_fp8(const float* x, uint8_t* y, long n, float mul, sbk_stream_t stream);
int sbk_absmax_f32(const float* x, long n, float* out, sbk_stream_t stream);

/* Same contraction for few-row operands (M <= 512: the beams x utterances rows of a decoder step):
 * waves own 32-column tiles and K slices, partial sums go through `workspace` (floats; the more,
 * the more K slices, at most 8*M*N is useful) and are combined in a fixed order. */
int sbk_gemm_nt_splitk_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                           const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int act,
                           float alpha, float* workspace, size_t workspace_floats, sbk_stream_t stream);

/* LayerNorm fused into the few-row contraction: C = epilogue(LN(A) . W^T + b) with gamma/beta pre-folded
 * by the caller: Wf[n,k] = W[n,k]*gamma[k], bf[n] = b[n] + sum_k W[n,k]*beta[k].  K in {128,256,512},
 * M <= 512 (SBK_EINVAL otherwise). */
int sbk_gemm_ln_nt_f32(const float* A, int lda, const float* Wf, int ldw, const float* bf, const float* residual,
                       int ldr, float* C, int ldc, int M, int N, int K, float eps, int act, float alpha,
                       sbk_stream_t stream);

/* ---- LayerNorm over the last dimension (nnet/normalization.py:185-242,
 * torch.nn.LayerNorm at Conformer.py:126,152,426,437): y = act(LN(x)), x [rows,d]. */
int sbk_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int d, float eps,
                      int act, sbk_stream_t stream);

/* ---- a7: one ConvolutionFrontEnd block (lobes/models/convolution.py:311-317):
 * reflect-pad(1) + Conv2d(3x3, stride 2, bias) (nnet/CNN.py:654-751) ->
 * LayerNorm over (F',C') eps (nnet/normalization.py:185-242) -> LeakyReLU(slope).
 *   x [B,Tin,Fin,Cin] -> y [B,Tout,Fout,Cout], Tout = (Tin-1)/2+1, Fout = (Fin-1)/2+1
 *   wt [Cin*9, Cout] = conv.weight[Cout,Cin,kF,kT] re-laid-out as row (ci*3+kf)*3+kt
 *   gamma/beta [Fout*Cout] */
int sbk_conv_block_f32(const float* x, const float* wt, const float* bias, const float* gamma, const float* beta,
                       float* y, int B, int Tin, int Fin, int Cin, int Cout, float eps, float slope,
                       sbk_stream_t stream);

/* ---- a12: RelPosMHAXL core (nnet/attention.py:555-742 minus the two Linear layers):
 *   qkv  [B,T,H,3*Dh] = F.linear(x, in_proj_weight) viewed per head as (q|k|v) (:623-626)
 *   pos  [2T-1,H*Dh]  = linear_pos(RelPosEncXL(T))                            (:655)
 *   bias_u/bias_v [H*Dh]: pos_bias_u/v storage read as (H,Dh)                  (:660-664)
 *   key_len [B] int32 or NULL: keys >= key_len[b] are masked (key_padding_mask)
 *   out  [B,T,H*Dh]   context before out_proj;  attn [B,H,T,T] or NULL (attention weights)
 *   scale = 1/sqrt(embed_dim) (:521).  Dh in {8,16,32,36,64}; T up to ~1100 (LDS strip) when attn is requested.
 *   chunk_size > 0: the Dynamic Chunk attention mask of make_transformer_src_mask (TransformerASR.py:47-103) --
 *   query i (chunk c = i / chunk_size) sees the keys [max(0, (c - left_chunks) * chunk_size), (c + 1) * chunk_size);
 *   left_chunks < 0: unlimited left context; chunk_size = 0: no mask.  Rows without any allowed key give a zero
 *   context (the post-softmax masked_fill of :713-728). */
int sbk_relpos_attention_f32(const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                             const int32_t* key_len, float* out, float* attn, int B, int T, int H, int Dh,
                             float scale, int chunk_size, int left_chunks, sbk_stream_t stream);

/* ---- RoPEMHA core (nnet/attention.py:1167-1392): what sits between in_proj and out_proj.
 *   out = softmax( rot(q) rot(k)^T scale , keys < key_len ) v,  scale = 1/sqrt(embed_dim) (:1272)
 *   qkv as for sbk_relpos_attention_f32; cosines / sines [table_rows >= T, Dh] = the buffers of
 *   PrecomputedRoPESinusoids (:955-1053): rot(x)[c] = x[c]*cos[t][c] + x[c^1]*sines[t][c].
 *   cosines = sines = NULL (attn must be NULL too): no rotation, i.e. plain scaled-dot-product self-attention -- the
 *   Whisper encoder's MHA (transformers WhisperAttention, reached from integrations/huggingface/whisper.py:372). */
int sbk_rope_attention_f32(const float* qkv, const float* cosines, const float* sines, const int32_t* key_len,
                           float* out, float* attn, int B, int T, int H, int Dh, int table_rows, float scale,
                           int chunk_size, int left_chunks, sbk_stream_t stream);

/* The same attention (no attention-weights output, head_dim 64) on the bf16 matrix cores, for the opt-in reduced-
 * precision path (run_opts precision="bf16"; the Whisper encoder's MHA at d = 1280 / 20 heads is its main user): q, k,
 * v and the probabilities are rounded to bf16 into v_mfma_f32_32x32x16_bf16, the scores, the softmax statistics and
 * the context accumulate in fp32; inputs and output stay fp32 in HBM. */
int sbk_rope_attention_bf16(const float* qkv, const float* cosines, const float* sines, const int32_t* key_len,
                            float* out, int B, int T, int H, int Dh, int table_rows, float scale, int chunk_size,
                            int left_chunks, sbk_stream_t stream);

/* ---- a13: middle of ConvolutionModule (Conformer.py:315-330): GLU over channels of the
 * pointwise-conv output followed by the depthwise Conv1d (kernel ksize, zero padding
 * (ksize-1)/2, groups = d) + bias.   h [B,T,2d] -> y [B,T,d];  w [d,ksize]; bias [d].
 * chunk_size > 0: Dynamic Chunk Convolution (Conformer.py:190-313) -- an output frame of chunk c = t / chunk_size sees
 * its past across chunk borders but zeros instead of every frame >= (c + 1) * chunk_size. */
int sbk_glu_dwconv_f32(const float* h, const float* w, const float* bias, float* y, int B, int T, int d, int ksize,
                       int chunk_size, sbk_stream_t stream);

/* log_softmax(x / temperature) * weight over the last dimension, x [rows,V] (seq2seq.py:1933). */
int sbk_log_softmax_f32(const float* x, float* out, int rows, int V, float temperature, float weight,
                        sbk_stream_t stream);

/* ---- a15-a21: the searchers ----------------------------------------------
 * Weights of TransformerASR's decoder (Transformer.py:659-963) + embedding + seq_lin, by the
 * reference's state_dict names (all device pointers, fp32):
 *   layers[l]: norm{1,2,3}.norm.{weight,bias}; self_attn.att / multihead_attn.att
 *              {in_proj_weight [3d,d], in_proj_bias [3d], out_proj.weight [d,d], out_proj.bias};
 *              pos_ffn.ffn.{0,3}.{weight,bias}                                              */
typedef struct {
  const float *ln1_g, *ln1_b, *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;
  const float *ln2_g, *ln2_b, *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b;
  const float *ln3_g, *ln3_b, *ff1_w, *ff1_b, *ff2_w, *ff2_b;
  /* optional (NULL = unused): the three LayerNorm-fed projections with gamma/beta folded in
   * (Wf = W*gamma[k], bf = b + W.beta): self-attention in_proj, cross-attention q rows, ffn.0 */
  const float *sa_in_wf, *sa_in_bf, *ca_q_wf, *ca_q_bf, *ff1_wf, *ff1_bf;
  /* optional (NULL = unused; ABI 6): the key / value rows [2d,d] of ca_in_w as sbk_split_bf16x3 writes them -- the
   * projection of the encoder memory (once per utterance, B*T rows) then runs as sbk_gemm_nt_f32x3 */
  const uint16_t* ca_kv_w3;
  /* optional (NULL = unused; ABI 8): the panel images (sbk_split_x3p) of sa_in_w [3d,d], sa_out_w, the q rows of ca_in_w
   * [d,d], ca_out_w, ff1_w [d_ffn,d], ff2_w [d,d_ffn] -- a step with enough hypothesis rows (~200 on) then runs these
   * projections as sbk_gemm_nt_x3r (fp32 result on the bf16 matrix pipe; LayerNorm as its own launch) */
  const uint16_t *sa_in_wp, *sa_out_wp, *ca_q_wp, *ca_out_wp, *ff1_wp, *ff2_wp;
  /* optional (NULL = unused; ABI 9): the panel images of the FOLDED matrices sa_in_wf, ca_q_wf, ff1_wf -- with them (and the
   * folded biases above) a routed step runs norm1 / norm2 / norm3 inside the projection they feed (sbk_gemm_ln_nt_x3r) */
  const uint16_t *sa_in_wfp, *ca_q_wfp, *ff1_wfp;
} sbk_decoder_layer;

typedef struct {
  const sbk_decoder_layer* layers; /* HOST array of n_layers entries */
  const float* emb;                /* custom_tgt_module.layers.0.emb.Embedding.weight [V,d] */
  const float* pe;                 /* positional_encoding_decoder.pe [max_len,d] */
  const float *final_ln_g, *final_ln_b; /* decoder.norm.norm */
  const float *seq_w, *seq_b;      /* seq_lin.w [V,d],[V] (may be NULL for sbk_decoder_prefix_f32) */
  const float *seq_wf, *seq_bf;    /* optional: seq_lin with decoder.norm folded in (see sbk_decoder_layer) */
  int32_t d_model, nhead, d_ffn, n_layers, vocab, max_len, ffn_act;
  float ln_eps;
  float emb_scale; /* multiplier of the token embedding: 0 = sqrt(d_model) (NormalizedEmbedding, nnet/embedding.py);
                      1 for the Whisper decoder, whose `pe` is its learned embed_positions table */
  const uint16_t* seq_w3; /* optional (NULL = unused; ABI 6): seq_w as sbk_split_bf16x3 writes it -- the vocabulary
                             projection of a step with >= 1 024 hypothesis rows then runs as sbk_gemm_nt_f32x3 */
  const uint16_t* seq_wp; /* optional (ABI 8): seq_w's panel image (sbk_split_x3p): the vocabulary projection as sbk_gemm_nt_x3r */
  const uint16_t* seq_wfp; /* optional (ABI 9): seq_wf's panel image: decoder.norm inside the vocabulary projection (sbk_gemm_ln_nt_x3r) */
} sbk_decoder_weights;

/* ---- ABI 8: fp8 (OCP e4m3) activations AND weights on the 2 x-rate fp8 matrix instruction (configs[4]: "fp8 MFMA on CDNA4";
 * the Whisper encoder layer, integrations/huggingface/whisper.py:318-353, under run_opts precision "fp8").  Operands are
 * byte matrices with ONE fp32 scale per row (value = scale[row] * e4m3):
 *   sbk_quant_rows_fp8:   x [rows, d] fp32 (row stride ldx) -> q, scale[r] = max |x[r,:]| / 448 (weights: per output channel,
 *                         once per matrix);
 *   sbk_layernorm_fp8o:   act(LayerNorm(x)) written in that form (the next contraction's A operand, no fp32 round trip);
 *   sbk_gemm_nt_fp8a:     C = residual + alpha * act(a_scale[m] w_scale[n] (A8 . W8^T) + bias), fp32 accumulation on
 *                         v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales; K % 128 == 0; a_scale / w_scale may be
 *                         NULL (= 1); outputs: C fp32 and / or Cb bf16 and / or C8 = e4m3(result / c8_scale) with a FIXED
 *                         scale (the hidden layer of a feed-forward pair, whose row maxima are unknown before the last
 *                         column tile: with c8_scale = 1 it is the A operand of the next call with a_scale = NULL).
 * Not a parity path: e4m3 keeps 3 significand bits (tolerances in tests/test_whisper.py). */
int sbk_quant_rows_fp8(const float* x, int ldx, uint8_t* q, float* scale, int rows, int d, sbk_stream_t stream);
/* the same from bf16 rows (the bf16 attention kernel's context -> the out-projection's fp8 operand); ldx in elements */
int sbk_quant_rows_bf16_fp8(const uint16_t* xb, int ldx, uint8_t* q, float* scale, int rows, int d, sbk_stream_t stream);
int sbk_layernorm_fp8o(const float* x, const float* gamma, const float* beta, uint8_t* q, float* scale, int rows, int d,
                       float eps, int act, sbk_stream_t stream);
int sbk_gemm_nt_fp8a(const uint8_t* A8, int lda, const float* a_scale, const uint8_t* W8, int ldw, const float* w_scale,
                     const float* bias, const float* residual, int ldr, float* C, int ldc, uint16_t* Cb, int ldcb, uint8_t* C8,
                     int ldc8, float c8_scale, int M, int N, int K, int act, float alpha, sbk_stream_t stream);

/* ---- a20: TransformerLM (lobes/models/transformer/TransformerLM.py:22-187; encoder-only, regularMHA,
 * fixed_abs_sine positions, no embedding_proj) used as a full scorer (decoders/scorer.py:413-577).
 * By the reference's state_dict names (device pointers, fp32):
 *   layers[l]: encoder.layers.l.self_att.att.{in_proj_weight [3d,d], in_proj_bias, out_proj.weight, out_proj.bias};
 *              norm1.norm / norm2.norm; pos_ffn.ffn.{0,3}.w.{weight,bias}
 *   emb = custom_src_module.emb.Embedding.weight [V,d]; pe = positional_encoding.pe [max_len,d];
 *   final_ln = encoder.norm.norm; out0 / out_ln / out2 = output_proj.layers.{0,1,2}              */
typedef struct {
  const float *in_w, *in_b, *out_w, *out_b, *ln1_g, *ln1_b;
  const float *ff1_w, *ff1_b, *ff2_w, *ff2_b, *ln2_g, *ln2_b;
} sbk_lm_layer;

typedef struct {
  const sbk_lm_layer* layers; /* HOST array of n_layers entries */
  const float *emb, *pe, *final_ln_g, *final_ln_b;
  const float *out0_w, *out0_b, *out_ln_g, *out_ln_b, *out2_w, *out2_b;
  int32_t d_model, nhead, d_ffn, n_layers, vocab, max_len, ffn_act;
  int32_t normalize_before; /* Transformer.py:452-480: pre-norm (1) or post-norm (0) layers */
  int32_t pad_idx;          /* keys whose token equals pad_idx are masked (TransformerLM.py:165-187; 0) */
  float ln_eps;             /* 1e-6 everywhere in TransformerLM */
} sbk_lm_weights;

/* TransformerLM.forward (TransformerLM.py:116-158) through the KV-cached step:
 * tokens [n,L] int32 -> logits [n,L,V]. */
size_t sbk_lm_prefix_workspace_bytes(const sbk_lm_weights* LM, int n, int L);
int sbk_lm_prefix_f32(const sbk_lm_weights* LM, const int32_t* tokens, void* workspace, size_t workspace_bytes,
                      float* logits, int n, int L, sbk_stream_t stream);

/* S2SBeamSearcher options (seq2seq.py:752-768) after the host resolved ratios to step counts
 * (min/max_steps = int(T * ratio), :1336-1338) and scorer weights (ctc_weight; attn weight is
 * 1 - ctc_weight when a CTC scorer is present, :803-804). */
typedef struct {
  int32_t bos, eos, blank, beam, min_steps, max_steps;
  int32_t length_normalization, using_eos_threshold, check_every;
  int32_t overlap_ctc; /* bit 0: the survivors' CTC state update runs on a library-owned helper stream beside
                          the next decoder step; bit 1: so does the full-vocabulary CTC score pass (lowest
                          single-batch latency; costs throughput when the caller already keeps several
                          batches in flight on different streams); 0: everything on `stream` */
  float ctc_weight, temperature, eos_threshold, minus_inf;
  /* optional TransformerLM full scorer, scored BEFORE the CTC scorer like the recipe's
   * full_scorers=[transformerlm, ctc] (scorer.py:1246-1253): log_probs += lm_weight *
   * log_softmax(LM(prefix) / lm_temperature).  lm = NULL or lm_weight = 0: no LM. */
  float lm_weight, lm_temperature;
  const sbk_lm_weights* lm;
  int32_t topk; /* 0 / 1: the best hypothesis per utterance (last token stripped); > 1: return_topk
                   (seq2seq.py:757-760,1712) -- the outputs hold `topk` rows per utterance in descending
                   score order and keep their last token, like the reference's padded topk_hyps */
  /* Grouped search (optional, NULL = the scalar min_steps / max_steps for everybody): DEVICE int32 [B] arrays with
   * every utterance's own min / max decode steps.  Several independent batches of the reference -- each with its own
   * padded T and hence its own int(T * ratio) step limits (seq2seq.py:1336-1338) -- can then share ONE search: their
   * rows are stacked (enc zero-padded to the longest T; frames beyond enc_len are never read), max_steps is the
   * largest limit, and an utterance simply stops taking part after its own last step.  Results per utterance are
   * those of the separate searches; what changes is that the decoder GEMMs see the rows of all batches at once. */
  const int32_t* utt_min_steps;
  const int32_t* utt_max_steps;
  int32_t graph_mode; /* 0: one launch list per step; 1: the step counter lives in device memory and two
                         consecutive steps are captured once into a hipGraph and replayed (single-stream
                         latency; ignored with overlap_ctc, profiling or T > 900); 2: device-side counter
                         with plain launches (what 1 falls back to) */
  int32_t ctc_candidates; /* 0: CTC is a FULL scorer (every token scored, blank column blocked: scorer.py:1280-1285);
                             k > 0: CTC is a PARTIAL scorer (ScorerBuilder(partial_scorers=[ctc]), :1287-1300): only the
                             k = int(beam * scorer_beam_scale) best tokens of each hypothesis -- ranked after the eos
                             rules and the full scorers -- and <eos> get a CTC score, the rest get minus_inf */
  /* ---- S2SWhisperBeamSearcher (seq2seq.py:1937-2206) on the same search (ABI 4).  All optional (zero = off):
   * prompt [B, prompt_len] DEVICE int32: the initial tokens of every utterance (prefix / prompt / sot / language /
   *   task, :2069-2102; the per-utterance language token allowed, :2113-2119).  Positions 0 .. prompt_len-2 only fill
   *   the KV cache of every hypothesis (reset_mem), the last prompt token is the first decoder input (= bos) and
   *   search step s runs at decoder position prompt_len-1+s; prompt_len-1+max_steps <= max_len.  Not with an LM scorer,
   *   graph_mode or the grouped search.
   * logit_bias / first_bias [V] DEVICE: additive 0 / -inf masks on the logits of every step / of step 0 only
   *   (suppress_tokens :2185-2187, suppress_blank :2176-2183).
   * temperature_post: 1 = log_softmax(logits) / temperature (:2189-2192) instead of log_softmax(logits / temperature).
   * probe: out_probe [B] DEVICE = softmax(logits at prompt position probe_pos)[probe_token] -- no_speech_probs
   *   (:2161-2170); out_probe NULL = off. */
  int32_t ctc_window_size; /* CTCScorer(ctc_window_size) (scorer.py:183-187): > 0 = the CTC prefix scores of a step are
                              computed over the frames [min peak - w, max peak + w) only, the peaks being the arg-max over
                              the decoded positions of the last decoder layer's head-averaged cross-attention, min / max
                              taken over the whole batch (ctc.py:189-200, as the reference computes them for a
                              transformer decoder).  Not with the grouped search, graph_mode or overlap_ctc. */
  const int32_t* prompt;
  int32_t prompt_len;
  int32_t temperature_post;
  const float* logit_bias;
  const float* first_bias;
  int32_t probe_pos, probe_token;
  float* out_probe;
  const uint16_t* ctc_w3; /* optional (NULL = unused; ABI 6): ctc_w as sbk_split_bf16x3 writes it (the CTC head over
                             the B*T encoder frames, scorer.py:239-255, as sbk_gemm_nt_f32x3) */
} sbk_search_config;

/* ---- the CTC prefix scorer as a per-step API (ABI 11) -----------------------------------------------------------------------
 * replaces: decoders/scorer.py:108-255 CTCScorer.reset_mem / score / permute_mem = decoders/ctc.py:79-295
 * CTCPrefixScore.__init__ / forward_step / permute_mem -- what a searcher written against the reference's ScorerBuilder
 * (scorer.py:1221-1315) calls once per decoding step.  (sbk_beam_search_f32 drives the same kernels itself.)  The caller owns the
 * state: `x` [B,T,V] and a workspace of sbk_ctc_scorer_workspace_bytes(B, T, V, beam) bytes, both 16-byte aligned device memory.
 *   reset   x = log_softmax(ctc_fc(enc)) on entry; converted IN PLACE to the masked linear posteriors the scorer reads from then
 *           on (frames past enc_len[b]: 1 for token 0, else 0 -- the reference's mask); initial state for `beam` hypotheses per
 *           utterance (ctc.py:103-124).
 *   score   step = number of tokens decoded so far (0 at <bos>); inp_tokens [B*beam] = the last token of every hypothesis;
 *           scores [B*beam, V] = psi - psi_prev (ctc.py:262): psi[eos] = the prefix's own end probability, psi[blank] = -1e20.
 *           ctc_window_size > 0: attn_window = device {min, max} of the attention peaks of this step (ctc.py:189-200).
 *   permute after the beam update of the same step: hypothesis row n of the NEW beam extends row parent[n] (0 .. B*beam-1) of
 *           the old one by token[n]; parent_last_tok = the inp_tokens passed to score().  (ctc.py:243-295 selects the same
 *           states by flat candidate indices.) */
size_t sbk_ctc_scorer_workspace_bytes(int B, int T, int V, int beam);
int sbk_ctc_scorer_reset_f32(float* x, const int32_t* enc_len, void* workspace, size_t workspace_bytes, int B, int T, int V,
                             int beam, int blank, sbk_stream_t stream);
int sbk_ctc_scorer_score_f32(const float* x, const int32_t* enc_len, void* workspace, size_t workspace_bytes,
                             const int32_t* inp_tokens, int step, const int32_t* attn_window, int ctc_window_size, float* scores,
                             int B, int T, int V, int beam, int blank, int eos, sbk_stream_t stream);
int sbk_ctc_scorer_permute_f32(const float* x, void* workspace, size_t workspace_bytes, const int32_t* parent,
                               const int32_t* token, const int32_t* parent_last_tok, int step, const int32_t* attn_window,
                               int ctc_window_size, int B, int T, int V, int beam, int blank, sbk_stream_t stream);

/* S2STransformerBeamSearcher.forward (seq2seq.py:1632-1723, :1853-1934) with an optional full
 * CTC scorer (scorer.py:108-255,1221-1315; ctc.py:26-295).
 *   enc [B,T,d], enc_len [B] = round(T * wav_len); ctc_w [V,d], ctc_b [V] (NULL when ctc_weight = 0)
 *   out_tokens [B*topk,max_steps] (topk = 1: best hypothesis, EOS stripped, zero padded), out_len [B*topk]
 *   (= token count - 1), out_score [B*topk], out_logp [B*topk,max_steps]; out_max_len [1] (device, may be NULL): length of the
 *   longest finished hypothesis in the batch = pad width the reference divides lengths by (:1461)
 *   host_flag: non-NULL = the stop rule ("every utterance holds `beam` finished hypotheses") is polled every
 *              cfg->check_every steps; NULL => run max_steps.  The poll is asynchronous since ABI 4: a 4-byte
 *              device->host copy into a pinned ring of the library plus an event, read back two polls later -- the
 *              stream is never drained, the host runs at most ~3 poll intervals ahead of the device, and a search
 *              runs at most that far past its stop point (which cannot change its result).  The pointed-to word
 *              itself is no longer written.
 *   steps_run: HOST int32 out. */
size_t sbk_beam_search_workspace_bytes(const sbk_decoder_weights* W, const sbk_search_config* cfg, int B, int T);
int sbk_beam_search_f32(const sbk_decoder_weights* W, const sbk_search_config* cfg, const float* enc,
                        const int32_t* enc_len, const float* ctc_w, const float* ctc_b, void* workspace,
                        size_t workspace_bytes, int32_t* out_tokens, int32_t* out_len, float* out_score,
                        float* out_logp, int32_t* out_max_len, int32_t* out_longest, int32_t* host_flag,
                        int32_t* steps_run, int B, int T, sbk_stream_t stream);

/* S2STransformerGreedySearcher.forward (seq2seq.py:176-367, temperature 0): per-step arg-max.
 *   out_tokens [B,max_steps] (EOS-latched), out_scores [B,max_steps] (log-prob of the arg-max, 0 after the end) */
size_t sbk_greedy_search_workspace_bytes(const sbk_decoder_weights* W, int B, int T, int max_steps);
int sbk_greedy_search_f32(const sbk_decoder_weights* W, const float* enc, const int32_t* enc_len, void* workspace,
                          size_t workspace_bytes, int32_t* out_tokens, float* out_scores, int32_t* host_flag,
                          int32_t* steps_run, int B, int T, int min_steps, int max_steps, int bos, int eos,
                          int check_every, sbk_stream_t stream);

/* S2SWhisperGreedySearcher.forward (seq2seq.py:421-636 on S2SGreedySearcher.forward :176-327, temperature 0) for a
 * decoder described by sbk_decoder_weights (the Whisper decoder: emb_scale 1, learned positions, GELU, seq_w = the
 * tied token embedding, seq_b zeros).  prompt [B,P] int32 (device): the initial tokens (:542-573, per-utterance
 * language token allowed); positions 0..P-2 only fill the KV cache, the last prompt token is the first decoder input.
 * Step k = 0.. samples arg-max of  logits + logit_bias (+ first_bias at k = 0)  -- [V] additive masks of 0 / -inf
 * (suppress_tokens :627-629, suppress_blank :619-625), NULL = none -- with the EOS latch, scores and layout of
 * sbk_greedy_search_f32: out_tokens / out_scores [B,max_new].  out_probe [B] (may be NULL): softmax(logits at prompt
 * position probe_pos)[probe_token] -- no_speech_probs (:604-612).  enc_len: pass T for every utterance (the Whisper
 * decoder attends to all encoder frames). */
size_t sbk_prompted_greedy_search_workspace_bytes(const sbk_decoder_weights* W, int B, int T, int P, int max_new);
int sbk_prompted_greedy_search_f32(const sbk_decoder_weights* W, const float* enc, const int32_t* enc_len,
                                   const int32_t* prompt, int P, const float* logit_bias, const float* first_bias,
                                   void* workspace, size_t workspace_bytes, int32_t* out_tokens, float* out_scores,
                                   int probe_pos, int probe_token, float* out_probe, int32_t* host_flag,
                                   int32_t* steps_run, int B, int T, int max_new, int eos, int check_every,
                                   sbk_stream_t stream);

/* TransformerASR.decode (TransformerASR.py:426-473) through the KV-cached step:
 * tokens [n,L] int32, enc [n,T,d], enc_len [n] -> pred [n,L,d] (decoder.norm output). */
size_t sbk_decoder_prefix_workspace_bytes(const sbk_decoder_weights* W, int n, int T, int L);
int sbk_decoder_prefix_f32(const sbk_decoder_weights* W, const int32_t* tokens, const float* enc,
                           const int32_t* enc_len, void* workspace, size_t workspace_bytes, float* pred, int n, int T,
                           int L, sbk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SBK_H_ */
