// Fused STFT -> power -> mel -> dB (-> top_db floor -> global norm)   (sbk_fbank_f32)
//
// Roofline: HBM.  Algorithmic traffic per audio-second (16 kHz, 80 mels, hop
// 160): 64 000 B of samples in + 32 032 B of features out (+ the same 32 KB
// re-read/re-written by the floor pass, L2-resident for LibriSpeech-length
// utterances).  Nothing else touches HBM: frames are cut straight out of the
// waveform (each sample is fetched ~n_fft/hop times, all but the first from
// L1/L2), the FFT runs entirely in LDS, and the twiddle table and the
// compacted mel filterbank are LDS-staged once per workgroup.
//
// Kernel 1: one wavefront per frame (4 frames per 256-thread workgroup).
//   Stockham autosort FFT, mixed radix {2,3,4,5} so that both recipe sizes
//   (n_fft = 512 = 4^4*2 and n_fft = 400 = 5*5*4*4) stay in LDS; power
//   spectrum; triangular filters as contiguous bin runs (CSR by filter);
//   10*log10(max(.,amin)); per-tile max for the per-utterance floor.
// Kernel 2: floor at (utterance max - top_db) and optional (x-mean)/std.
#include "common.h"

namespace {

constexpr int kMaxRadix = 12;
struct Radices {
  int n;
  int r[kMaxRadix];
};

struct FbankArgs {
  const float* wav;
  const float* window;
  const float* twiddle;
  const float* mel_w;
  const int32_t* mel_ptr;
  const int32_t* mel_bin;
  float* out;
  float* tile_max;
  int B, N, T, n_fft, hop, n_mels, nnz, ntiles;
  float amin;
  float* spec;  // optional [B,T,n_fft/2+1,2] complex STFT output (STFT.forward); when set, the mel stage is skipped
  int reflect;     // 1: torch.stft's default reflect padding of the centred frames (Whisper); 0: zero padding (Fbank)
  float log_mult;  // 10 (dB, Fbank) or 1 (plain log10, Whisper)
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ void __launch_bounds__(256) fbank_frames_kernel(FbankArgs a, Radices rad) {
  SBK_DYN_LDS(float, lds);
  const int n_fft = a.n_fft, n_stft = a.n_fft / 2 + 1;
  float2* tw = reinterpret_cast<float2*>(lds);                    // [n_fft]
  float2* bufs = tw + n_fft;                                      // [4 waves][2][n_fft]
  float* pw = reinterpret_cast<float*>(bufs + 4 * 2 * n_fft);     // [4 waves][n_stft]
  float* melw = pw + 4 * n_stft;                                  // [nnz]
  float* wmax = melw + a.nnz;                                     // [4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int t = blockIdx.x * 4 + wave;

  for (int i = tid; i < n_fft; i += 256) tw[i] = make_float2(a.twiddle[2 * i], a.twiddle[2 * i + 1]);
  for (int i = tid; i < a.nnz; i += 256) melw[i] = a.mel_w[i];

  float2* cur = bufs + (wave * 2 + 0) * n_fft;
  float2* nxt = bufs + (wave * 2 + 1) * n_fft;
  {  // cut the (centre-padded) frame out of the waveform and window it
    const float* w = a.wav + (size_t)b * a.N;
    const long base = (long)t * a.hop - n_fft / 2;
    for (int n = lane; n < n_fft; n += 64) {
      long idx = base + n;
      if (a.reflect) {  // mirror without repeating the edge sample: x[-k] = x[k], x[N-1+k] = x[N-1-k]
        if (idx < 0) idx = -idx;
        if (idx >= a.N) idx = 2 * ((long)a.N - 1) - idx;
      }
      const float x = (t < a.T && idx >= 0 && idx < a.N) ? w[idx] : 0.0f;
      cur[n] = make_float2(x * a.window[n], 0.0f);
    }
  }
  __syncthreads();

  int Ns = 1;
  for (int p = 0; p < rad.n; ++p) {
    const int R = rad.r[p];
    const int nb = n_fft / R;          // butterflies in this pass
    const int tstep = n_fft / (Ns * R);  // twiddle stride of this pass
    const int rstep = n_fft / R;       // stride of the R-point DFT's own roots
    for (int j = lane; j < nb; j += 64) {
      const int k = j % Ns;
      float2 v[5];
      for (int q = 0; q < R; ++q) v[q] = cmul(cur[j + q * nb], tw[k * q * tstep]);
      const int dst = (j / Ns) * Ns * R + k;
      for (int u = 0; u < R; ++u) {
        float2 acc = v[0];
        for (int q = 1; q < R; ++q) {
          const float2 c = cmul(v[q], tw[((u * q) % R) * rstep]);
          acc.x += c.x;
          acc.y += c.y;
        }
        nxt[dst + u * Ns] = acc;
      }
    }
    __syncthreads();
    float2* tmp = cur;
    cur = nxt;
    nxt = tmp;
    Ns *= R;
  }

  if (a.spec) {  // plain STFT: (re, im) per bin, no mel stage (uniform branch)
    if (t < a.T) {
      float2* dst = reinterpret_cast<float2*>(a.spec) + ((size_t)b * a.T + t) * n_stft;
      for (int f = lane; f < n_stft; f += 64) dst[f] = cur[f];
    }
    return;
  }
  float* mypw = pw + wave * n_stft;
  for (int f = lane; f < n_stft; f += 64) mypw[f] = cur[f].x * cur[f].x + cur[f].y * cur[f].y;
  __syncthreads();

  float mx = -INFINITY;
  for (int m = lane; m < a.n_mels; m += 64) {
    const int p0 = a.mel_ptr[m], p1 = a.mel_ptr[m + 1], f0 = a.mel_bin[m];
    float acc = 0.0f;
    for (int i = p0; i < p1; ++i) acc = fmaf(mypw[f0 + (i - p0)], melw[i], acc);
    // log10 in f64: correctly rounded to f32 (silence must give exactly 10*log10(amin) like the reference)
    const float db = a.log_mult * (float)log10((double)fmaxf(acc, a.amin));
    if (t < a.T) {
      a.out[((size_t)b * a.T + t) * a.n_mels + m] = db;
      mx = fmaxf(mx, db);
    }
  }
  mx = sbk::wave_max(mx);
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  if (tid == 0) a.tile_max[(size_t)b * a.ntiles + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

__global__ void __launch_bounds__(256) fbank_floor_norm_kernel(float* __restrict__ x, const float* __restrict__ tile_max,
                                                               int ntiles, long per_utt, int n_mels, float top_db,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ sd, float eps) {
  __shared__ float red[4];
  const int b = blockIdx.y, tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < ntiles; i += 256) mx = fmaxf(mx, tile_max[(size_t)b * ntiles + i]);
  mx = sbk::wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  const float floor_db = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - top_db;
  float* xb = x + (size_t)b * per_utt;
  for (long i = (long)blockIdx.x * 256 + tid; i < per_utt; i += (long)gridDim.x * 256) {
    float v = fmaxf(xb[i], floor_db);
    if (mean) {
      const int m = (int)(i % n_mels);
      v = (v - mean[m]) / fmaxf(sd[m], eps);
    }
    xb[i] = v;
  }
}


// Whisper's dynamic-range step (integrations/huggingface/whisper.py:312-314, openai/whisper audio.py): floor at
// (max over the WHOLE BATCH - 8), then (x + 4) / 4; written mel-major [B, n_mels, T] as the encoder's Conv1d reads it.
// x [B,T,M] is the frames kernel's output, tile_max its [B*ntiles] per-tile maxima.
__global__ void __launch_bounds__(256) whisper_floor_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ tile_max, int n_tile_max,
                                                            float* __restrict__ out, int T, int M) {
  __shared__ float red[4];
  __shared__ float tile[32][33];
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < n_tile_max; i += 256) mx = fmaxf(mx, tile_max[i]);
  mx = sbk::wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  const float floor_v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - 8.0f;
  // 32 x 32 (frame, mel) tile transposed through LDS: coalesced reads along mel, coalesced writes along time
  const int b = blockIdx.z, t0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int c = tid & 31, r0 = tid >> 5;
  for (int r = r0; r < 32; r += 8) {
    const int t = t0 + r, m = m0 + c;
    tile[r][c] = (t < T && m < M) ? x[((size_t)b * T + t) * M + m] : 0.0f;
  }
  __syncthreads();
  for (int r = r0; r < 32; r += 8) {
    const int m = m0 + r, t = t0 + c;
    if (m < M && t < T) out[((size_t)b * M + m) * T + t] = (fmaxf(tile[c][r], floor_v) + 4.0f) / 4.0f;
  }
}
}  // namespace

extern "C" int sbk_fbank_f32(const float* wav, const float* window, const float* twiddle, const int32_t* radices,
                             int n_radix, const float* mel_w, const int32_t* mel_ptr, const int32_t* mel_bin,
                             float* out, float* tile_max, int B, int N, int n_fft, int hop, int n_mels, int nnz,
                             float amin, float top_db, const float* norm_mean, const float* norm_std, float norm_eps,
                             sbk_stream_t stream) {
  if (B == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(wav && window && twiddle && radices && mel_w && mel_ptr && mel_bin && out && tile_max,
              "fbank: null operand");
  SBK_REQUIRE(B >= 0 && N >= 0 && n_fft >= 2 && hop > 0 && n_mels > 0 && nnz >= 0, "fbank: bad shape");
  SBK_REQUIRE(n_radix > 0 && n_radix <= kMaxRadix, "fbank: %d FFT passes", n_radix);
  SBK_REQUIRE((norm_mean == nullptr) == (norm_std == nullptr), "fbank: need both mean and std or neither");
  Radices rad;
  rad.n = n_radix;
  long prod = 1;
  for (int i = 0; i < n_radix; ++i) {
    rad.r[i] = radices[i];
    SBK_REQUIRE(radices[i] >= 2 && radices[i] <= 5, "fbank: radix %d unsupported (n_fft must factor into 2,3,4,5)",
                radices[i]);
    prod *= radices[i];
  }
  SBK_REQUIRE(prod == n_fft, "fbank: radices do not multiply to n_fft=%d", n_fft);
  if (B == 0) return 0;
  const int T = 1 + N / hop;
  const int ntiles = sbk::cdiv(T, 4);
  const int n_stft = n_fft / 2 + 1;
  const size_t lds = (size_t)n_fft * 8 + (size_t)4 * 2 * n_fft * 8 + (size_t)4 * n_stft * 4 + (size_t)nnz * 4 + 16;
  SBK_REQUIRE(lds <= 160 * 1024, "fbank: n_fft=%d needs %zu B of LDS", n_fft, lds);
  FbankArgs a{wav, window, twiddle, mel_w, mel_ptr, mel_bin, out, tile_max, B, N, T, n_fft, hop, n_mels, nnz, ntiles, amin, nullptr,
              0, 10.0f};
  hipStream_t st = sbk::as_stream(stream);
  sbk::ProfScope prof("fbank", 5.0 * n_fft * 9.0 * B * T, 4.0 * ((double)B * N + 3.0 * B * T * n_mels), st);
  SBK_LAUNCH(fbank_frames_kernel, dim3(ntiles, B), dim3(256), lds, st, a, rad);
  int rc = sbk::launch_status("sbk_fbank_f32/frames");
  if (rc) return rc;
  const long per_utt = (long)T * n_mels;
  const int gx = (int)((per_utt + 2047) / 2048 < 1 ? 1 : (per_utt + 2047) / 2048);
  SBK_LAUNCH(fbank_floor_norm_kernel, dim3(gx, B), dim3(256), 0, st, out, tile_max, ntiles, per_utt, n_mels, top_db,
             norm_mean, norm_std, norm_eps);
  return sbk::launch_status("sbk_fbank_f32/floor");
}


// STFT.forward (processing/features.py:141-188): [B,N] -> [B,T,n_fft/2+1,2] (re, im).
extern "C" int sbk_stft_f32(const float* wav, const float* window, const float* twiddle, const int32_t* radices,
                            int n_radix, float* spec, int B, int N, int n_fft, int hop, sbk_stream_t stream) {
  if (B == 0) return 0;  // empty batch: nothing to launch, the data pointers may be NULL
  SBK_REQUIRE(wav && window && twiddle && radices && spec, "stft: null operand");
  SBK_REQUIRE(B >= 0 && N >= 0 && n_fft >= 2 && hop > 0, "stft: bad shape");
  SBK_REQUIRE(n_radix > 0 && n_radix <= kMaxRadix, "stft: %d FFT passes", n_radix);
  Radices rad;
  rad.n = n_radix;
  long prod = 1;
  for (int i = 0; i < n_radix; ++i) {
    rad.r[i] = radices[i];
    SBK_REQUIRE(radices[i] >= 2 && radices[i] <= 5, "stft: radix %d unsupported", radices[i]);
    prod *= radices[i];
  }
  SBK_REQUIRE(prod == n_fft, "stft: radices do not multiply to n_fft=%d", n_fft);
  if (B == 0) return 0;
  const int T = 1 + N / hop, ntiles = sbk::cdiv(T, 4), n_stft = n_fft / 2 + 1;
  const size_t lds = (size_t)n_fft * 8 + (size_t)4 * 2 * n_fft * 8 + (size_t)4 * n_stft * 4 + 16;
  SBK_REQUIRE(lds <= 160 * 1024, "stft: n_fft=%d needs %zu B of LDS", n_fft, lds);
  FbankArgs a{wav, window, twiddle, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, T, n_fft, hop, 0, 0, ntiles, 0.0f, spec};
  SBK_LAUNCH(fbank_frames_kernel, dim3(ntiles, B), dim3(256), lds, sbk::as_stream(stream), a, rad);
  return sbk::launch_status("sbk_stft_f32");
}

namespace {
// spectral_magnitude (processing/features.py:341-378): (re^2 + im^2) ^ power, optional log.
__global__ void __launch_bounds__(256) spectral_magnitude_kernel(const float2* __restrict__ x, float* __restrict__ y,
                                                                 long n, float power, int take_log, float eps) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = x[i].x * x[i].x + x[i].y * x[i].y;
    if (power < 1.0f) v = v + eps;  // the reference adds eps before a fractional power
    if (power != 1.0f) v = powf(v, power);
    y[i] = take_log ? logf(v + eps) : v;
  }
}

// _amplitude_to_DB (processing/features.py:736-759): 10*log10(max(x,amin)) then the per-utterance floor.
__global__ void __launch_bounds__(256) to_db_kernel(float* __restrict__ x, float* __restrict__ tile_max, long per_utt,
                                                    float mult, float amin, float db_offset) {
  __shared__ float red[4];
  const int b = blockIdx.y, tid = threadIdx.x;
  float* xb = x + (size_t)b * per_utt;
  float mx = -INFINITY;
  for (long i = (long)blockIdx.x * 256 + tid; i < per_utt; i += (long)gridDim.x * 256) {
    const float v = mult * (float)log10((double)fmaxf(xb[i], amin)) - db_offset;
    xb[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = sbk::wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) tile_max[(size_t)b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// a1: PCM16 frames -> float32 mono, the soundfile convention the reference loads with (sample / 32768,
// dataio/audio_io.py:141-209) followed by AudioNormalizer's channel mean (dataio/preprocess.py:76-84).
// Streaming: 2*channels bytes in, 4 bytes out per frame; 8 frames per thread.
__global__ void __launch_bounds__(256) pcm16_to_f32_kernel(const int16_t* __restrict__ pcm, float* __restrict__ out,
                                                           long frames, int channels) {
  const float inv = 1.0f / 32768.0f;
  for (long f0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8; f0 < frames; f0 += (long)gridDim.x * 256 * 8) {
    if (channels == 1 && f0 + 8 <= frames && (reinterpret_cast<uintptr_t>(pcm + f0) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out + f0) & 15) == 0) {
      const int4 raw = *reinterpret_cast<const int4*>(pcm + f0);  // 8 samples in one 16-byte load
      const int w[4] = {raw.x, raw.y, raw.z, raw.w};
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] = (float)(int16_t)(w[k] & 0xFFFF) * inv;
        v[2 * k + 1] = (float)(int16_t)((unsigned)w[k] >> 16) * inv;
      }
      *reinterpret_cast<float4*>(out + f0) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out + f0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      for (long f = f0; f < f0 + 8 && f < frames; ++f) {
        float acc = 0.0f;
        for (int c = 0; c < channels; ++c) acc += (float)pcm[f * channels + c] * inv;  // exact: |pcm| < 2^15
        out[f] = channels == 1 ? acc : acc / (float)channels;  // torch.mean over the channel axis
      }
    }
  }
}
}  // namespace

extern "C" int sbk_pcm16_to_f32(const int16_t* pcm, float* out, long frames, int channels, sbk_stream_t stream) {
  if (frames == 0) return 0;
  SBK_REQUIRE(pcm && out && frames > 0 && channels >= 1, "pcm16_to_f32: bad arguments");
  const long groups = (frames + 2047) / 2048;
  const int blocks = (int)(groups < 8192 ? groups : 8192);
  sbk::ProfScope prof("pcm16_to_f32", 0.0, (2.0 * channels + 4.0) * (double)frames, sbk::as_stream(stream));
  SBK_LAUNCH(pcm16_to_f32_kernel, dim3(blocks), dim3(256), 0, sbk::as_stream(stream), pcm, out, frames, channels);
  return sbk::launch_status("sbk_pcm16_to_f32");
}

extern "C" int sbk_spectral_magnitude_f32(const float* stft, float* out, long n, float power, int take_log, float eps,
                                          sbk_stream_t stream) {
  SBK_REQUIRE(stft && out && n >= 0, "spectral_magnitude: bad arguments");
  if (n == 0) return 0;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  SBK_LAUNCH(spectral_magnitude_kernel, dim3(blocks), dim3(256), 0, sbk::as_stream(stream),
             reinterpret_cast<const float2*>(stft), out, n, power, take_log, eps);
  return sbk::launch_status("sbk_spectral_magnitude_f32");
}

// In place: x [B, per_utt] linear filterbank energies -> dB with the per-utterance (max - top_db) floor.
// tile_max: workspace [B, 64].
extern "C" int sbk_amplitude_to_db_f32(float* x, float* tile_max, int B, long per_utt, float multiplier, float amin,
                                       float db_offset, float top_db, sbk_stream_t stream) {
  SBK_REQUIRE(x && tile_max && B >= 0 && per_utt > 0, "amplitude_to_db: bad arguments");
  if (B == 0) return 0;
  hipStream_t st = sbk::as_stream(stream);
  const int nt = 64;
  SBK_LAUNCH(to_db_kernel, dim3(nt, B), dim3(256), 0, st, x, tile_max, per_utt, multiplier, amin, db_offset);
  int rc = sbk::launch_status("sbk_amplitude_to_db_f32/db");
  if (rc) return rc;
  const int gx = (int)((per_utt + 2047) / 2048 < 1 ? 1 : (per_utt + 2047) / 2048);
  SBK_LAUNCH(fbank_floor_norm_kernel, dim3(gx, B), dim3(256), 0, st, x, (const float*)tile_max, nt, per_utt, 1, top_db,
             (const float*)nullptr, (const float*)nullptr, 0.0f);
  return sbk::launch_status("sbk_amplitude_to_db_f32/floor");
}


// Whisper log-mel front-end (integrations/huggingface/whisper.py:276-316 `log_mel_spectrogram`):
//   torch.stft(audio, n_fft, hop, hann_window(n_fft), center=True, reflect padding) -> drop the last frame ->
//   |.|^2 -> mel filters -> log10(max(., 1e-10)) -> max(., batch max - 8) -> (. + 4) / 4       [B, n_mels, N/hop]
// window / twiddle / radices / CSR mel filters as for sbk_fbank_f32; tmp [B,T,n_mels] and tile_max [B*ceil(T/4)]
// are workspaces.
extern "C" int sbk_whisper_log_mel_f32(const float* wav, const float* window, const float* twiddle,
                                       const int32_t* radices, int n_radix, const float* mel_w, const int32_t* mel_ptr,
                                       const int32_t* mel_bin, float* tmp, float* tile_max, float* out, int B, int N,
                                       int n_fft, int hop, int n_mels, int nnz, sbk_stream_t stream) {
  if (B == 0) return 0;
  SBK_REQUIRE(wav && window && twiddle && radices && mel_w && mel_ptr && mel_bin && tmp && tile_max && out,
              "whisper_log_mel: null operand");
  SBK_REQUIRE(N >= n_fft / 2 + 1 && n_fft >= 2 && hop > 0 && n_mels > 0 && nnz >= 0 && N / hop >= 1,
              "whisper_log_mel: bad shape (reflect padding needs N > n_fft/2)");
  SBK_REQUIRE(n_radix > 0 && n_radix <= kMaxRadix, "whisper_log_mel: %d FFT passes", n_radix);
  Radices rad;
  rad.n = n_radix;
  long prod = 1;
  for (int i = 0; i < n_radix; ++i) {
    rad.r[i] = radices[i];
    SBK_REQUIRE(radices[i] >= 2 && radices[i] <= 5, "whisper_log_mel: radix %d unsupported", radices[i]);
    prod *= radices[i];
  }
  SBK_REQUIRE(prod == n_fft, "whisper_log_mel: radices do not multiply to n_fft=%d", n_fft);
  const int T = N / hop;  // 1 + N/hop centred frames, minus the last one (stft[..., :-1])
  const int ntiles = sbk::cdiv(T, 4);
  const int n_stft = n_fft / 2 + 1;
  const size_t lds = (size_t)n_fft * 8 + (size_t)4 * 2 * n_fft * 8 + (size_t)4 * n_stft * 4 + (size_t)nnz * 4 + 16;
  SBK_REQUIRE(lds <= 160 * 1024, "whisper_log_mel: n_fft=%d needs %zu B of LDS", n_fft, lds);
  FbankArgs a{wav, window, twiddle, mel_w, mel_ptr, mel_bin, tmp, tile_max, B, N, T, n_fft, hop, n_mels, nnz, ntiles, 1e-10f,
              nullptr, 1, 1.0f};
  hipStream_t st = sbk::as_stream(stream);
  sbk::ProfScope prof("whisper_log_mel", 5.0 * n_fft * 9.0 * B * T, 4.0 * ((double)B * N + 3.0 * B * T * n_mels), st);
  SBK_LAUNCH(fbank_frames_kernel, dim3(ntiles, B), dim3(256), lds, st, a, rad);
  int rc = sbk::launch_status("sbk_whisper_log_mel_f32/frames");
  if (rc) return rc;
  SBK_LAUNCH(whisper_floor_kernel, dim3(sbk::cdiv(T, 32), sbk::cdiv(n_mels, 32), B), dim3(256), 0, st, (const float*)tmp,
             (const float*)tile_max, B * ntiles, out, T, n_mels);
  return sbk::launch_status("sbk_whisper_log_mel_f32/floor");
}
