"""speechbrain.integrations.huggingface.whisper mirror -- the log-mel front-end of WhisperASR (BASELINE.json
configs[4]): ``pad_or_trim`` + ``log_mel_spectrogram`` (integrations/huggingface/whisper.py:276-350), on the HIP
kernels (csrc/fbank.hip, ``sbk_whisper_log_mel_f32``).

The encoder / decoder of the reference live in HuggingFace ``transformers`` (``WhisperModel``, call sites :372-374,
:417-423; third-party, pinned 4.46.3 / 4.53.2 in uv.lock); they are NOT re-implemented here yet -- only the
front-end, which IS in-tree in the reference, is.  The mel filters are those of the HF feature extractor
(``transformers.audio_utils.mel_filter_bank(..., norm="slaney", mel_scale="slaney")``), restated below and pinned
against the installed ``transformers`` in tests/test_whisper.py."""
import math

import numpy as np
import torch

from speechbrain_amd import native
from speechbrain_amd.processing.features import factor_radices

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000 samples in a 30-second chunk (whisper.py:25-28)


def _hz_to_mel_slaney(hz):
    hz = np.asarray(hz, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    logstep = 27.0 / np.log(6.4)
    mel = 3.0 * hz / 200.0
    return np.where(hz >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(hz, 1e-30) / min_log_hz) * logstep, mel)


def _mel_to_hz_slaney(mel):
    mel = np.asarray(mel, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(mel >= min_log_mel, min_log_hz * np.exp(logstep * (mel - min_log_mel)), 200.0 * mel / 3.0)


def slaney_mel_filters(n_mels=80, n_fft=N_FFT, sample_rate=SAMPLE_RATE, f_min=0.0, f_max=8000.0):
    """[n_fft/2+1, n_mels] triangular filters on the Slaney mel scale with Slaney (area) normalisation: what
    WhisperFeatureExtractor builds (and librosa.filters.mel does)."""
    n_bins = n_fft // 2 + 1
    fft_freqs = np.linspace(0, sample_rate // 2, n_bins)
    filter_freqs = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2))
    diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    fb *= (2.0 / (filter_freqs[2: n_mels + 2] - filter_freqs[:n_mels]))[None, :]
    return torch.from_numpy(fb.astype(np.float32))


class WhisperLogMel(torch.nn.Module):
    """``_get_mel`` of the reference's Whisper wrapper (whisper.py:262-316): waveforms [B, time] at 16 kHz ->
    [B, n_mels, 3000].  ``n_mels`` = 80 (up to large-v2) or 128 (large-v3)."""

    def __init__(self, n_mels=128, n_fft=N_FFT, hop_length=HOP_LENGTH, n_samples=N_SAMPLES, mel_filters=None):
        super().__init__()
        self._n_fft, self._hop_length, self._n_samples, self.n_mels = n_fft, hop_length, n_samples, n_mels
        fb = slaney_mel_filters(n_mels, n_fft) if mel_filters is None else torch.as_tensor(mel_filters, dtype=torch.float32)
        if fb.shape[0] == n_mels and fb.shape[1] != n_mels:  # the reference accepts either orientation (:176-182)
            fb = fb.t()
        w, ptr, first = [], [0], []
        for j in range(n_mels):  # CSR by filter: each triangular filter is one contiguous run of bins
            nz = torch.nonzero(fb[:, j]).flatten()
            lo, hi = (int(nz[0]), int(nz[-1]) + 1) if nz.numel() else (0, 0)
            first.append(lo)
            w.append(fb[lo:hi, j])
            ptr.append(ptr[-1] + hi - lo)
        m = torch.arange(n_fft, dtype=torch.float64) * (2.0 * math.pi / n_fft)
        self.radices = factor_radices(n_fft)
        self.register_buffer("_mel_filters", fb.t().contiguous(), persistent=False)  # [n_mels, 201] like the reference
        self.register_buffer("window", torch.hann_window(n_fft), persistent=False)
        self.register_buffer("twiddle", torch.stack([torch.cos(m), -torch.sin(m)], dim=1).float().contiguous(), persistent=False)
        self.register_buffer("mel_w", torch.cat(w).contiguous(), persistent=False)
        self.register_buffer("mel_ptr", torch.tensor(ptr, dtype=torch.int32), persistent=False)
        self.register_buffer("mel_bin", torch.tensor(first, dtype=torch.int32), persistent=False)

    def pad_or_trim(self, array, length=None, axis=-1):
        """whisper.py:318-350: zero right-pad or cut to ``length`` samples along ``axis``."""
        length = self._n_samples if length is None else length
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            pad_widths = [(0, 0)] * array.ndim
            pad_widths[axis] = (0, length - array.shape[axis])
            array = torch.nn.functional.pad(array, [p for sizes in pad_widths[::-1] for p in sizes])
        return array

    def log_mel_spectrogram(self, audio, padding: int = 0):
        if padding > 0:
            audio = torch.nn.functional.pad(audio, (0, padding))
        return native.whisper_log_mel(audio.float().contiguous(), self.window, self.twiddle, self.radices, self.mel_w,
                                      self.mel_ptr, self.mel_bin, self._n_fft, self._hop_length, self.n_mels)

    def forward(self, wav):
        return self.log_mel_spectrogram(self.pad_or_trim(wav))


# ------------------------------------------------------------------------------------------- encoder / decoder
# The reference wraps HuggingFace's WhisperModel (whisper.py:59-117; encoder call :372-374, decoder call :417-436).
# Here the same parameters (HF state_dict names, so HF / SpeechBrain checkpoints load unchanged) drive the MI355X
# kernels: the two input convolutions are GEMMs over an in-place strided window of the time-major signal, attention is
# the rotary flash kernel of the Conformer encoder without a rotation table, every Linear / LayerNorm / GELU is the fused
# GEMM / row kernel of the rest of the path.
import json  # noqa: E402
import os  # noqa: E402

from torch import nn  # noqa: E402


class _Attention(nn.Module):
    """Parameter holder with HF's names (modeling_whisper.WhisperAttention): k_proj has no bias."""

    def __init__(self, d, heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = d, heads, d // heads
        assert self.head_dim * heads == d
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)

    def stacked(self, interleave_heads):
        """(in_proj weight [3d,d], bias [3d]) of the three projections, cached until a parameter changes.
        interleave_heads: rows ordered [head][q|k|v][head_dim] (what the encoder attention kernel reads in place);
        otherwise [q|k|v][d] (torch.nn.MultiheadAttention's in_proj layout, read by the decoder kernels)."""
        ps = (self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.v_proj.weight, self.v_proj.bias)
        key = (interleave_heads,) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_stack_key", None) != key:
            with torch.no_grad():
                zero = torch.zeros_like(self.q_proj.bias)
                w = torch.stack([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])       # [3,d,d]
                b = torch.stack([self.q_proj.bias, zero, self.v_proj.bias])                          # [3,d]
                if interleave_heads:
                    H, Dh, d = self.num_heads, self.head_dim, self.embed_dim
                    w = w.view(3, H, Dh, d).permute(1, 0, 2, 3)
                    b = b.view(3, H, Dh).permute(1, 0, 2)
                self._stack = (w.reshape(3 * self.embed_dim, self.embed_dim).contiguous(), b.reshape(-1).contiguous())
            # (built by kernels on THIS thread's stream; a batch in flight on another stream may ask for it a moment later:
            # the consumer waits for the producer's event on the device, as for every derived weight image -- ADVICE r3)
            self._stack_ready = native._Ready(self.q_proj.weight.device)
            self._stack_key = key
        self._stack_ready.wait(self.q_proj.weight.device)
        return self._stack


class _EncoderLayer(nn.Module):
    def __init__(self, d, heads, ffn):
        super().__init__()
        self.self_attn = _Attention(d, heads)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, ffn)
        self.fc2 = nn.Linear(ffn, d)
        self.final_layer_norm = nn.LayerNorm(d)


class _DecoderLayer(nn.Module):
    def __init__(self, d, heads, ffn):
        super().__init__()
        self.self_attn = _Attention(d, heads)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.encoder_attn = _Attention(d, heads)
        self.encoder_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, ffn)
        self.fc2 = nn.Linear(ffn, d)
        self.final_layer_norm = nn.LayerNorm(d)



def att_ok(layers):
    """The bf16 attention kernel of the bf16-activation path is instantiated for head_dim 64."""
    return len(layers) > 0 and all(l.self_attn.head_dim == 64 for l in layers)


class WhisperEncoder(nn.Module):
    """modeling_whisper.WhisperEncoder: conv1(k3,s1)+GELU -> conv2(k3,s2)+GELU -> + embed_positions -> pre-norm
    layers (MHA without masks, GELU feed-forward) -> layer_norm."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg["d_model"]
        self.num_mel_bins, self.max_source_positions = cfg["num_mel_bins"], cfg["max_source_positions"]
        self.conv1 = nn.Conv1d(self.num_mel_bins, d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(self.max_source_positions, d)
        self.layers = nn.ModuleList([_EncoderLayer(d, cfg["encoder_attention_heads"], cfg["encoder_ffn_dim"])
                                     for _ in range(cfg["encoder_layers"])])
        self.layer_norm = nn.LayerNorm(d)

    def _conv_weight(self, conv):
        """[out, in, 3] -> [out, 3*in] with the tap as the slow index: one row of the strided window."""
        key = (conv.weight.data_ptr(), conv.weight._version)
        if getattr(conv, "_gemm_key", None) != key:
            with torch.no_grad():
                conv._gemm_w = conv.weight.permute(0, 2, 1).reshape(conv.weight.shape[0], -1).contiguous()
            conv._gemm_ready = native._Ready(conv.weight.device)  # (see _Attention.stacked)
            conv._gemm_key = key
        conv._gemm_ready.wait(conv.weight.device)
        return conv._gemm_w

    def forward(self, input_features, output_hidden_states=False):
        """input_features [B, n_mels, 2 * max_source_positions] -> last hidden state [B, max_source_positions, d]
        (and, if asked, the tuple of hidden states: embeddings, every layer's output, the last one normalised --
        HF's convention, which ``forward_encoder`` of the reference stacks)."""
        B, C, T = input_features.shape
        expected = 2 * self.max_source_positions
        if T != expected or C != self.num_mel_bins:
            raise ValueError(f"Whisper expects the mel input features to be [B, {self.num_mel_bins}, {expected}], "
                             f"but found {tuple(input_features.shape)}. Pad or trim the audio to the model's chunk length.")
        d = self.conv1.out_channels
        dev = input_features.device
        # time-major copy with one zero frame on either side (the convolutions' padding)
        xp = torch.zeros(B, T + 2, C, dtype=torch.float32, device=dev)
        xp[:, 1: T + 1] = input_features.transpose(1, 2)
        h1 = torch.zeros(B, T + 2, d, dtype=torch.float32, device=dev)  # conv1 output, padded for conv2
        w1, w2 = self._conv_weight(self.conv1), self._conv_weight(self.conv2)
        pos = self.embed_positions.weight
        x = torch.empty(B, T // 2, d, dtype=torch.float32, device=dev)
        for b in range(B):
            # frame t of conv1 reads padded frames t .. t+2: 3*C contiguous floats starting at t*C
            native.gemm_nt_rows(xp[b].reshape(-1), T, 3 * C, C, w1, self.conv1.bias, act=native.ACT_GELU,
                                out=h1[b, 1: T + 1])
            # frame t of conv2 (stride 2) reads padded frames 2t .. 2t+2; + positions after the GELU
            native.gemm_nt_rows(h1[b].reshape(-1), T // 2, 3 * d, 2 * d, w2, self.conv2.bias, residual=pos,
                                act=native.ACT_GELU, out=x[b])
        hidden = [x] if output_hidden_states else None
        Tq = T // 2
        # precision "bf16": the operands of the four contractions of a layer are WRITTEN as bf16 by their producers
        # (LayerNorm, attention, the GELU epilogue) and stream global -> LDS by LDS-DMA (native.gemm_nt_bf16a); the
        # residual stream, the qkv rows and every accumulation stay fp32.  Same roundings as the fp32-activation bf16
        # kernels, at the points where those round on load.
        bf16a = (native.precision() == "bf16" and native.BF16_ACTIVATIONS and att_ok(self.layers) and native.bf16a_ok(d)
                 and native.bf16a_ok(self.layers[0].fc1.out_features))
        # precision "fp8" (BASELINE configs[4]): the same pipeline with e4m3 operands on the 2 x-rate fp8 matrix instruction
        # for the four contractions of a layer -- LayerNorm writes fp8 rows with one scale per row, the weights carry one
        # scale per output channel, the feed-forward hidden layer is written as fp8 by the first contraction's GELU
        # epilogue; the attention runs on bf16 rows as above and its context is quantised row by row for the out-projection
        fp8a = (native.precision() == "fp8" and native.FP8_ACTIVATIONS and att_ok(self.layers) and native.fp8a_ok(d)
                and native.fp8a_ok(self.layers[0].fc1.out_features) and B * Tq >= 256)
        for layer in self.layers:
            att = layer.self_attn
            ln = layer.self_attn_layer_norm
            w_in, b_in = att.stacked(interleave_heads=True)
            if fp8a:
                h = native.layernorm_fp8(x, ln.weight, ln.bias, ln.eps)
                qkv = native.gemm_nt_fp8a(h, w_in, b_in, out_dtype=torch.bfloat16)
                ctx = native.attention_bf16(qkv, None, att.num_heads, att.head_dim ** -0.5)
                x = native.gemm_nt_fp8a(native.quant_rows_fp8(ctx), att.out_proj.weight, att.out_proj.bias, residual=x)
                ln = layer.final_layer_norm
                h = native.layernorm_fp8(x, ln.weight, ln.bias, ln.eps)
                h = native.gemm_nt_fp8a(h, layer.fc1.weight, layer.fc1.bias, act=native.ACT_GELU, out_dtype="fp8")
                x = native.gemm_nt_fp8a(h, layer.fc2.weight, layer.fc2.bias, residual=x)
                if hidden is not None:
                    hidden.append(x)
                continue
            if bf16a:
                h = native.layernorm_bf16(x, ln.weight, ln.bias, ln.eps)
                qkv = native.gemm_nt_bf16a(h, w_in, b_in, out_dtype=torch.bfloat16)
                ctx = native.attention_bf16(qkv, None, att.num_heads, att.head_dim ** -0.5)  # K / V^T tiles shared through LDS
                x = native.gemm_nt_bf16a(ctx, att.out_proj.weight, att.out_proj.bias, residual=x)
                ln = layer.final_layer_norm
                h = native.layernorm_bf16(x, ln.weight, ln.bias, ln.eps)
                h = native.gemm_nt_bf16a(h, layer.fc1.weight, layer.fc1.bias, act=native.ACT_GELU, out_dtype=torch.bfloat16)
                x = native.gemm_nt_bf16a(h, layer.fc2.weight, layer.fc2.bias, residual=x)
                if hidden is not None:
                    hidden.append(x)
                continue
            h = native.layernorm(x, ln.weight, ln.bias, ln.eps)
            qkv = native.gemm_nt(h, w_in, b_in)
            ctx, _ = native.rope_attention(qkv, None, None, None, att.num_heads, att.head_dim ** -0.5)  # no rotation
            x = native.gemm_nt(ctx, att.out_proj.weight, att.out_proj.bias, residual=x)
            ln = layer.final_layer_norm
            h = native.layernorm(x, ln.weight, ln.bias, ln.eps)
            h = native.gemm_nt(h, layer.fc1.weight, layer.fc1.bias, act=native.ACT_GELU)
            x = native.gemm_nt(h, layer.fc2.weight, layer.fc2.bias, residual=x)
            if hidden is not None:
                hidden.append(x)
        x = native.layernorm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        if hidden is not None:
            hidden[-1] = x
            return x, tuple(hidden)
        return x


class WhisperDecoder(nn.Module):
    """Parameter tree of modeling_whisper.WhisperDecoder (embed_tokens, learned embed_positions, pre-norm layers with
    self- and cross-attention, GELU feed-forward, layer_norm); the arithmetic is the KV-cached decoder step of
    csrc/decoder.hip, reached through ``Whisper.forward_decoder`` and the Whisper searchers."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg["d_model"]
        self.max_target_positions = cfg["max_target_positions"]
        self.embed_tokens = nn.Embedding(cfg["vocab_size"], d, padding_idx=cfg.get("pad_token_id"))
        self.embed_positions = nn.Embedding(self.max_target_positions, d)
        self.layers = nn.ModuleList([_DecoderLayer(d, cfg["decoder_attention_heads"], cfg["decoder_ffn_dim"])
                                     for _ in range(cfg["decoder_layers"])])
        self.layer_norm = nn.LayerNorm(d)


class WhisperModel(nn.Module):
    def __init__(self, cfg, encoder_only=False):
        super().__init__()
        self.config = dict(cfg)
        self.encoder = WhisperEncoder(cfg)
        self.decoder = None if encoder_only else WhisperDecoder(cfg)


def _read_checkpoint(source):
    """{name: tensor} from a HuggingFace model directory (model.safetensors or pytorch_model.bin)."""
    st = os.path.join(source, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file

        return load_file(st)
    pt = os.path.join(source, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu")
    raise FileNotFoundError(f"{source}: neither model.safetensors nor pytorch_model.bin (a LOCAL HuggingFace model "
                            "directory is needed; this package does not download)")


class Whisper(WhisperLogMel):
    """speechbrain.integrations.huggingface.whisper.Whisper (whisper.py:59-665) on the MI355X kernels.

    ``source`` is a LOCAL HuggingFace model directory (config.json, model.safetensors / pytorch_model.bin,
    preprocessor_config.json, and -- unless ``encoder_only`` -- the tokenizer files); nothing is downloaded.
    ``forward(wav, decoder_input_ids)``, ``forward_encoder(mel)``, ``forward_decoder(encoder_states,
    decoder_input_ids)`` keep the reference's signatures and return values (attention maps are not produced:
    ``output_attentions`` raises)."""

    def __init__(self, source, save_path=None, sampling_rate=16000, encoder_only=False, freeze=False, freeze_encoder=False,
                 output_attentions=False, output_all_hiddens=False, language=None, task="transcribe"):
        with open(os.path.join(source, "config.json")) as f:
            cfg = json.load(f)
        if cfg.get("activation_function", "gelu") != "gelu":
            raise NotImplementedError("Whisper with an activation other than GELU")
        if output_attentions:
            raise NotImplementedError("attention maps are not produced by the fused decoder kernels")
        fe = {}
        pp = os.path.join(source, "preprocessor_config.json")
        if os.path.exists(pp):
            with open(pp) as f:
                fe = json.load(f)
        n_fft, hop = fe.get("n_fft", N_FFT), fe.get("hop_length", HOP_LENGTH)
        n_samples = fe.get("n_samples", fe.get("chunk_length", CHUNK_LENGTH) * fe.get("sampling_rate", SAMPLE_RATE))
        super().__init__(n_mels=cfg["num_mel_bins"], n_fft=n_fft, hop_length=hop, n_samples=n_samples)
        self.sampling_rate, self.encoder_only, self.freeze, self.freeze_encoder = sampling_rate, encoder_only, freeze, freeze_encoder
        self.output_attentions, self.output_all_hiddens = output_attentions, output_all_hiddens
        self.language, self.task = language, task
        self.model = WhisperModel(cfg, encoder_only=encoder_only)
        state = _read_checkpoint(source)
        state = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state.items()}
        state = {k: v.float() for k, v in state.items()
                 if k != "proj_out.weight" and not (encoder_only and k.startswith("decoder."))}
        self.model.load_state_dict(state, strict=True)
        self.tokenizer = None
        if not encoder_only:
            self.load_tokenizer(source, bos_token="<|startoftranscript|>")
            if self.tokenizer is not None and self.is_multilingual:
                self.tokenizer.set_prefix_tokens(language=self.language or "en", task=self.task)
        for p in self.model.parameters():  # inference path: parameters are constants
            p.requires_grad_(False)

    @classmethod
    def from_config(cls, cfg, encoder_only=True, seed=0):
        """A randomly initialised model of the given HuggingFace config dict (benchmarks: no checkpoint on disk)."""
        self = cls.__new__(cls)
        WhisperLogMel.__init__(self, n_mels=cfg["num_mel_bins"])
        self.sampling_rate, self.encoder_only, self.freeze, self.freeze_encoder = SAMPLE_RATE, encoder_only, True, False
        self.output_attentions, self.output_all_hiddens, self.language, self.task = False, False, None, "transcribe"
        torch.manual_seed(seed)
        self.model = WhisperModel(cfg, encoder_only=encoder_only)
        self.tokenizer = None
        for p in self.model.parameters():
            p.requires_grad_(False)
        return self

    def load_tokenizer(self, source, **kwargs):
        """The HuggingFace tokenizer of the model directory (host-side text processing, as in the reference,
        huggingface.py:420-430); a directory without tokenizer files leaves ``tokenizer`` None."""
        if not any(os.path.exists(os.path.join(source, f)) for f in ("tokenizer.json", "vocab.json")):
            return
        from transformers import AutoTokenizer

        self.tokenizer = AutoTokenizer.from_pretrained(source, **kwargs)

    @property
    def is_multilingual(self):
        """whisper.py:604-607."""
        return self.model.config["vocab_size"] >= 51865

    def forward(self, wav, decoder_input_ids=None):
        """whisper.py:212-256: mel -> encoder (-> decoder logits for the given prefix)."""
        with torch.no_grad():
            out_encoder = self.forward_encoder(self._get_mel(wav))
            if self.encoder_only:
                return out_encoder
            states = out_encoder[-1] if self.output_all_hiddens else out_encoder
            logits, attn, _ = self.forward_decoder(states, decoder_input_ids)
            return out_encoder, logits, attn

    def pad_or_trim(self, array, length: int = N_SAMPLES, axis=-1):
        """whisper.py:318-350 -- the default length is the 30-second constant, whatever the feature extractor says."""
        return super().pad_or_trim(array, length, axis)

    def _get_mel(self, wav):
        """whisper.py:258-274."""
        return self.log_mel_spectrogram(self.pad_or_trim(wav))

    def forward_encoder(self, mel):
        """whisper.py:356-378: the last hidden state, or all hidden states stacked [layers+1, B, T', d]."""
        with torch.no_grad():
            if self.output_all_hiddens:
                return torch.stack(self.model.encoder(mel.float().contiguous(), output_hidden_states=True)[1])
            return self.model.encoder(mel.float().contiguous())

    # ---- decoder -------------------------------------------------------------------------------------------------
    def decoder_handle(self):
        """The decoder's weights as the C ABI's sbk_decoder_weights (rebuilt when a parameter changes): q/k/v stacked
        as one in_proj per attention (k without bias), learned positions as the position table, token embedding
        unscaled (or * sqrt(d) with ``scale_embedding``) and tied to the output projection."""
        dec = self.model.decoder
        if dec is None:
            raise RuntimeError("this Whisper was built with encoder_only=True")
        key = tuple((p.data_ptr(), p._version) for p in dec.parameters())
        h = getattr(self, "_dec_handle", None)
        if h is None or h.key != key:
            layers = []
            for L in dec.layers:
                layers.append(dict(
                    ln1=(L.self_attn_layer_norm.weight, L.self_attn_layer_norm.bias), sa_in=L.self_attn.stacked(False),
                    sa_out=(L.self_attn.out_proj.weight, L.self_attn.out_proj.bias),
                    ln2=(L.encoder_attn_layer_norm.weight, L.encoder_attn_layer_norm.bias),
                    ca_in=L.encoder_attn.stacked(False), ca_out=(L.encoder_attn.out_proj.weight, L.encoder_attn.out_proj.bias),
                    ln3=(L.final_layer_norm.weight, L.final_layer_norm.bias), ff1=(L.fc1.weight, L.fc1.bias),
                    ff2=(L.fc2.weight, L.fc2.bias)))
            emb = dec.embed_tokens.weight
            cfg = self.model.config
            h = native.DecoderHandle.from_tensors(
                layers, emb=emb, pe=dec.embed_positions.weight, final_ln=(dec.layer_norm.weight, dec.layer_norm.bias),
                seq=(emb, torch.zeros(emb.shape[0], device=emb.device)), nhead=cfg["decoder_attention_heads"],
                ffn_act=native.ACT_GELU, ln_eps=dec.layer_norm.eps,
                emb_scale=math.sqrt(cfg["d_model"]) if cfg.get("scale_embedding") else 1.0, key=key)
            self._dec_handle = h
        return h

    def forward_decoder(self, encoder_states, decoder_input_ids, use_cache=True, past_key_values=None):
        """whisper.py:380-439: logits [B,L,V] of the decoder for the token prefix (KV-cached inside the call).
        Returns (logits, None, None): attention maps and the HuggingFace cache object are not produced -- stepwise
        decoding is what the Whisper searchers do on the device."""
        if past_key_values is not None:
            raise NotImplementedError("incremental decoding with a HuggingFace cache object: use the Whisper searchers "
                                      "(speechbrain_amd.decoders.seq2seq), which keep the KV cache on the device")
        with torch.no_grad():
            enc = encoder_states.float().contiguous()
            B, T, _ = enc.shape
            ids = decoder_input_ids.to(device=enc.device, dtype=torch.int32).contiguous()
            full = torch.full((B,), T, dtype=torch.int32, device=enc.device)  # every encoder frame is attended to
            hidden = native.decoder_prefix(self.decoder_handle(), ids, enc, full)
            logits = native.gemm_nt(hidden, self.model.decoder.embed_tokens.weight)
        return logits, None, None

    # ---- token ids (whisper.py:441-616; all read from the tokenizer) ------------------------------------------------
    def _id(self, token):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: the model directory holds no tokenizer files (or encoder_only=True)")
        return self.tokenizer.convert_tokens_to_ids(token)

    @property
    def transcribe(self):
        return self._id("<|transcribe|>")

    @property
    def translate(self):
        return self._id("<|translate|>")

    @property
    def bos(self):
        return self._id("<|startoftranscript|>")

    @property
    def eos(self):
        return self._id("<|endoftext|>")

    @property
    def bos_lm(self):
        return self._id("<|startoflm|>")

    @property
    def bos_prev(self):
        return self._id("<|startofprev|>")

    @property
    def no_timestamps(self):
        return self._id("<|notimestamps|>")

    @property
    def timestamp_begin(self):
        return self._id("<|0.00|>")

    @property
    def no_speech(self):
        return self.no_timestamps - 1

    @property
    def non_speech_tokens(self):
        """Token ids of symbols that annotate rather than transcribe (brackets, quotes, note signs, ...): what
        ``suppress_tokens="-1"`` expands to (whisper.py:463-500, after openai/whisper's tokenizer)."""
        have = getattr(self, "_non_speech", None)
        if have is None:
            enc = lambda s: self.tokenizer.encode(s, add_special_tokens=False)  # noqa: E731
            single = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
            multi = "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
            notes = set("♩♪♫♬♭♮♯")  # U+2640..U+267F: their first byte-level token is safe to suppress
            ids = {enc(" -")[0], enc(" '")[0]}  # a hyphen / apostrophe may join words but not start one
            for sym in single + multi + sorted(notes):
                for toks in (enc(sym), enc(" " + sym)):
                    if len(toks) == 1 or sym in notes:
                        ids.add(toks[0])
            have = self._non_speech = tuple(sorted(ids))
        return have

    @property
    def all_language_tokens(self):
        """whisper.py:441-453: the language tokens follow <|startoftranscript|> in the vocabulary, in LANGUAGES order."""
        have = getattr(self, "_lang_tokens", None)
        if have is None:
            from transformers.models.whisper.tokenization_whisper import LANGUAGES

            bos = self.tokenizer.convert_tokens_to_ids(self.tokenizer.bos_token)
            have = self._lang_tokens = tuple(bos + 1 + i for i in range(len(LANGUAGES)))
        return have

    @property
    def all_language_codes(self):
        """whisper.py:455-461."""
        have = getattr(self, "_lang_codes", None)
        if have is None:
            from transformers.models.whisper.tokenization_whisper import LANGUAGES

            have = self._lang_codes = tuple(LANGUAGES.keys())
        return have

    @torch.no_grad()
    def detect_language(self, mel):
        """whisper.py:617-665: one decoder step on <|startoftranscript|>, the logits restricted to the language tokens.
        Returns (language_tokens [B], list of {code: probability})."""
        if getattr(self.tokenizer, "language", None) is None:
            raise ValueError("This model doesn't have language tokens so it can't perform lang id")
        enc_states = self.model.encoder(mel.float().contiguous())
        B = mel.shape[0]
        ids = torch.full((B, 1), self.bos, dtype=torch.int32, device=mel.device)
        logits = self.forward_decoder(enc_states, ids)[0][:, 0]
        mask = torch.ones(logits.shape[-1], dtype=torch.bool, device=logits.device)
        mask[list(self.all_language_tokens)] = False
        logits = logits.masked_fill(mask, -float("inf"))  # (host-side glue on a [B, V] tensor, like the reference)
        language_tokens = logits.argmax(dim=-1)
        probs = logits.softmax(dim=-1).cpu()
        language_probs = [{c: probs[i, j].item() for j, c in zip(self.all_language_tokens, self.all_language_codes)}
                          for i in range(B)]
        return language_tokens, language_probs

    def set_language_token(self, language):
        self.language = language
        self.tokenizer.set_prefix_tokens(language=language)

    def set_task(self, task):
        self.task = task
        self.tokenizer.set_prefix_tokens(task=task)
