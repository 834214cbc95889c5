"""Dynamic Chunk (masked) and streaming (chunk-by-chunk) Conformer encoder, against the REFERENCE's
outputs in tests/golden/streaming.npz (oracle/make_golden.py --streaming-only) -- including the reference's own
strong numerical test, tests/unittests/test_conformer.py:5-98 (masked path == streaming path), replayed on the HIP
kernels.  Tolerances are absolute fp32: 2e-5 against the reference's values (its own two paths differ by up to 8e-7),
mean |masked - streaming| < 1e-6 like the reference's test."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streaming.npz")


def _md(a, b):
    return float((a.cpu() - torch.as_tensor(b)).abs().max())


def test_reference_conformer_layer_streaming_equivalence(backend):
    nat, dev = backend
    from speechbrain_amd.lobes.models.transformer.Conformer import ConformerEncoderLayer
    from speechbrain_amd.lobes.models.transformer.TransformerASR import make_transformer_src_mask
    from speechbrain_amd.nnet.attention import RelPosEncXL
    from speechbrain_amd.utils.dynamic_chunk_training import DynChunkTrainConfig

    g = np.load(GOLD)
    layer = ConformerEncoderLayer(d_model=16, d_ffn=32, nhead=1, kernel_size=5)
    layer.load_state_dict({k[len("layer/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("layer/sd/")})
    layer = layer.to(dev).eval()
    pos = RelPosEncXL(16).to(dev)
    x = torch.from_numpy(g["layer/x"]).to(dev)
    cfg = DynChunkTrainConfig(chunk_size=8, left_context_size=1)
    with torch.no_grad():
        masked, _ = layer(x, src_mask=make_transformer_src_mask(x, dynchunktrain_config=cfg), pos_embs=pos(x),
                          dynchunktrain_config=cfg)
        ctx = layer.make_streaming_context(cfg.left_context_size * cfg.chunk_size)
        chunks = []
        for i in range(3):
            c = x[:, 8 * i: 8 * i + 8].contiguous()
            n = 8 + (0 if ctx.mha_left_context is None else ctx.mha_left_context.size(1))
            chunks.append(layer.forward_streaming(c, ctx, pos_embs=pos.make_pe(n))[0])
        stream = torch.cat(chunks, 1)
    assert float((masked - stream).abs().mean()) < 1.0e-6  # the reference's criterion (test_conformer.py:22,98)
    assert _md(masked, g["layer/masked"]) <= 2e-5 and _md(stream, g["layer/stream"]) <= 2e-5


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_transformer_asr_dynamic_chunk_and_streaming(backend, tag):
    nat, dev = backend
    from speechbrain_amd.lobes.models.transformer.TransformerASR import TransformerASR, make_transformer_src_mask
    from speechbrain_amd.utils.dynamic_chunk_training import DynChunkTrainConfig

    g = np.load(GOLD)
    cs, lc, ks = [int(v) for v in g[f"{tag}/cfg"]]
    cfg = DynChunkTrainConfig(chunk_size=cs, left_context_size=None if lc < 0 else lc)
    tr = TransformerASR(input_size=40, tgt_vocab=30, d_model=32, nhead=4, num_encoder_layers=2, num_decoder_layers=0,
                        d_ffn=64, dropout=0.0, activation=torch.nn.GELU, encoder_module="conformer",
                        attention_type=str(g[f"{tag}/att"]), normalize_before=True, causal=False, kernel_size=ks)
    sd = {k[len(f"{tag}/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/sd/")}
    missing, unexpected = tr.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(".pe") for k in missing)
    tr = tr.to(dev).eval()
    src, wl = torch.from_numpy(g[f"{tag}/src"]).to(dev), torch.from_numpy(g[f"{tag}/wav_len"]).to(dev)
    assert torch.equal(make_transformer_src_mask(src, dynchunktrain_config=cfg).cpu(), torch.from_numpy(g[f"{tag}/src_mask"]))
    with torch.no_grad():
        assert _md(tr.encode(src, wl, dynchunktrain_config=cfg), g[f"{tag}/masked"]) <= 2e-5
        masked = tr.encode(src, None, dynchunktrain_config=cfg)
        assert _md(masked, g[f"{tag}/masked_nopad"]) <= 2e-5
        if lc >= 0:
            ctx = tr.make_streaming_context(cfg)
            stream = torch.cat([tr.encode_streaming(src[:, t0: t0 + cs].contiguous(), ctx)
                                for t0 in range(0, src.shape[1], cs)], 1)
            assert _md(stream, g[f"{tag}/stream"]) <= 2e-5
            assert _md(stream, masked.cpu()) <= 2e-5  # masked == streaming on our kernels as well


def test_chunk_masked_attention_weights(backend):
    """The strip kernel (attention weights requested) applies the same chunk mask: rows sum to 1 over their allowed
    keys, masked entries are exactly 0, and the context equals the flash kernel's."""
    nat, dev = backend
    g = torch.Generator().manual_seed(4)
    B, T, H, Dh = 2, 45, 2, 16
    d = H * Dh
    qkv = torch.randn(B, T, 3 * d, generator=g).to(dev)
    pos = torch.randn(2 * T - 1, d, generator=g).to(dev)
    u, v = (0.1 * torch.randn(d, generator=g)).to(dev), (0.1 * torch.randn(d, generator=g)).to(dev)
    kl = torch.tensor([45, 30], dtype=torch.int32).to(dev)
    for cs, lc in ((7, 1), (16, -1), (5, 0)):
        ctx1, attn = nat.relpos_attention(qkv, pos, u, v, kl, H, 0.25, want_attn=True, chunk_size=cs, left_chunks=lc)
        ctx2, _ = nat.relpos_attention(qkv, pos, u, v, kl, H, 0.25, want_attn=False, chunk_size=cs, left_chunks=lc)
        a = attn.cpu()
        i = torch.arange(T)
        hi = (i // cs + 1) * cs
        lo = torch.zeros_like(i) if lc < 0 else ((i // cs - lc) * cs).clamp(min=0)
        for b in range(B):
            allowed = (i[None] < torch.minimum(hi, torch.tensor(int(kl[b])))[:, None]) & (i[None] >= lo[:, None])
            assert float(a[b][:, ~allowed].abs().max()) == 0.0
            rows = allowed.any(1)
            assert float((a[b].sum(-1)[:, rows] - 1).abs().max()) <= 1e-5
        assert _md(ctx1, ctx2.cpu()) <= 2e-5


def test_streaming_feature_wrapper(backend):
    """lobes/features.py:505-670 around Fbank -> ConvolutionFrontEnd on the HIP kernels: filter bookkeeping equals the
    reference's and the chunk-by-chunk features equal the reference's chunk-by-chunk features."""
    nat, dev = backend
    from speechbrain_amd.lobes.features import Fbank, StreamingFeatureWrapper
    from speechbrain_amd.lobes.models.convolution import ConvolutionFrontEnd
    from speechbrain_amd.utils.filter_analysis import stack_filter_properties

    g = np.load(GOLD)
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32)
    cnn = ConvolutionFrontEnd(input_shape=(8, 10, 80), num_blocks=2, num_layers_per_block=1, out_channels=(8, 4),
                              kernel_sizes=(3, 3), strides=(2, 2), residuals=(False, False))
    cnn.load_state_dict({k[len("fw/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fw/sd/")})
    pipe = torch.nn.Sequential(fb, cnn).to(dev).eval()
    props = stack_filter_properties([fb, cnn])
    wrap = StreamingFeatureWrapper(pipe, props)
    chunk = 640 * 4
    assert [props.window_size, props.stride, wrap.get_required_padding(), wrap.get_output_count_per_pad_frame(),
            wrap.get_recommended_final_chunk_count(chunk)] == [int(v) for v in g["fw/props"]]
    wav = torch.from_numpy(g["fw/wav"]).to(dev)
    with torch.no_grad():
        ctx = wrap.make_streaming_context()
        feats = torch.cat([wrap(wav[:, t0: t0 + chunk], ctx) for t0 in range(0, wav.shape[1], chunk)], 1)
    assert feats.shape == g["fw/feats"].shape
    # each chunk is its own "utterance" for Fbank's per-utterance top_db floor, in the reference as here
    assert _md(feats, g["fw/feats"]) <= 2e-3


def test_streaming_asr_encode_chunk_matches_reference(backend):
    """StreamingASR (inference/ASR.py:978-1363): feature wrapper -> streaming Conformer -> projection, chunk by chunk,
    against the reference's own StreamingASR.encode_chunk outputs; plus the decode / transcribe surface."""
    nat, dev = backend
    from speechbrain_amd.inference.ASR import StreamingASR
    from speechbrain_amd.lobes.features import Fbank, StreamingFeatureWrapper
    from speechbrain_amd.lobes.models.convolution import ConvolutionFrontEnd
    from speechbrain_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    from speechbrain_amd.nnet.containers import LengthsCapableSequential
    from speechbrain_amd.nnet.linear import Linear
    from speechbrain_amd.processing.features import InputNormalization
    from speechbrain_amd.utils.dynamic_chunk_training import DynChunkTrainConfig
    from speechbrain_amd.utils.filter_analysis import stack_filter_properties

    g = np.load(GOLD)
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32)
    norm = InputNormalization(norm_type="global", update_until_epoch=4)
    norm.glob_mean, norm.glob_std, norm.count = torch.from_numpy(g["asr/mean"]), torch.from_numpy(g["asr/std"]), 1
    cnn = ConvolutionFrontEnd(input_shape=(8, 10, 80), num_blocks=2, num_layers_per_block=1, out_channels=(8, 4),
                              kernel_sizes=(3, 3), strides=(2, 2), residuals=(False, False))
    tr = TransformerASR(input_size=80, tgt_vocab=30, d_model=32, nhead=4, num_encoder_layers=2, num_decoder_layers=0,
                        d_ffn=64, dropout=0.0, activation=torch.nn.GELU, encoder_module="conformer",
                        attention_type="RelPosMHAXL", normalize_before=True, causal=False, kernel_size=7)
    proj = Linear(input_size=32, n_neurons=30)

    def sd(prefix):
        return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}

    cnn.load_state_dict(sd("asr/sd_cnn/"))
    tr.load_state_dict(sd("asr/sd_tr/"), strict=False)
    proj.load_state_dict(sd("asr/sd_proj/"))
    front = LengthsCapableSequential(compute_features=fb, normalize=norm, model=cnn)
    wrapper = StreamingFeatureWrapper(front, stack_filter_properties([fb, cnn])).eval()
    asr = StreamingASR(modules={"enc": EncoderWrapper(tr), "proj_enc": proj},
                       hparams={"fea_streaming_extractor": wrapper, "make_decoder_streaming_context": lambda: None,
                                "decoding_function": lambda x, ctx: [t.argmax(-1).tolist() for t in x],
                                "make_tokenizer_streaming_context": lambda: None,
                                "tokenizer_decode_streaming": lambda tok, ids, ctx: " ".join(map(str, ids)),
                                "tokenizer": None}, run_opts={"device": str(dev)})
    wrapper.to(dev)
    cfg = DynChunkTrainConfig(chunk_size=4, left_context_size=2)
    n = asr.get_chunk_size_frames(cfg)
    assert n == int(g["asr/chunk"][0])
    wav = torch.from_numpy(g["asr/wav"])
    ctx = asr.make_streaming_context(cfg)
    encs = [asr.encode_chunk(ctx, wav[:, t0: t0 + n]) for t0 in range(0, wav.shape[1], n)]
    enc = torch.cat(encs, 1)
    assert enc.shape == g["asr/enc"].shape
    assert _md(enc, g["asr/enc"]) <= 2e-4  # fp32; the per-chunk Fbank floor and CNN LayerNorm are in the chain
    ctx2 = asr.make_streaming_context(cfg)
    words = [asr.transcribe_chunk(ctx2, wav[:, t0: t0 + n]) for t0 in range(0, wav.shape[1], n)]
    assert len(words) == 5 and all(len(w) == 2 and isinstance(w[0], str) for w in words)
    assert words[0][0] == " ".join(map(str, torch.as_tensor(g["asr/enc"][0, :4]).argmax(-1).tolist()))
