"""Host-side (CPU) logic of the drop-in surface: state_dict keys / shapes, initialisers, builders."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_state_dict_keys_match_reference():
    from speechbrain_amd.inference.builders import build_modules

    g = np.load(os.path.join(GOLD, "model_tiny_ctc.npz"))
    ref = {k[3:]: g[k].shape for k in g.files if k.startswith("sd/")}
    m = build_modules(dict(d_model=32, nhead=4, d_ffn=64, n_enc=2, n_dec=2, n_fft=512, win_length=32), vocab=40)
    mods = torch.nn.ModuleDict({k: m[k] for k in ("CNN", "Transformer", "seq_lin", "ctc_lin")})
    ours = {k: tuple(v.shape) for k, v in mods.state_dict().items()}
    assert ours == {k: tuple(v) for k, v in ref.items()}


def test_same_seed_initialisation_as_reference():
    """Constructing TransformerASR under the same seed draws the same weights as the reference
    (fingerprint written by oracle/make_golden.py from the real constructors)."""
    from speechbrain_amd.lobes.models.transformer.TransformerASR import TransformerASR

    fp = np.load(os.path.join(GOLD, "init_fingerprint.npz"))
    torch.manual_seed(0)
    tr = TransformerASR(input_size=640, tgt_vocab=100, d_model=48, nhead=4, num_encoder_layers=2, num_decoder_layers=2,
                        d_ffn=96, activation=torch.nn.GELU, encoder_module="conformer", attention_type="RelPosMHAXL",
                        normalize_before=True, causal=False)
    sd = tr.state_dict()
    assert set(sd) == set(fp.files)
    for k in fp.files:
        assert sd[k].numel() == fp[k][2]
        assert abs(float(sd[k].double().sum()) - fp[k][0]) <= 1e-6 * max(1.0, fp[k][1]), k


def test_conformer_l_parameter_counts():
    """encoder 75.9 M, decoder 25.2 M, TransformerASR 104.0 M, CNN 25.5 k parameters."""
    from speechbrain_amd.inference.builders import build_modules

    m = build_modules("L")
    n = lambda mod: sum(p.numel() for p in mod.parameters())  # noqa: E731
    assert abs(n(m["Transformer"].encoder) / 1e6 - 75.9) < 0.1
    assert abs(n(m["Transformer"].decoder) / 1e6 - 25.2) < 0.1
    assert abs(n(m["Transformer"]) / 1e6 - 104.0) < 0.1
    assert n(m["seq_lin"]) == 512 * 5000 + 5000
    assert abs(n(m["CNN"]) / 1e3 - 25.5) < 0.2


def test_fft_radix_plans():
    from speechbrain_amd.processing.features import factor_radices

    for n in (512, 400, 256, 480, 1024):
        r = factor_radices(n)
        assert all(x in (2, 3, 4, 5) for x in r) and int(np.prod(r)) == n


def test_speechbrain_import_shim():
    import importlib
    import sys

    import speechbrain_amd.compat as compat

    assert "speechbrain" not in sys.modules
    try:
        compat.install()
        fbank_cls = importlib.import_module("speechbrain.lobes.features").Fbank
        from speechbrain_amd.lobes.features import Fbank

        assert fbank_cls is Fbank
        from speechbrain.lobes.models.transformer.TransformerASR import TransformerASR  # noqa: F401
        from speechbrain.decoders import S2STransformerBeamSearcher  # noqa: F401
        from speechbrain.inference.ASR import EncoderDecoderASR  # noqa: F401
        from speechbrain.nnet.attention import RelPosMHAXL  # noqa: F401
        from speechbrain.processing.features import STFT, Filterbank, InputNormalization, spectral_magnitude  # noqa: F401
        from speechbrain.utils.parameter_transfer import Pretrainer  # noqa: F401
        from speechbrain.utils.metric_stats import ErrorRateStats  # noqa: F401
        for sub in compat._MODULES:  # every mirrored module resolves through the shim, to the same object
            assert importlib.import_module(f"speechbrain.{sub}") is importlib.import_module(f"speechbrain_amd.{sub}")
    finally:
        for k in [k for k in sys.modules if k == "speechbrain" or k.startswith("speechbrain.")]:
            del sys.modules[k]


def test_hyperpyyaml_subset_loader():
    """utils/hpyaml.py: the HyperPyYAML features inference hyperparams.yaml files use."""
    from speechbrain_amd.utils.hpyaml import load_hyperpyyaml

    text = """
d_model: 32
half: !ref <d_model> // 2
ratio: !ref <d_model> * 1.5
folder: out
save: !ref <folder>/save/<d_model>
act: !name:torch.nn.GELU
leaky: !name:torch.nn.functional.leaky_relu
    negative_slope: 0.2
shape: (8, 10, 80)
quoted: "(1, 2)"
lin: !new:speechbrain.nnet.linear.Linear
    input_size: !ref <d_model>
    n_neurons: 7
pair: [!ref <lin>, !ref <lin>]
ml: !new:torch.nn.ModuleList
    - [!ref <lin>]
cp: !copy <lin>
sm: !new:torch.nn.LogSoftmax
    dim: -1
nested:
    a: !ref <lin>
    b: [1, 2.5, null, true]
pick: !ref <nested[b][1]>
applied: !apply:max
    - 3
    - 9
"""
    h = load_hyperpyyaml(text)
    assert h["half"] == 16 and h["ratio"] == 48.0 and h["save"] == "out/save/32"
    assert h["act"] is torch.nn.GELU and h["shape"] == (8, 10, 80) and h["quoted"] == "(1, 2)"
    assert float(h["leaky"](torch.tensor(-1.0))) == pytest.approx(-0.2)
    assert type(h["lin"]).__module__ == "speechbrain_amd.nnet.linear"  # speechbrain.* is served by speechbrain_amd.*
    assert h["pair"][0] is h["lin"] and h["pair"][1] is h["lin"] and h["ml"][0] is h["lin"] and h["nested"]["a"] is h["lin"]
    assert h["cp"] is not h["lin"] and torch.equal(h["cp"].w.weight, h["lin"].w.weight)
    assert h["nested"]["b"] == [1, 2.5, None, True] and h["pick"] == 2.5 and h["applied"] == 9
    assert load_hyperpyyaml(text, overrides={"d_model": 64})["lin"].w.weight.shape == (7, 64)
    assert load_hyperpyyaml(text, overrides="d_model: 16")["half"] == 8
    # nested overrides change ONE argument of an object / one sub-key of a mapping (hyperpyyaml recursive_update)
    h2 = load_hyperpyyaml(text, overrides={"lin": {"n_neurons": 9}, "nested": {"b": [5, 6]}, "sm": {"dim": 0}})
    assert h2["lin"].w.weight.shape == (9, 32) and h2["pair"][0] is h2["lin"]
    assert h2["nested"]["b"] == [5, 6] and h2["pick"] == 6 and h2["nested"]["a"] is h2["lin"] and h2["sm"].dim == 0
    assert load_hyperpyyaml(text, overrides="lin: {n_neurons: 3}")["lin"].w.weight.shape == (3, 32)
    live = torch.nn.Identity()  # a live object replaces the whole key
    assert load_hyperpyyaml(text, overrides={"sm": live, "extra": 4})["sm"] is live
    assert load_hyperpyyaml(text, overrides={"extra": 4})["extra"] == 4
    with pytest.raises(ValueError):
        load_hyperpyyaml("x: !PLACEHOLDER\n")
    assert load_hyperpyyaml("x: !PLACEHOLDER\ny: !ref <x>", overrides={"x": 3})["y"] == 3
    with pytest.raises(KeyError):
        load_hyperpyyaml("y: !ref <missing>")
    with pytest.raises(ValueError):
        load_hyperpyyaml("a: !ref <b>\nb: !ref <a>")
    with pytest.raises(ImportError):
        load_hyperpyyaml("x: !new:speechbrain.lobes.models.ECAPA_TDNN.ECAPA_TDNN")  # outside the path
    with pytest.raises(NotImplementedError):
        load_hyperpyyaml("x: !include:other.yaml")


def test_pretrainer_local_sources(tmp_path):
    """utils/parameter_transfer.py: default <source>/<name>.ckpt, explicit paths, conditions, hooks, errors."""
    from speechbrain_amd.nnet.linear import Linear
    from speechbrain_amd.processing.features import InputNormalization
    from speechbrain_amd.utils.parameter_transfer import Pretrainer

    src = Linear(input_size=4, n_neurons=3)
    torch.save(src.state_dict(), tmp_path / "lin.ckpt")
    other = tmp_path / "elsewhere"
    other.mkdir()
    norm_src = InputNormalization(norm_type="global")
    norm_src.glob_mean, norm_src.glob_std, norm_src.count = torch.arange(5.0), torch.ones(5) * 2, 7
    norm_src._save(other / "stats.ckpt")
    lin, norm, skipped, hooked = Linear(input_size=4, n_neurons=3), InputNormalization(norm_type="global"), Linear(input_size=4, n_neurons=3), {}
    pt = Pretrainer(loadables={"lin": lin, "norm": norm, "skipped": skipped, "hooked": hooked},
                    paths={"norm": str(other / "stats.ckpt"), "hooked": str(tmp_path / "lin.ckpt")},
                    conditions={"skipped": False}, custom_hooks={"hooked": lambda obj, path: obj.update(path=str(path))})
    assert Pretrainer.split_path("a/b/c.ckpt") == ("a/b", "c.ckpt") and Pretrainer.split_path("c.ckpt") == ("./", "c.ckpt")
    with pytest.raises(RuntimeError):
        pt.load_collected()  # before collect_files
    got = pt.collect_files(default_source=str(tmp_path))
    assert set(got) == {"lin", "norm", "hooked"}
    pt.load_collected()
    assert torch.equal(lin.w.weight, src.w.weight) and torch.equal(norm.glob_mean, torch.arange(5.0)) and norm.count == 7
    assert hooked["path"].endswith("lin.ckpt")
    with pytest.raises(FileNotFoundError):
        Pretrainer(loadables={"absent": lin}).collect_files(default_source=str(tmp_path))
    with pytest.raises(ValueError):
        Pretrainer(loadables={"nopath": lin}).collect_files()


def test_wav_reader_formats(tmp_path):
    """inference/interfaces.py:read_wav -- the sample formats libsndfile converts to float the same way."""
    import struct

    from speechbrain_amd.inference.interfaces import read_wav

    rng = np.random.default_rng(0)
    ref = np.clip(rng.normal(0, 0.3, (1000, 2)), -0.99, 0.99)

    def write(name, code, bits, payload, extensible=False):
        ch, sr = 2, 16000
        block = ch * bits // 8
        fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else code, ch, sr, sr * block, block, bits)
        if extensible:
            fmt += struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", code) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00"
        chunks += b"data" + struct.pack("<I", len(payload)) + payload
        path = tmp_path / name
        path.write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
        return path

    i16 = np.round(ref * 32767).astype("<i2")
    x, sr = read_wav(write("p16.wav", 1, 16, i16.tobytes()))
    assert sr == 16000 and x.shape == (1000, 2) and np.array_equal(x, i16.astype(np.float32) / 32768.0)
    x, _ = read_wav(write("p16x.wav", 1, 16, i16.tobytes(), extensible=True))
    assert np.array_equal(x, i16.astype(np.float32) / 32768.0)
    u8 = np.round(ref * 127 + 128).astype(np.uint8)
    x, _ = read_wav(write("p8.wav", 1, 8, u8.tobytes()))
    assert np.array_equal(x, (u8.astype(np.float32) - 128) / 128)
    i32 = np.round(ref * 2147483000).astype("<i4")
    x, _ = read_wav(write("p32.wav", 1, 32, i32.tobytes()))
    assert np.allclose(x, i32 / 2147483648.0, atol=1e-7)
    i24 = np.round(ref * 8388607).astype(np.int32)
    b24 = np.stack([(i24 >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.uint8)
    x, _ = read_wav(write("p24.wav", 1, 24, b24.tobytes()))
    assert np.array_equal(x, i24.astype(np.float32) / 8388608.0)
    f32 = ref.astype("<f4")
    x, _ = read_wav(write("f32.wav", 3, 32, f32.tobytes()))
    assert np.array_equal(x, f32)
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.wav"
        bad.write_bytes(b"OggS" + b"\x00" * 40)
        read_wav(bad)
    with pytest.raises(NotImplementedError):
        read_wav(write("adpcm.wav", 2, 4, b"\x00" * 64))


def test_bench_roofline_entry_prices_each_kernel_against_its_own_peak():
    """bench.py's roofline object: fp32-MFMA kernels against 157.3 TF/s, the split-operand contraction against the dense
    bf16 MFMA peak / 6 (with the fraction of the fp32 peak beside it), memory-bound kernels against 8 TB/s."""
    import bench

    v = {"ms": 1000.0, "count": 10, "flops": 139.0e12, "bytes": 1.0e12}
    e = bench.roofline_entry("gemm_nt_f32x3", v, 4000.0)
    assert e["bound"] == "mfma" and abs(e["peak"] - 2500.0 / 6.0) < 0.1
    assert abs(e["frac"] - 139.0 / (2500.0 / 6.0)) < 1e-3 and abs(e["frac_of_fp32_mfma_peak"] - 139.0 / 157.3) < 1e-3
    assert e["launches"] == 10 and abs(e["share_of_gpu_time"] - 0.25) < 1e-9 and "peak_note" in e
    for name in ("gemm_nt_x3p", "gemm_x3r"):  # (pre-split operands / the decode step's few-row projections: the same six bf16 products)
        e = bench.roofline_entry(name, v, 4000.0)
        assert abs(e["peak"] - 2500.0 / 6.0) < 0.1 and abs(e["frac"] - 139.0 / (2500.0 / 6.0)) < 1e-3
    e = bench.roofline_entry("gemm_nt_persistent", v, 4000.0)
    assert e["peak"] == 157.3 and "frac_of_fp32_mfma_peak" not in e
    e = bench.roofline_entry("cross_attn_step", v, 4000.0)
    assert e["bound"] == "hbm" and e["unit"] == "GB/s" and abs(e["achieved"] - 1000.0) < 1e-6 and e["peak"] == 8000.0


def test_bench_traffic_rows_of_this_rounds_kernels_only(tmp_path):
    """bench.pmc_traffic: the `traffic` field of the roofline object is the PMC figure of THIS round's kernel as shipped or null --
    a row of an earlier round, and a row counted before a later change of the kernel's schedule ("superseded"), are quoted in
    the note, never reported; the committed profiles/pmc_traffic.json parses and yields a note for the dominant kernel."""
    import json

    import bench

    now = bench.PMC_ROUND
    rows = {"_source": "test", "gemm_x3r": {"round": now, "bytes_per_launch": 100, "algorithmic_bytes_per_launch": 40, "note": "n", "commit": "c"},
            "gemm_nt_x3p": {"round": now - 1, "bytes_per_launch": 7, "algorithmic_bytes_per_launch": 3, "note": "old"},
            "_mfma_busy": {"gemm_x3r M=1": {"round": now, "mfma_busy": 0.25}, "gemm_x3r M=2": {"round": now - 1, "mfma_busy": 0.5}}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(rows))
    roof = {}
    assert bench.pmc_traffic("gemm_x3r", roof, str(f)) == 100
    assert "algorithmic 40" in roof["traffic_note"] and roof["traffic_provenance"]["commit"] == "c" and roof["mfma_busy_pmc"] == {"gemm_x3r M=1": 0.25}
    roof = {}
    assert bench.pmc_traffic("gemm_nt_x3p", roof, str(f)) is None and "traffic_note" not in roof  # an earlier round's row
    rows["gemm_x3r"]["superseded"] = "counted before the tile order changed"
    f.write_text(json.dumps(rows))
    roof = {}
    assert bench.pmc_traffic("gemm_x3r", roof, str(f)) is None
    assert "not measured for the kernel as shipped" in roof["traffic_note"] and "100 B/launch" in roof["traffic_note"]
    roof = {}
    t = bench.pmc_traffic("gemm_x3r", roof)  # the committed file: this round's row of the kernel as shipped
    assert "traffic_note" in roof and "traffic_provenance" in roof and t and t > 9437184 and roof["mfma_busy_pmc"]

