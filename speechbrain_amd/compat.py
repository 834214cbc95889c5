"""Drop-in import shim: make ``import speechbrain...`` resolve to this package for the modules on the
EncoderDecoderASR path, so a HyperPyYAML written for the reference (``!new:speechbrain.lobes.features
.Fbank`` ...) instantiates the MI355X implementation.

    import speechbrain_amd.compat; speechbrain_amd.compat.install()

Must not be combined with the real SpeechBrain in one process (namespace caution).
"""
import importlib
import sys
import types

_MODULES = [
    "processing", "processing.features", "lobes", "lobes.features", "lobes.models", "lobes.models.convolution",
    "lobes.models.transformer", "lobes.models.transformer.Conformer", "lobes.models.transformer.Transformer",
    "lobes.models.transformer.TransformerASR", "lobes.models.transformer.TransformerLM", "nnet", "nnet.attention", "nnet.activations", "nnet.CNN",
    "nnet.containers", "nnet.embedding", "nnet.linear", "nnet.normalization", "decoders", "decoders.seq2seq",
    "decoders.scorer", "decoders.utils", "inference", "inference.ASR", "inference.interfaces", "utils",
    "utils.data_utils", "utils.parameter_transfer", "utils.metric_stats", "utils.edit_distance",
    "utils.dynamic_chunk_training", "utils.filter_analysis",
]


def install(name: str = "speechbrain"):
    if name in sys.modules and not getattr(sys.modules[name], "__sbk_shim__", False):
        raise RuntimeError(f"a real '{name}' package is already imported in this process")
    root = types.ModuleType(name)
    root.__sbk_shim__ = True
    root.__path__ = []
    sys.modules[name] = root
    for sub in _MODULES:
        mod = importlib.import_module(f"speechbrain_amd.{sub}")
        sys.modules[f"{name}.{sub}"] = mod
        parent = sys.modules[name if "." not in sub else f"{name}.{sub.rsplit('.', 1)[0]}"]
        setattr(parent, sub.rsplit(".", 1)[-1], mod)
    return root
