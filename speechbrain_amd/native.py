"""ctypes binding of libsbk_hip.so (include/sbk.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every
numeric op on the hot path is one of the C-ABI entry points below.  There is NO
CPU fallback: a CPU tensor, or a missing library, raises.
"""

from __future__ import annotations

import contextlib
import ctypes
import os
import threading
import weakref
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsbk_hip.so")

ACT_NONE, ACT_SWISH, ACT_GELU, ACT_RELU, ACT_LEAKY_RELU = 0, 1, 2, 3, 4

_lib = None


class SbkError(RuntimeError):
    pass


class DecoderLayer(ctypes.Structure):  # sbk_decoder_layer
    _fields_ = [(n, c_void_p) for n in (
        "ln1_g", "ln1_b", "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ln2_g", "ln2_b", "ca_in_w", "ca_in_b",
        "ca_out_w", "ca_out_b", "ln3_g", "ln3_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b", "sa_in_wf", "sa_in_bf", "ca_q_wf",
        "ca_q_bf", "ff1_wf", "ff1_bf", "ca_kv_w3", "sa_in_wp", "sa_out_wp", "ca_q_wp", "ca_out_wp", "ff1_wp", "ff2_wp",
        "sa_in_wfp", "ca_q_wfp", "ff1_wfp")]


class DecoderWeights(ctypes.Structure):  # sbk_decoder_weights
    _fields_ = [("layers", POINTER(DecoderLayer)), ("emb", c_void_p), ("pe", c_void_p), ("final_ln_g", c_void_p),
                ("final_ln_b", c_void_p), ("seq_w", c_void_p), ("seq_b", c_void_p), ("seq_wf", c_void_p),
                ("seq_bf", c_void_p), ("d_model", c_int32),
                ("nhead", c_int32), ("d_ffn", c_int32), ("n_layers", c_int32), ("vocab", c_int32),
                ("max_len", c_int32), ("ffn_act", c_int32), ("ln_eps", c_float), ("emb_scale", c_float),
                ("seq_w3", c_void_p), ("seq_wp", c_void_p), ("seq_wfp", c_void_p)]


class LMLayer(ctypes.Structure):  # sbk_lm_layer
    _fields_ = [(n, c_void_p) for n in ("in_w", "in_b", "out_w", "out_b", "ln1_g", "ln1_b", "ff1_w", "ff1_b", "ff2_w",
                                        "ff2_b", "ln2_g", "ln2_b")]


class LMWeights(ctypes.Structure):  # sbk_lm_weights
    _fields_ = [("layers", POINTER(LMLayer))] + [(n, c_void_p) for n in (
        "emb", "pe", "final_ln_g", "final_ln_b", "out0_w", "out0_b", "out_ln_g", "out_ln_b", "out2_w", "out2_b")] + [
        (n, c_int32) for n in ("d_model", "nhead", "d_ffn", "n_layers", "vocab", "max_len", "ffn_act",
                               "normalize_before", "pad_idx")] + [("ln_eps", c_float)]


class SearchConfig(ctypes.Structure):  # sbk_search_config
    _fields_ = [("bos", c_int32), ("eos", c_int32), ("blank", c_int32), ("beam", c_int32), ("min_steps", c_int32),
                ("max_steps", c_int32), ("length_normalization", c_int32), ("using_eos_threshold", c_int32),
                ("check_every", c_int32), ("overlap_ctc", c_int32), ("ctc_weight", c_float), ("temperature", c_float),
                ("eos_threshold", c_float), ("minus_inf", c_float), ("lm_weight", c_float), ("lm_temperature", c_float),
                ("lm", POINTER(LMWeights)), ("topk", c_int32), ("utt_min_steps", c_void_p),
                ("utt_max_steps", c_void_p), ("graph_mode", c_int32), ("ctc_candidates", c_int32),
                ("ctc_window_size", c_int32), ("prompt", c_void_p), ("prompt_len", c_int32), ("temperature_post", c_int32), ("logit_bias", c_void_p),
                ("first_bias", c_void_p), ("probe_pos", c_int32), ("probe_token", c_int32), ("out_probe", c_void_p),
                ("ctc_w3", c_void_p)]


def _declare(lib):
    p, i, f = c_void_p, c_int, c_float
    sig = {
        "sbk_abi_version": ([], c_int),
        "sbk_last_error": ([], c_char_p),
        "sbk_stream_workspace_bytes": ([], ctypes.c_size_t),
        "sbk_stream_workspace_set": ([p, p, ctypes.c_size_t], c_int),
        "sbk_stream_workspace_release": ([p], c_int),
        "sbk_prof_enable": ([i], None),
        "sbk_prof_reset": ([], None),
        "sbk_prof_report": ([ctypes.c_char_p, ctypes.c_size_t], ctypes.c_size_t),
        "sbk_prof_ctc_psi_repeat_f32": ([p, p, p, p, p, i, i, i, i, i, i, POINTER(c_float), p], c_int),
        "sbk_prof_set_knob": ([i, i], None),
        "sbk_prof_get_knob": ([i], c_int),
        "sbk_ctc_scorer_workspace_bytes": ([i, i, i, i], ctypes.c_size_t),
        "sbk_ctc_scorer_reset_f32": ([p, p, p, ctypes.c_size_t, i, i, i, i, i, p], c_int),
        "sbk_ctc_scorer_score_f32": ([p, p, p, ctypes.c_size_t, p, i, p, i, p, i, i, i, i, i, i, p], c_int),
        "sbk_ctc_scorer_permute_f32": ([p, p, ctypes.c_size_t, p, p, p, i, p, i, i, i, i, i, i, p], c_int),
        "sbk_prof_persist_stamps": ([p, i], c_int),
        "sbk_prof_mfma_peak_f32": ([p, i, i, i, POINTER(c_float), p], c_int),
        "sbk_prof_stream_f32": ([p, p, ctypes.c_long, i, i, POINTER(c_float), p], c_int),
        "sbk_prof_gemm_repeat_f32": ([p, p, p, i, i, i, p, ctypes.c_size_t, i, POINTER(c_float), p], c_int),
        "sbk_pcm16_to_f32": ([p, p, ctypes.c_long, i, p], c_int),
        "sbk_fbank_f32": ([p, p, p, POINTER(c_int32), i, p, p, p, p, p, i, i, i, i, i, i, f, f, p, p, f, p], c_int),
        "sbk_whisper_log_mel_f32": ([p, p, p, POINTER(c_int32), i, p, p, p, p, p, p, i, i, i, i, i, i, p], c_int),
        "sbk_stft_f32": ([p, p, p, POINTER(c_int32), i, p, i, i, i, i, p], c_int),
        "sbk_spectral_magnitude_f32": ([p, p, ctypes.c_long, f, i, f, p], c_int),
        "sbk_amplitude_to_db_f32": ([p, p, i, ctypes.c_long, f, f, f, f, p], c_int),
        "sbk_input_norm_global_f32": ([p, p, p, p, i, i, f, p], c_int),
        "sbk_input_norm_stats_workspace_bytes": ([i, i], ctypes.c_size_t),
        "sbk_input_norm_stats_f32": ([p, p, p, p, i, i, i, i, i, f, i, p], c_int),
        "sbk_gemm_nt_f32": ([p, i, p, i, p, p, i, p, i, i, i, i, i, f, p, i, p], c_int),
        "sbk_f32_to_bf16": ([p, p, ctypes.c_long, p], c_int),
        "sbk_split_bf16x3": ([p, i, p, i, i, p], c_int),
        "sbk_gemm_nt_f32x3": ([p, i, p, p, p, i, p, i, i, i, i, i, f, p, i, p], c_int),
        "sbk_x3p_panel_bytes": ([i, i], ctypes.c_size_t),
        "sbk_split_x3p": ([p, i, p, i, i, p], c_int),
        "sbk_layernorm_x3p": ([p, p, p, p, i, i, f, i, p], c_int),
        "sbk_gemm_nt_x3r": ([p, i, p, p, p, i, p, i, i, i, i, i, f, p], c_int),
        "sbk_gemm_ln_nt_x3r": ([p, i, p, p, p, i, p, i, i, i, i, f, i, f, p], c_int),
        "sbk_quant_rows_fp8": ([p, i, p, p, i, i, p], c_int),
        "sbk_quant_rows_bf16_fp8": ([p, i, p, p, i, i, p], c_int),
        "sbk_layernorm_fp8o": ([p, p, p, p, p, i, i, f, i, p], c_int),
        "sbk_gemm_nt_fp8a": ([p, i, p, p, i, p, p, p, i, p, i, p, i, p, i, f, i, i, i, i, f, p], c_int),
        "sbk_input_norm_global_masked_f32": ([p, p, p, p, p, i, i, i, f, p], c_int),
        "sbk_gemm_nt_x3p": ([p, p, p, p, i, p, i, p, i, i, i, i, f, p, i, p], c_int),
        "sbk_gemm_nt_bf16": ([p, i, p, i, p, p, i, p, i, i, i, i, i, f, p, i, p], c_int),
        "sbk_gemm_nt_f16": ([p, i, p, i, p, p, i, p, i, i, i, i, i, f, p, i, p], c_int),
        "sbk_gemm_nt_bf16a": ([p, i, p, i, p, p, i, p, i, p, i, i, i, i, i, f, p], c_int),
        "sbk_layernorm_bf16o": ([p, p, p, p, i, i, f, i, p], c_int),
        "sbk_rope_attention_bf16o": ([p, p, p, p, p, i, i, i, i, i, f, i, i, p], c_int),
        "sbk_attention_bf16io_workspace_bytes": ([i, i, i], ctypes.c_size_t),
        "sbk_attention_bf16io": ([p, p, p, p, i, i, i, i, f, p], c_int),
        "sbk_f32_to_f16": ([p, p, ctypes.c_long, p], c_int),
        "sbk_gemm_nt_fp8": ([p, i, p, p, i, f, p, p, i, p, i, i, i, i, i, f, p, i, p], c_int),
        "sbk_f32_to_fp8": ([p, p, ctypes.c_long, f, p], c_int),
        "sbk_absmax_f32": ([p, ctypes.c_long, p, p], c_int),
        "sbk_gemm_ln_nt_f32": ([p, i, p, i, p, p, i, p, i, i, i, i, f, i, f, p], c_int),
        "sbk_gemm_nt_splitk_f32": ([p, i, p, i, p, p, i, p, i, i, i, i, i, f, p, ctypes.c_size_t, p], c_int),
        "sbk_conv_block_f32": ([p, p, p, p, p, p, i, i, i, i, i, f, f, p], c_int),
        "sbk_relpos_attention_f32": ([p, p, p, p, p, p, p, i, i, i, i, f, i, i, p], c_int),
        "sbk_rope_attention_f32": ([p, p, p, p, p, p, i, i, i, i, i, f, i, i, p], c_int),
        "sbk_rope_attention_bf16": ([p, p, p, p, p, i, i, i, i, i, f, i, i, p], c_int),
        "sbk_glu_dwconv_f32": ([p, p, p, p, i, i, i, i, i, p], c_int),
        "sbk_layernorm_f32": ([p, p, p, p, i, i, f, i, p], c_int),
        "sbk_log_softmax_f32": ([p, p, i, i, f, f, p], c_int),
        "sbk_beam_search_workspace_bytes": ([POINTER(DecoderWeights), POINTER(SearchConfig), i, i], ctypes.c_size_t),
        "sbk_beam_search_f32": ([POINTER(DecoderWeights), POINTER(SearchConfig), p, p, p, p, p, ctypes.c_size_t, p, p,
                                 p, p, p, p, p, POINTER(c_int32), i, i, p], c_int),
        "sbk_greedy_search_workspace_bytes": ([POINTER(DecoderWeights), i, i, i], ctypes.c_size_t),
        "sbk_greedy_search_f32": ([POINTER(DecoderWeights), p, p, p, ctypes.c_size_t, p, p, p, POINTER(c_int32), i, i,
                                   i, i, i, i, i, p], c_int),
        "sbk_prompted_greedy_search_workspace_bytes": ([POINTER(DecoderWeights), i, i, i, i], ctypes.c_size_t),
        "sbk_prompted_greedy_search_f32": ([POINTER(DecoderWeights), p, p, p, i, p, p, p, ctypes.c_size_t, p, p, i, i, p, p,
                                            POINTER(c_int32), i, i, i, i, i, p], c_int),
        "sbk_decoder_prefix_workspace_bytes": ([POINTER(DecoderWeights), i, i, i], ctypes.c_size_t),
        "sbk_decoder_prefix_f32": ([POINTER(DecoderWeights), p, p, p, p, ctypes.c_size_t, p, i, i, i, p], c_int),
        "sbk_lm_prefix_workspace_bytes": ([POINTER(LMWeights), i, i], ctypes.c_size_t),
        "sbk_lm_prefix_f32": ([POINTER(LMWeights), p, p, ctypes.c_size_t, p, i, i, p], c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    return sig


EXPORTS: tuple = ()


def load(path: Optional[str] = None):
    """Load the HIP library (idempotent).  Raises if it has not been built."""
    global _lib, EXPORTS
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise SbkError(
            f"{path} is missing: the MI355X kernels are not built. Run `python -m speechbrain_amd.csrc.build` "
            "(hipcc, gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(path)
    EXPORTS = tuple(_declare(lib).keys())
    if lib.sbk_abi_version() != 11:
        raise SbkError(f"ABI version mismatch: {lib.sbk_abi_version()}")
    _lib = lib
    return lib


def _chk(rc: int, what: str):
    if rc != 0:
        raise SbkError(f"{what} failed (rc={rc}): {_lib.sbk_last_error().decode()}")


def _dev_ok(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise SbkError(
                "speechbrain_amd ops run on an MI355X (HIP) device only; got a CPU tensor. "
                "Move the module and inputs to 'cuda' (there is no CPU fallback)."
            )
        if not t.is_contiguous():
            raise SbkError("non-contiguous tensor passed to a kernel")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


# The persistent / stream-K kernels need slabs and arrival tickets per stream; the library allocates nothing (include/sbk.h,
# "stream workspace"): the binding registers ONE torch-allocated workspace per (library, device, stream) the first time an
# op is issued on that stream, and keeps it for the life of the process (torch's streams come from a fixed pool and are
# never destroyed).  Every op fetches its stream through _stream(), so no call can reach a kernel without one -- the same
# kernels run on every stream, and a worker stream's first launch allocates nothing inside the library.
_STREAM_WS = {}
_STREAM_WS_LOCK = threading.RLock()
# torch hands out stream HANDLES from a pool (32 per priority): two live Stream objects -- of two transcribers, or a
# transcriber's and ShardedTranscriber's copy stream -- can be the same handle.  An owner that means to release a stream's
# workspace when it retires first RETAINS it; the workspace goes back only when its last owner has released it.
_STREAM_WS_REFS = {}


def retain_stream_workspace(stream):
    """Declare an owner of the (device, handle) of ``stream`` (a ConcurrentTranscriber for each of its worker streams)."""
    key = (id(load()), stream.device.index, stream.cuda_stream)
    with _STREAM_WS_LOCK:
        _STREAM_WS_REFS[key] = _STREAM_WS_REFS.get(key, 0) + 1


def _register_workspace(key, device, handle):
    lib = load()
    with _STREAM_WS_LOCK:
        if key in _STREAM_WS:
            return
        nbytes = lib.sbk_stream_workspace_bytes()
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        off = (-ws.data_ptr()) % 256
        # the library keys its table on hipGetDevice(): make the tensor's device the current one for the call
        ctx = torch.cuda.device(device) if getattr(device, "type", None) == "cuda" else contextlib.nullcontext()
        with ctx:
            _chk(lib.sbk_stream_workspace_set(c_void_p(handle) if handle is not None else None, c_void_p(ws.data_ptr() + off), nbytes),
                 "sbk_stream_workspace_set")
        _STREAM_WS[key] = ws


def forget_stream_owner(stream):
    """An owner of ``stream`` went away WITHOUT releasing (a transcriber dropped without ``close()``): its claim is withdrawn, the
    workspace stays registered -- the handle's next owner releases it, or it lives as long as the process (one per pooled handle)."""
    key = (id(load()), stream.device.index, stream.cuda_stream)
    with _STREAM_WS_LOCK:
        left = _STREAM_WS_REFS.get(key, 0) - 1
        if left > 0:
            _STREAM_WS_REFS[key] = left
        else:
            _STREAM_WS_REFS.pop(key, None)


def release_stream_workspace(stream) -> bool:
    """Give back the workspace registered for a ``torch.cuda.Stream`` that will not be used again (a worker stream of a
    ConcurrentTranscriber that is being closed): sbk_stream_workspace_release, then the ~134 MB return to torch's allocator once
    the stream has drained.  Streams that never issued an op have none (False)."""
    lib = load()
    key = (id(lib), stream.device.index, stream.cuda_stream)
    with _STREAM_WS_LOCK:
        left = _STREAM_WS_REFS.get(key, 0) - 1
        if left > 0:  # another live owner of the same pooled handle: it keeps the workspace
            _STREAM_WS_REFS[key] = left
            return False
        _STREAM_WS_REFS.pop(key, None)
        if key not in _STREAM_WS:
            return False
    stream.synchronize()  # (outside the lock: other streams' first launches register their workspaces meanwhile)
    with _STREAM_WS_LOCK:
        if _STREAM_WS_REFS.get(key, 0) > 0:  # (retained again while the stream drained)
            return False
        ws = _STREAM_WS.pop(key, None)
        if ws is None:
            return False
        with torch.cuda.device(stream.device):
            _chk(lib.sbk_stream_workspace_release(c_void_p(stream.cuda_stream)), "sbk_stream_workspace_release")
    return True


def _stream(t: torch.Tensor):
    lib = load()
    if t.is_cuda:
        h = torch.cuda.current_stream(t.device).cuda_stream
        key = (id(lib), t.device.index, h)
        if key not in _STREAM_WS:
            _register_workspace(key, t.device, h)
        return c_void_p(h)
    key = (id(lib), "host", 0)  # (host tensors only reach a kernel under the test emulator: its one "stream" is NULL)
    if key not in _STREAM_WS:
        _register_workspace(key, t.device, None)
    return None


def _f32(t):
    if t.dtype != torch.float32:
        raise SbkError(f"expected float32, got {t.dtype}")
    return t


# ------------------------------------------------------------------ precision of the dense contractions
_tls = threading.local()


def precision() -> str:
    """"fp32" (parity path, default) or an opt-in reduced-precision operand type for the large GEMMs, all with fp32
    accumulation: "bf16" (also the bf16 attention kernel), "fp16", "fp8" (e4m3, per-tensor scales)."""
    return getattr(_tls, "precision", "fp32")


@contextlib.contextmanager
def precision_scope(p):
    """Per host thread (the batches in flight of ConcurrentTranscriber each carry their own)."""
    p = p or "fp32"
    if p not in ("fp32", "bf16", "fp16", "fp8"):
        raise NotImplementedError(f"precision {p!r}: 'fp32' (parity) or the opt-in 'bf16' / 'fp16' / 'fp8' GEMM operands")
    old = precision()
    _tls.precision = p
    try:
        yield
    finally:
        _tls.precision = old


# fp32 contractions with many rows (the encoder's) run on the bf16 matrix pipe through the exact three-way operand split
# (sbk_gemm_nt_f32x3, include/sbk.h): the same fp32 result class at 2-3x the rate of the fp32 MFMA kernels.  "0" keeps
# every contraction on v_mfma_f32_32x32x2_f32 (A/B runs, tests of the fp32-MFMA kernels).
F32X3 = os.environ.get("SBK_F32X3", "1") != "0"
F32X3_MIN_ROWS = int(os.environ.get("SBK_F32X3_MIN_ROWS", "2048"))
F32X3_MIN_TILES = int(os.environ.get("SBK_F32X3_MIN_TILES", "192"))
# ... and, with the activation operand pre-split as well, through sbk_gemm_nt_x3p (csrc/gemm_x3p.hip: 256-wide tiles)
X3P = os.environ.get("SBK_X3P", "1") != "0"
# the decode step's few-row projections on the bf16 matrix pipe (sbk_gemm_nt_x3r: panel images of the decoder's weights)
X3R = os.environ.get("SBK_X3R", "1") != "0"
X3P_MIN_TILES = int(os.environ.get("SBK_X3P_MIN_TILES", "96"))


def f32x3_ok(M: int, K: int, w: torch.Tensor) -> bool:
    """Shapes routed to sbk_gemm_nt_f32x3.  Measured on MI355X (tools/microbench.py --x3, profiles/r03_f32x3_sweep.log):
    from ~0.75 tiles of 128 x 128 per CU on the split-operand kernel wins (M = 8 000, N = 512: 38.9 vs 43.2 us); below, the
    fp32-MFMA tile kernels do (M = 4 032, N = 512: 25.4 vs 30.6 us)."""
    if not (F32X3 and M >= F32X3_MIN_ROWS and K % 32 == 0 and K >= 64 and w.dim() == 2 and w.is_contiguous()):
        return False
    return w.shape[0] % 4 == 0 and ((M + 127) // 128) * ((w.shape[0] + 127) // 128) >= F32X3_MIN_TILES


def _aligned16(*ts) -> bool:
    """(the split-operand kernel moves rows of C / residual / bias as 16-byte vectors)"""
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


# ------------------------------------------------------------------ ops
def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0, out=None,
            seq_len=None, rows_per_seq=0):
    """out[M,N] = residual + alpha * act(a[M,K] @ w[N,K]^T + bias).  `a` may have leading dims.

    With ``seq_len`` (int32 [batch]) rows are [batch][rows_per_seq] and the rows past each
    sequence's length contribute 0 before the residual is added."""
    lib = load()
    if isinstance(a, Panel):  # the producer wrote the operand's panel image (layernorm_x3p, gemm_nt_x3p(panel_out=True))
        return gemm_nt_x3p(a, w, bias, residual, act, alpha, out=out, seq_len=seq_len, rows_per_seq=rows_per_seq)
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    if precision() != "fp32" and M >= 256 and K % (16 if precision() == "fp8" else 8) == 0 and w.is_contiguous():
        return gemm_nt_bf16(a, w, bias, residual, act, alpha, seq_len, rows_per_seq, out=out, kind=precision())  # opt-in fast path
    _dev_ok(a2, w, bias, residual)
    _f32(a2), _f32(w)
    if out is None:
        out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _dev_ok(seq_len)
    if x3p_ok(M, K, w) and _aligned16(a2, out, bias, r2):  # both operands pre-split, 256-wide tiles (csrc/gemm_x3p.hip)
        return gemm_nt_x3p(split_x3p(a2), w, bias, r2, act, alpha, out=out, seq_len=seq_len, rows_per_seq=rows_per_seq)
    if f32x3_ok(M, K, w) and _aligned16(a2, out, bias, r2):
        _chk(lib.sbk_gemm_nt_f32x3(_p(a2), K, _p(lp_weight(w, "x3")), _p(bias), _p(r2), N, _p(out), N, M, N, K, act,
                                   float(alpha), _p(seq_len), int(rows_per_seq), _stream(a2)), "sbk_gemm_nt_f32x3")
        return out
    _chk(lib.sbk_gemm_nt_f32(_p(a2), K, _p(w), w.stride(0), _p(bias), _p(r2), N, _p(out), N, M, N, K, act,
                             float(alpha), _p(seq_len), int(rows_per_seq), _stream(a2)), "sbk_gemm_nt_f32")
    return out


def gemm_nt_rows(a_flat: torch.Tensor, M: int, K: int, lda: int, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE,
                 alpha=1.0, out=None):
    """gemm_nt whose A operand is M rows of K floats starting every ``lda`` floats in the flat tensor ``a_flat``.
    ``lda`` < K makes consecutive rows overlap: a strided window over a time-major signal, i.e. a 1-D convolution read
    in place (no im2col copy).  out [M,N] (contiguous)."""
    lib = load()
    N = w.shape[0]
    if M == 0:
        return out if out is not None else torch.empty(0, N, dtype=torch.float32, device=a_flat.device)
    if a_flat.dim() != 1 or a_flat.numel() < (M - 1) * lda + K or w.shape[1] != K:
        raise SbkError(f"gemm_nt_rows: {M} rows of {K} every {lda} do not fit {tuple(a_flat.shape)} / {tuple(w.shape)}")
    _dev_ok(a_flat, w, bias, residual, out)
    _f32(a_flat), _f32(w)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a_flat.device)
    # (precision "fp8" with fp8 activations: the sliding-window operand stays fp32 in memory, so the two convolutions of the
    # Whisper front-end take the bf16-operand kernel there too -- 2 % of the encoder's flops)
    if (precision() == "bf16" or (precision() == "fp8" and FP8_ACTIVATIONS)) and M >= 256 and K % 8 == 0 and lda % 4 == 0:
        _chk(lib.sbk_gemm_nt_bf16(_p(a_flat), int(lda), _p(bf16_weight(w)), K, _p(bias), _p(residual), N, _p(out), N, M, N, K,
                                  act, float(alpha), None, 0, _stream(a_flat)), "sbk_gemm_nt_bf16")
        return out
    if x3p_ok(M, K, w) and _aligned16(out, bias, residual):
        return gemm_nt_x3p(split_x3p(a_flat, rows=M, K=K, ldx=lda), w, bias, residual, act, alpha, out=out)
    if f32x3_ok(M, K, w) and lda % 4 == 0 and _aligned16(a_flat, out, bias, residual):
        _chk(lib.sbk_gemm_nt_f32x3(_p(a_flat), int(lda), _p(lp_weight(w, "x3")), _p(bias), _p(residual), N, _p(out), N, M, N, K,
                                   act, float(alpha), None, 0, _stream(a_flat)), "sbk_gemm_nt_f32x3")
        return out
    _chk(lib.sbk_gemm_nt_f32(_p(a_flat), int(lda), _p(w), w.stride(0), _p(bias), _p(residual), N, _p(out), N, M, N, K, act,
                             float(alpha), None, 0, _stream(a_flat)), "sbk_gemm_nt_f32")
    return out


# ------------------------------------------------------------------ both operands pre-split, panel layout (csrc/gemm_x3p.hip)
class Panel:
    """The panel image (include/sbk.h, sbk_split_x3p) of an fp32 matrix [rows, K]: the A operand of sbk_gemm_nt_x3p."""
    __slots__ = ("data", "rows", "K", "lead")

    def __init__(self, data, rows, K, lead=None):
        self.data, self.rows, self.K, self.lead = data, rows, K, lead  # lead: leading dims of the activation it stands for

    @property
    def device(self):
        return self.data.device

    # what a consumer written for the fp32 activation asks of its operand (RoPEMHA.core / the *_group cores: ADVICE r4)
    @property
    def shape(self):
        return torch.Size((tuple(self.lead) if self.lead is not None else (self.rows,)) + (self.K,))

    @property
    def dtype(self):
        return torch.float32

    def new_empty_rows(self) -> torch.Tensor:
        """An fp32 tensor of the shape of the activation this image stands for (``torch.empty_like`` of a tensor operand)."""
        return torch.empty(self.shape, dtype=torch.float32, device=self.data.device)


def panel_empty(rows: int, K: int, device, lead=None) -> Panel:
    n = ((rows + 63) // 64) * 64 * K * 3  # int16 elements
    return Panel(torch.empty(n, dtype=torch.int16, device=device), rows, K, lead)


def split_x3p(x: torch.Tensor, rows: Optional[int] = None, K: Optional[int] = None, ldx: Optional[int] = None) -> Panel:
    """x [..., K] fp32 (contiguous) -> its panel image; with rows / K / ldx: `rows` windows of K floats every ldx floats of
    the flat tensor x (gemm_nt_rows' sliding-window operand)."""
    lib = load()
    _dev_ok(x)
    _f32(x)
    lead = None
    if rows is None:
        K = x.shape[-1]
        lead = tuple(x.shape[:-1])
        x = x.reshape(-1, K)
        rows, ldx = x.shape[0], K
    out = panel_empty(rows, K, x.device, lead)
    _chk(lib.sbk_split_x3p(_p(x), int(ldx), _p(out.data), rows, K, _stream(x)), "sbk_split_x3p")
    return out


def panel_for(x: torch.Tensor, w: torch.Tensor) -> bool:
    """True when the contraction x[..., K] @ w^T takes the both-operands-pre-split route on the fp32 path, i.e. when the
    kernel that PRODUCES x should write it as a Panel (layernorm_x3p, gemm_nt_x3p(panel_out=True)) instead of fp32."""
    K = x.shape[-1] if not isinstance(x, Panel) else x.K
    M = (x.numel() // K) if not isinstance(x, Panel) else x.rows
    return precision() == "fp32" and x3p_ok(M, K, w)


def layernorm_x3p(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act=ACT_NONE) -> Panel:
    """``layernorm`` written directly as the panel image of its result (sbk_layernorm_x3p): the A operand of the
    gemm_nt that follows, without the fp32 round trip and the split pass."""
    lib = load()
    d = gamma.numel()
    x2 = x.reshape(-1, d)
    _dev_ok(x2, gamma, beta)
    _f32(x2)
    out = panel_empty(x2.shape[0], d, x.device, tuple(x.shape[:-1]))
    _chk(lib.sbk_layernorm_x3p(_p(x2), _p(gamma), _p(beta), _p(out.data), x2.shape[0], d, float(eps), act, _stream(x2)),
         "sbk_layernorm_x3p")
    return out


def x3p_ok(M: int, K: int, w: torch.Tensor) -> bool:
    """Shapes routed to sbk_gemm_nt_x3p: from F32X3_MIN_ROWS rows on, K % 16 == 0, and enough 256 x 128 tiles to give most
    CUs one (tools/microbench.py --x3p)."""
    if not (F32X3 and X3P and M >= F32X3_MIN_ROWS and K % 16 == 0 and K >= 32 and w.dim() == 2 and w.is_contiguous()):
        return False
    return w.shape[0] % 4 == 0 and ((M + 255) // 256) * ((w.shape[0] + 127) // 128) >= X3P_MIN_TILES


def gemm_nt_x3p(a: Panel, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0, out=None, seq_len=None,
                rows_per_seq=0, panel_out=False, fp32_out=True):
    """out[M,N] = residual + alpha * act(A @ w^T + bias) with A given as its panel image.  ``panel_out``: also (or, with
    ``fp32_out=False``, only) return the result as a Panel -- the A operand of the next contraction."""
    lib = load()
    M, K, N = a.rows, a.K, w.shape[0]
    _dev_ok(a.data, w, bias, residual, seq_len)
    wp = lp_weight(w, "x3p")
    lead = a.lead if a.lead is not None else (M,)
    if fp32_out and out is None:
        out = torch.empty(*lead, N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    pc = panel_empty(M, N, a.device, lead) if panel_out else None
    # The kernel moves rows of C / residual / bias as 16-byte vectors.  A tensor operand falls back to another kernel in
    # gemm_nt; a Panel operand has no fp32 form to fall back with, so unaligned views go through aligned temporaries
    # (fresh torch allocations are 512-byte aligned) -- ADVICE r4
    user_out = None
    if bias is not None and bias.data_ptr() % 16:
        bias = bias.clone()
    if r2 is not None and r2.data_ptr() % 16:
        r2 = r2.clone()
    if fp32_out and out.data_ptr() % 16:
        user_out, out = out, torch.empty(out.shape, dtype=torch.float32, device=out.device)
    _chk(lib.sbk_gemm_nt_x3p(_p(a.data), _p(wp), _p(bias), _p(r2), N, _p(out) if fp32_out else None, N,
                             _p(pc.data) if pc is not None else None, M, N, K, act, float(alpha), _p(seq_len), int(rows_per_seq),
                             _stream(a.data)), "sbk_gemm_nt_x3p")
    if user_out is not None:
        user_out.copy_(out)
        out = user_out
    if panel_out:
        return (out, pc) if fp32_out else pc
    return out


def gemm_nt_x3r(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0):
    """out[M,N] = residual + alpha * act(a @ w^T + bias) for FEW rows (a decoding step's projections) on the bf16 matrix
    pipe (sbk_gemm_nt_x3r): ``a`` fp32 (split in registers), ``w`` is used through its cached panel image."""
    lib = load()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    _dev_ok(a2, w, bias, residual)
    _f32(a2)
    out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _chk(lib.sbk_gemm_nt_x3r(_p(a2), K, _p(lp_weight(w, "x3p")), _p(bias), _p(r2), N, _p(out), N, M, N, K, act, float(alpha),
                             _stream(w)), "sbk_gemm_nt_x3r")
    return out


def gemm_ln_nt_x3r(a: torch.Tensor, wf: torch.Tensor, bf, eps, residual=None, act=ACT_NONE, alpha=1.0):
    """residual + alpha * act(LN(a) @ W^T + b) for FEW rows on the bf16 matrix pipe, the LayerNorm in the projection's
    prologue (sbk_gemm_ln_nt_x3r): gamma / beta pre-folded into (wf, bf) as for ``gemm_ln_nt``; ``wf`` is used through its
    cached panel image.  K = 256, 512, 1024 or 1280."""
    lib = load()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], wf.shape[0]
    _dev_ok(a2, wf, bf, residual)
    _f32(a2)
    out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _chk(lib.sbk_gemm_ln_nt_x3r(_p(a2), K, _p(lp_weight(wf, "x3p")), _p(bf), _p(r2), N, _p(out), N, M, N, K,
                                float(eps), act, float(alpha), _stream(wf)), "sbk_gemm_ln_nt_x3r")
    return out


class _Ready:
    """A derived weight image (bf16 / x3 copy, LayerNorm-folded projection, a handle's tables) is produced by kernels on
    the stream of the thread that first asked for it; the batches in flight run on other streams and may ask for it a
    few microseconds later.  The producer records an event; a consumer on another stream waits for it (on the
    device, not on the host) until it has completed once."""
    __slots__ = ("ev", "sid")

    def __init__(self, device):
        self.ev, self.sid = None, 0
        if device.type == "cuda":
            cur = torch.cuda.current_stream(device)
            self.ev = torch.cuda.Event()
            self.ev.record(cur)
            self.sid = cur.cuda_stream

    def wait(self, device):
        ev = self.ev
        if ev is None:
            return
        if ev.query():
            self.ev = None  # (benign race: the event has completed for everybody)
            return
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream != self.sid:
            cur.wait_event(ev)


_BF16_WEIGHTS = {}  # (id(base tensor), kind, view geometry) -> (weakref to the base, _version, image, scale, _Ready, data_ptr)
# re-entrant: the weak-reference callback below takes the lock too, and a cyclic-GC pass started by an allocation INSIDE a
# locked region may run it on the thread that already holds the lock
_BF16_LOCK = threading.RLock()


def lp_weight(w: torch.Tensor, kind: str = "bf16"):
    """The reduced-precision image of a weight matrix ("bf16" / "fp16": int16 bits; "fp8": (uint8 e4m3 bits, w_scale);
    "x3": the exact three-piece bf16 split [N, K/32, 3, 32] of sbk_split_bf16x3 -- not reduced: the pieces sum to w),
    cached for the lifetime of THAT tensor object: the entry holds a weak reference to its source and is used only while
    ``ref() is w`` and the version counter is unchanged (a freed model's addresses are commonly handed to the next
    model of the same shapes by the caching allocator, and load_state_dict leaves ``_version`` alike -- a (data_ptr,
    version, shape) key would then serve the OLD model's weights).  Entries die with their tensor (weakref callback);
    insert / evict under a lock (worker threads)."""
    # a VIEW of a parameter (conv weight [2d,d,1] reshaped to [2d,d], a slice of a stacked projection) is a new tensor object
    # on every call: the entry belongs to the view's base and the view's geometry, so the image is made once
    base = w._base if w._base is not None else w
    key = (id(base), kind, w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    with _BF16_LOCK:
        hit = _BF16_WEIGHTS.get(key)
        # (data_ptr: `param.data = other` -- nn.Module._apply, a manual weight swap -- keeps the object and its version)
        if (hit is not None and hit[0]() is base and hit[1] == w._version and hit[2].device == w.device
                and hit[5] == w.data_ptr()):
            hit[4].wait(w.device)  # (made on another worker's stream a moment ago?)
            return hit[2] if kind not in ("fp8", "fp8r") else (hit[2], hit[3])
    lib = load()
    w2 = w.detach().contiguous()
    _dev_ok(w2)
    _f32(w2)
    scale = None
    if kind == "fp8":
        scale = max(float(w2.abs().max()), 1e-30) / 448.0  # (once per weight: a host round trip at load time)
        out = torch.empty(w2.shape, dtype=torch.uint8, device=w.device)
        _chk(lib.sbk_f32_to_fp8(_p(w2), _p(out), w2.numel(), 1.0 / scale, _stream(w2)), "sbk_f32_to_fp8")
    elif kind == "fp8r":  # e4m3 with one scale per output channel (sbk_quant_rows_fp8): the W operand of gemm_nt_fp8a
        out = torch.empty(w2.shape, dtype=torch.uint8, device=w.device)
        scale = torch.empty(w2.shape[0], dtype=torch.float32, device=w.device)
        _chk(lib.sbk_quant_rows_fp8(_p(w2), w2.shape[1], _p(out), _p(scale), w2.shape[0], w2.shape[1], _stream(w2)), "sbk_quant_rows_fp8")
    elif kind == "x3":
        N, K = w2.shape
        out = torch.empty(N, K // 32, 3, 32, dtype=torch.int16, device=w.device)
        _chk(lib.sbk_split_bf16x3(_p(w2), K, _p(out), N, K, _stream(w2)), "sbk_split_bf16x3")
    elif kind == "x3p":
        N, K = w2.shape
        out = torch.empty(((N + 63) // 64) * 64 * K * 3, dtype=torch.int16, device=w.device)
        _chk(lib.sbk_split_x3p(_p(w2), K, _p(out), N, K, _stream(w2)), "sbk_split_x3p")
    else:
        out = torch.empty(w2.shape, dtype=torch.int16, device=w.device)
        fn = lib.sbk_f32_to_bf16 if kind == "bf16" else lib.sbk_f32_to_f16
        _chk(fn(_p(w2), _p(out), w2.numel(), _stream(w2)), "sbk_f32_to_" + kind)

    def _drop(_ref, key=key):
        with _BF16_LOCK:
            cur = _BF16_WEIGHTS.get(key)
            if cur is not None and cur[0] is _ref:
                del _BF16_WEIGHTS[key]

    with _BF16_LOCK:
        _BF16_WEIGHTS[key] = (weakref.ref(base, _drop), w._version, out, scale, _Ready(w.device), w.data_ptr())
    return out if kind not in ("fp8", "fp8r") else (out, scale)


def bf16_weight(w: torch.Tensor) -> torch.Tensor:
    return lp_weight(w, "bf16")


def gemm_nt_bf16(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0, seq_len=None,
                 rows_per_seq=0, out=None, kind="bf16"):
    """gemm_nt with reduced-precision operands / fp32 accumulation (opt-in fast path; `w` is the fp32 parameter, its
    image of ``kind`` -- "bf16", "fp16" or "fp8" (e4m3 with per-tensor scales: the weight's at conversion time, the
    activation's max |a| computed on the device per call) -- is cached).  Falls back to the fp32 kernel for shapes the
    kernels do not take (K % 8 != 0; fp8: K % 16 != 0)."""
    K = a.shape[-1]
    if K % (16 if kind == "fp8" else 8) != 0:
        with precision_scope("fp32"):
            return gemm_nt(a, w, bias, residual, act, alpha, out=out, seq_len=seq_len, rows_per_seq=rows_per_seq)
    lib = load()
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    _dev_ok(a2, bias, residual, seq_len)
    _f32(a2)
    if out is None:
        out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    if kind == "fp8":
        wq, w_scale = lp_weight(w, "fp8")
        amax = torch.empty(1, dtype=torch.float32, device=a.device)
        _chk(lib.sbk_absmax_f32(_p(a2), a2.numel(), _p(amax), _stream(a2)), "sbk_absmax_f32")
        _chk(lib.sbk_gemm_nt_fp8(_p(a2), K, _p(amax), _p(wq), K, float(w_scale), _p(bias), _p(r2), N, _p(out), N, M, N, K,
                                 act, float(alpha), _p(seq_len), int(rows_per_seq), _stream(a2)), "sbk_gemm_nt_fp8")
        return out
    wb = lp_weight(w, kind)
    fn = lib.sbk_gemm_nt_bf16 if kind == "bf16" else lib.sbk_gemm_nt_f16
    _chk(fn(_p(a2), K, _p(wb), K, _p(bias), _p(r2), N, _p(out), N, M, N, K, act, float(alpha),
            _p(seq_len), int(rows_per_seq), _stream(a2)), "sbk_gemm_nt_" + kind)
    return out


# precision "bf16": keep the operands of consecutive contractions in bf16 in memory where a model's forward supports it
# (the Whisper encoder); False = every contraction reads fp32 activations and rounds them on load (A/B, tests)
BF16_ACTIVATIONS = os.environ.get("SBK_BF16_ACTIVATIONS", "1") != "0"
# rows from which the Conformer feed-forward pairs take that path.  Measured on MI355X: one stream, 32 x 20 s (16 000 rows)
# encodes in 15.7 instead of 17.0 ms, 32 x 10 s in 9.4 instead of 8.9 (tools/microbench.py --enc-bf16,
# profiles/r03_bf16_attention_lds_and_conformer_bf16.log); the eight-worker job of bench.py --precision bf16 runs at
# 12 093 audio-s/s with every batch on this path, 11 752 from 16 000 rows, 11 887 without (profiles/r03_bf16_bench_leg_ab.log)
BF16A_MIN_ROWS = int(os.environ.get("SBK_BF16A_MIN_ROWS", "256"))


def bf16a_ok(K: int) -> bool:
    """Shapes the bf16-activation contraction (sbk_gemm_nt_bf16a) takes: K a multiple of its 64-deep K tile."""
    return K % 64 == 0


def gemm_nt_bf16a(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0,
                  out_dtype=torch.float32):
    """epilogue(a . w^T) with ``a`` ALREADY bf16 in memory (torch.bfloat16 [..., K], written by layernorm_bf16 /
    rope_attention(out_dtype=bf16) / a previous gemm_nt_bf16a) and the cached bf16 image of the fp32 parameter ``w``;
    fp32 accumulation, fp32 residual; the result is fp32 or -- ``out_dtype=torch.bfloat16`` -- the next contraction's
    bf16 operand."""
    lib = load()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    _dev_ok(a2, bias, residual)
    if a2.dtype != torch.bfloat16 or not a2.is_contiguous():
        raise SbkError("gemm_nt_bf16a: the activation operand must be a contiguous torch.bfloat16 tensor")
    wb = lp_weight(w, "bf16")
    out = torch.empty(*a.shape[:-1], N, dtype=out_dtype, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    c32, cb = (out, None) if out_dtype == torch.float32 else (None, out)
    _chk(lib.sbk_gemm_nt_bf16a(_p(a2), K, _p(wb), K, _p(bias), _p(r2), N, _p(c32), N, _p(cb), N, M, N, K, act,
                               float(alpha), _stream(a2)), "sbk_gemm_nt_bf16a")
    return out


def layernorm_bf16(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act=ACT_NONE):
    """``layernorm`` written as bf16 (round to nearest even): the operand of a following gemm_nt_bf16a."""
    lib = load()
    d = gamma.numel()
    x2 = x.reshape(-1, d)
    _dev_ok(x2, gamma, beta)
    _f32(x2)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _chk(lib.sbk_layernorm_bf16o(_p(x2), _p(gamma), _p(beta), _p(out), x2.shape[0], d, float(eps), act, _stream(x2)),
         "sbk_layernorm_bf16o")
    return out


# precision "fp8": fp8 (e4m3) activations between the contractions of the Whisper encoder, one fp32 scale per row, on the
# 2 x-rate fp8 matrix instruction (sbk_gemm_nt_fp8a); False = the round-3 path (fp32 activations, per-tensor scales)
FP8_ACTIVATIONS = os.environ.get("SBK_FP8_ACTIVATIONS", "1") != "0"


class Fp8Rows:
    """An activation as e4m3 bytes with one fp32 scale per row (scale None: 1): value = scale[row] * e4m3."""
    __slots__ = ("q", "scale")

    def __init__(self, q, scale):
        self.q, self.scale = q, scale

    @property
    def shape(self):
        return self.q.shape


def fp8a_ok(K: int) -> bool:
    """Shapes the fp8-activation contraction (sbk_gemm_nt_fp8a) takes: K a multiple of its 128-deep K tile."""
    return K % 128 == 0


def layernorm_fp8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act=ACT_NONE) -> Fp8Rows:
    """``layernorm`` written as e4m3 rows with one scale per row (row maximum / 448): the operand of gemm_nt_fp8a."""
    lib = load()
    d = gamma.numel()
    x2 = x.reshape(-1, d)
    _dev_ok(x2, gamma, beta)
    _f32(x2)
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    _chk(lib.sbk_layernorm_fp8o(_p(x2), _p(gamma), _p(beta), _p(q), _p(scale), x2.shape[0], d, float(eps), act, _stream(x2)),
         "sbk_layernorm_fp8o")
    return Fp8Rows(q, scale)


def quant_rows_fp8(x: torch.Tensor) -> Fp8Rows:
    """x [..., d] fp32 or bf16 -> e4m3 rows with one scale per row."""
    lib = load()
    d = x.shape[-1]
    x2 = x.reshape(-1, d).contiguous()
    _dev_ok(x2)
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    if x2.dtype == torch.bfloat16:
        _chk(lib.sbk_quant_rows_bf16_fp8(_p(x2), d, _p(q), _p(scale), x2.shape[0], d, _stream(x2)), "sbk_quant_rows_bf16_fp8")
    else:
        _f32(x2)
        _chk(lib.sbk_quant_rows_fp8(_p(x2), d, _p(q), _p(scale), x2.shape[0], d, _stream(x2)), "sbk_quant_rows_fp8")
    return Fp8Rows(q, scale)


def gemm_nt_fp8a(a: Fp8Rows, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0, out_dtype=torch.float32):
    """epilogue(a . w^T) with ``a`` ALREADY fp8 in memory (layernorm_fp8 / quant_rows_fp8 / a previous call's fp8 output) and
    the cached per-channel-scaled e4m3 image of the fp32 parameter ``w``; fp32 accumulation, fp32 residual.  The result is
    fp32, torch.bfloat16 (the attention kernel's operand) or -- ``out_dtype="fp8"`` -- e4m3 at scale 1 (an Fp8Rows
    without scales: the hidden layer of a feed-forward pair)."""
    lib = load()
    K = a.q.shape[-1]
    a2 = a.q.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    _dev_ok(a2, a.scale, bias, residual)
    wq, w_scale = lp_weight(w, "fp8r")
    r2 = residual.reshape(-1, N) if residual is not None else None
    lead = tuple(a.q.shape[:-1])
    c = cb = c8 = None
    if out_dtype == "fp8":
        c8 = torch.empty(*lead, N, dtype=torch.uint8, device=a2.device)
    elif out_dtype == torch.bfloat16:
        cb = torch.empty(*lead, N, dtype=torch.bfloat16, device=a2.device)
    else:
        c = torch.empty(*lead, N, dtype=torch.float32, device=a2.device)
    _chk(lib.sbk_gemm_nt_fp8a(_p(a2), K, _p(a.scale), _p(wq), K, _p(w_scale), _p(bias), _p(r2), N, _p(c), N, _p(cb), N,
                              _p(c8), N, 1.0, M, N, K, act, float(alpha), _stream(a2)), "sbk_gemm_nt_fp8a")
    if c8 is not None:
        return Fp8Rows(c8, None)
    return cb if cb is not None else c


def gemm_nt_splitk(a, w, bias=None, residual=None, act=ACT_NONE, alpha=1.0, slices=8):
    """gemm_nt for few-row operands with a caller-provided split-K workspace."""
    lib = load()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    _dev_ok(a2, w, bias, residual)
    out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    ws = torch.empty(max(1, slices * M * N), dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _chk(lib.sbk_gemm_nt_splitk_f32(_p(a2), K, _p(w), w.stride(0), _p(bias), _p(r2), N, _p(out), N, M, N, K, act,
                                    float(alpha), _p(ws), ws.numel(), _stream(a2)), "sbk_gemm_nt_splitk_f32")
    return out


def gemm_ln_nt(a, wf, bf, eps, residual=None, act=ACT_NONE, alpha=1.0):
    """LN(a) @ W^T + b with gamma/beta pre-folded into (wf, bf) (see include/sbk.h)."""
    lib = load()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], wf.shape[0]
    _dev_ok(a2, wf, bf, residual)
    out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _chk(lib.sbk_gemm_ln_nt_f32(_p(a2), K, _p(wf), K, _p(bf), _p(r2), N, _p(out), N, M, N, K, float(eps), act,
                                float(alpha), _stream(a2)), "sbk_gemm_ln_nt_f32")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act=ACT_NONE, out=None):
    """LayerNorm over the trailing gamma.numel() elements of every row."""
    lib = load()
    d = gamma.numel()
    x2 = x.reshape(-1, d)
    _dev_ok(x2, gamma, beta)
    _f32(x2)
    if out is None:
        out = torch.empty_like(x)
    _chk(lib.sbk_layernorm_f32(_p(x2), _p(gamma), _p(beta), _p(out), x2.shape[0], d, float(eps), act, _stream(x2)),
         "sbk_layernorm_f32")
    return out


def input_norm_stats(x, n_valid, per_batch, std_norm, eps, avoid_padding_norm=False):
    """x [B,T,C] normalised with the mean / std of each utterance's valid frames (per_batch False: norm_type
    "sentence") or of all valid frames of the batch ("batch"); n_valid int32 [B]."""
    lib = load()
    _dev_ok(x, n_valid)
    _f32(x)
    B, T, C = x.shape
    out = torch.empty_like(x)
    ws = torch.empty(max(1, lib.sbk_input_norm_stats_workspace_bytes(B, C) // 4), dtype=torch.float32, device=x.device)
    _chk(lib.sbk_input_norm_stats_f32(_p(x), _p(n_valid), _p(out), _p(ws), B, T, C, int(bool(per_batch)),
                                      int(bool(std_norm)), float(eps), int(bool(avoid_padding_norm)), _stream(x)),
         "sbk_input_norm_stats_f32")
    return out


def input_norm_global(x, mean, std, eps):
    lib = load()
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    _dev_ok(x2, mean, std)
    out = torch.empty_like(x)
    _chk(lib.sbk_input_norm_global_f32(_p(x2), _p(mean), _p(std), _p(out), x2.shape[0], C, float(eps), _stream(x2)),
         "sbk_input_norm_global_f32")
    return out


def input_norm_global_masked(x, mean, std, n_valid, eps):
    """``input_norm_global`` with avoid_padding_norm: x [B,T,C], the frames t >= n_valid[b] pass through unchanged."""
    lib = load()
    B, T, C = x.shape
    _dev_ok(x, mean, std, n_valid)
    _f32(x)
    out = torch.empty_like(x)
    _chk(lib.sbk_input_norm_global_masked_f32(_p(x), _p(mean), _p(std), _p(n_valid), _p(out), B, T, C, float(eps), _stream(x)),
         "sbk_input_norm_global_masked_f32")
    return out


def pcm16_to_f32(pcm: torch.Tensor, channels: int = 1, out=None):
    """int16 PCM (any shape; interleaved channels last when channels > 1) -> float32 sample / 32768, channel mean."""
    lib = load()
    _dev_ok(pcm)
    if pcm.dtype != torch.int16:
        raise SbkError(f"expected int16 PCM, got {pcm.dtype}")
    shape = pcm.shape if channels == 1 else pcm.shape[:-1]
    frames = pcm.numel() // channels
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=pcm.device)
    _chk(lib.sbk_pcm16_to_f32(_p(pcm), _p(out), frames, int(channels), _stream(pcm)), "sbk_pcm16_to_f32")
    return out


def fbank(wav, window, twiddle, radices, mel_w, mel_ptr, mel_bin, n_fft, hop, n_mels, amin, top_db,
          norm_mean=None, norm_std=None, norm_eps=1e-10):
    """[B,N] waveforms -> [B,T,n_mels] log-mel features (see include/sbk.h)."""
    lib = load()
    _dev_ok(wav, window, twiddle, mel_w, mel_ptr, mel_bin, norm_mean, norm_std)
    _f32(wav)
    B, N = wav.shape
    T = 1 + N // hop
    out = torch.empty(B, T, n_mels, dtype=torch.float32, device=wav.device)
    tile_max = torch.empty(B, (T + 3) // 4, dtype=torch.float32, device=wav.device)
    rad = (c_int32 * len(radices))(*radices)
    _chk(lib.sbk_fbank_f32(_p(wav), _p(window), _p(twiddle), rad, len(radices), _p(mel_w), _p(mel_ptr), _p(mel_bin),
                           _p(out), _p(tile_max), B, N, n_fft, hop, n_mels, mel_w.numel(), float(amin), float(top_db),
                           _p(norm_mean), _p(norm_std), float(norm_eps), _stream(wav)), "sbk_fbank_f32")
    return out


def whisper_log_mel(wav, window, twiddle, radices, mel_w, mel_ptr, mel_bin, n_fft, hop, n_mels):
    """[B,N] waveforms -> Whisper log-mel [B, n_mels, N // hop] (see include/sbk.h)."""
    lib = load()
    _dev_ok(wav, window, twiddle, mel_w, mel_ptr, mel_bin)
    _f32(wav)
    B, N = wav.shape
    T = N // hop
    tmp = torch.empty(B, T, n_mels, dtype=torch.float32, device=wav.device)
    tile_max = torch.empty(B * ((T + 3) // 4), dtype=torch.float32, device=wav.device)
    out = torch.empty(B, n_mels, T, dtype=torch.float32, device=wav.device)
    rad = (c_int32 * len(radices))(*radices)
    _chk(lib.sbk_whisper_log_mel_f32(_p(wav), _p(window), _p(twiddle), rad, len(radices), _p(mel_w), _p(mel_ptr),
                                     _p(mel_bin), _p(tmp), _p(tile_max), _p(out), B, N, n_fft, hop, n_mels,
                                     mel_w.numel(), _stream(wav)), "sbk_whisper_log_mel_f32")
    return out


def stft(wav, window, twiddle, radices, n_fft, hop):
    """[B,N] -> complex STFT [B,T,n_fft/2+1,2]."""
    lib = load()
    _dev_ok(wav, window, twiddle)
    _f32(wav)
    B, N = wav.shape
    T = 1 + N // hop
    spec = torch.empty(B, T, n_fft // 2 + 1, 2, dtype=torch.float32, device=wav.device)
    rad = (c_int32 * len(radices))(*radices)
    _chk(lib.sbk_stft_f32(_p(wav), _p(window), _p(twiddle), rad, len(radices), _p(spec), B, N, n_fft, hop, _stream(wav)),
         "sbk_stft_f32")
    return spec


def spectral_magnitude(stft_t, power=1.0, log=False, eps=1e-14):
    lib = load()
    _dev_ok(stft_t)
    _f32(stft_t)
    out = torch.empty(stft_t.shape[:-1], dtype=torch.float32, device=stft_t.device)
    _chk(lib.sbk_spectral_magnitude_f32(_p(stft_t), _p(out), out.numel(), float(power), int(bool(log)), float(eps),
                                        _stream(stft_t)), "sbk_spectral_magnitude_f32")
    return out


def amplitude_to_db(x, multiplier, amin, db_offset, top_db):
    """In place on x [B, ...]: dB conversion + per-utterance floor."""
    lib = load()
    _dev_ok(x)
    _f32(x)
    B = x.shape[0]
    tile_max = torch.empty(B, 64, dtype=torch.float32, device=x.device)
    _chk(lib.sbk_amplitude_to_db_f32(_p(x), _p(tile_max), B, x[0].numel(), float(multiplier), float(amin),
                                     float(db_offset), float(top_db), _stream(x)), "sbk_amplitude_to_db_f32")
    return x


def conv_block(x, wt, bias, gamma, beta, cout, eps=1e-5, slope=0.01):
    """One ConvolutionFrontEnd block: x [B,T,F,Cin] -> [B,T',F',Cout] (see include/sbk.h)."""
    lib = load()
    _dev_ok(x, wt, bias, gamma, beta)
    _f32(x)
    B, Tin, Fin, Cin = x.shape
    Tout, Fout = (Tin - 1) // 2 + 1, (Fin - 1) // 2 + 1
    y = torch.empty(B, Tout, Fout, cout, dtype=torch.float32, device=x.device)
    _chk(lib.sbk_conv_block_f32(_p(x), _p(wt), _p(bias), _p(gamma), _p(beta), _p(y), B, Tin, Fin, Cin, cout,
                                float(eps), float(slope), _stream(x)), "sbk_conv_block_f32")
    return y


def relpos_attention(qkv, pos, bias_u, bias_v, key_len, H, scale, want_attn=False, chunk_size=0, left_chunks=-1,
                     out=None):
    """qkv [B,T,3*d] (per-head interleaved), pos [2T-1,d] -> context [B,T,d] (+ weights [B,H,T,T]).
    ``chunk_size`` > 0: Dynamic Chunk mask (``left_chunks`` < 0 = unlimited left context).  ``out``: optional
    contiguous [B,T,d] destination."""
    lib = load()
    _dev_ok(qkv, pos, bias_u, bias_v, key_len, out)
    _f32(qkv)
    B, T, d3 = qkv.shape
    d = d3 // 3
    if out is None:
        out = torch.empty(B, T, d, dtype=torch.float32, device=qkv.device)
    attn = torch.empty(B, H, T, T, dtype=torch.float32, device=qkv.device) if want_attn else None
    _chk(lib.sbk_relpos_attention_f32(_p(qkv), _p(pos), _p(bias_u), _p(bias_v), _p(key_len), _p(out), _p(attn), B, T,
                                      H, d // H, float(scale), int(chunk_size), int(left_chunks), _stream(qkv)),
         "sbk_relpos_attention_f32")
    return out, attn


def rope_attention(qkv, cosines, sines, key_len, H, scale, want_attn=False, chunk_size=0, left_chunks=-1, out=None,
                   out_dtype=torch.float32):
    """qkv [B,T,3*d] (per-head interleaved), cosines / sines [rows >= T, Dh] -> context [B,T,d].  ``cosines`` =
    ``sines`` = None: plain scaled-dot-product attention (no rotation; no attention-weights output)."""
    lib = load()
    _dev_ok(qkv, cosines, sines, key_len, out)
    _f32(qkv)
    B, T, d3 = qkv.shape
    d = d3 // 3
    rows = cosines.shape[0] if cosines is not None else 0
    if out_dtype == torch.bfloat16:  # bf16 operands AND a bf16 context (the operand of the output projection)
        if want_attn or d // H != 64 or out is not None:
            raise SbkError("rope_attention: the bf16 context output needs head_dim 64 and no attention-weights output")
        out = torch.empty(B, T, d, dtype=torch.bfloat16, device=qkv.device)
        _chk(lib.sbk_rope_attention_bf16o(_p(qkv), _p(cosines), _p(sines), _p(key_len), _p(out), B, T, H, d // H,
                                          rows, float(scale), int(chunk_size), int(left_chunks), _stream(qkv)),
             "sbk_rope_attention_bf16o")
        return out, None
    if out is None:
        out = torch.empty(B, T, d, dtype=torch.float32, device=qkv.device)
    if precision() == "bf16" and not want_attn and d // H == 64:  # opt-in: bf16 operands on the matrix cores
        _chk(lib.sbk_rope_attention_bf16(_p(qkv), _p(cosines), _p(sines), _p(key_len), _p(out), B, T, H, d // H,
                                         rows, float(scale), int(chunk_size), int(left_chunks), _stream(qkv)),
             "sbk_rope_attention_bf16")
        return out, None
    attn = torch.empty(B, H, T, T, dtype=torch.float32, device=qkv.device) if want_attn else None
    _chk(lib.sbk_rope_attention_f32(_p(qkv), _p(cosines), _p(sines), _p(key_len), _p(out), _p(attn), B, T, H, d // H,
                                    rows, float(scale), int(chunk_size), int(left_chunks), _stream(qkv)),
         "sbk_rope_attention_f32")
    return out, attn


def attention_bf16(qkv: torch.Tensor, key_len, H: int, scale: float):
    """Plain attention on bf16 rows: qkv [B,T,3*d] torch.bfloat16 (per-head interleaved, written by gemm_nt_bf16a) ->
    context [B,T,d] torch.bfloat16 (head_dim 64)."""
    lib = load()
    _dev_ok(qkv, key_len)
    B, T, d3 = qkv.shape
    d = d3 // 3
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous() or d // H != 64:
        raise SbkError("attention_bf16: contiguous torch.bfloat16 qkv rows with head_dim 64")
    out = torch.empty(B, T, d, dtype=torch.bfloat16, device=qkv.device)
    ws = torch.empty(lib.sbk_attention_bf16io_workspace_bytes(B, T, H) // 2, dtype=torch.bfloat16, device=qkv.device)
    _chk(lib.sbk_attention_bf16io(_p(qkv), _p(key_len), _p(out), _p(ws), B, T, H, d // H, float(scale), _stream(qkv)),
         "sbk_attention_bf16io")
    return out


def glu_dwconv(h, w, bias, ksize, chunk_size=0, out=None):
    """h [B,T,2d] -> depthwise_conv(GLU(h)) [B,T,d]; w [d,ksize]; ``chunk_size`` > 0: Dynamic Chunk Convolution."""
    lib = load()
    _dev_ok(h, w, bias, out)
    _f32(h)
    B, T, d2 = h.shape
    y = out if out is not None else torch.empty(B, T, d2 // 2, dtype=torch.float32, device=h.device)
    _chk(lib.sbk_glu_dwconv_f32(_p(h), _p(w), _p(bias), _p(y), B, T, d2 // 2, int(ksize), int(chunk_size), _stream(h)),
         "sbk_glu_dwconv_f32")
    return y


# ------------------------------------------------------------------ decoder / searchers
class DecoderHandle:
    """Device pointers of a TransformerASR decoder (+ seq_lin) laid out as sbk_decoder_weights.
    Holds references to the parameter tensors so the pointers stay valid."""

    def __init__(self, model, seq_lin=None, fold=True):
        dec = model.decoder
        layers = []
        for L in dec.layers:
            if not L.normalize_before:
                raise NotImplementedError("post-norm decoder layers are not on the Conformer ASR path")
            layers.append(dict(
                ln1=(L.norm1.norm.weight, L.norm1.norm.bias), sa_in=(L.self_attn.att.in_proj_weight, L.self_attn.att.in_proj_bias),
                sa_out=(L.self_attn.att.out_proj.weight, L.self_attn.att.out_proj.bias),
                ln2=(L.norm2.norm.weight, L.norm2.norm.bias),
                ca_in=(L.multihead_attn.att.in_proj_weight, L.multihead_attn.att.in_proj_bias),
                ca_out=(L.multihead_attn.att.out_proj.weight, L.multihead_attn.att.out_proj.bias),
                ln3=(L.norm3.norm.weight, L.norm3.norm.bias), ff1=(L.pos_ffn.ffn[0].weight, L.pos_ffn.ffn[0].bias),
                ff2=(L.pos_ffn.ffn[3].weight, L.pos_ffn.ffn[3].bias)))
        pe = model.positional_encoding_decoder.pe
        self._build(layers, emb=model.custom_tgt_module.layers[0].emb.Embedding.weight,
                    pe=pe.reshape(pe.shape[-2], pe.shape[-1]), final_ln=(dec.norm.norm.weight, dec.norm.norm.bias),
                    seq=(seq_lin.w.weight, seq_lin.w.bias) if seq_lin is not None else None, nhead=dec.layers[0].nhead,
                    ffn_act=dec.layers[-1].pos_ffn.act_code, ln_eps=dec.norm.eps, emb_scale=0.0, fold=fold)
        self.key = self.source_key(model, seq_lin)

    @classmethod
    def from_tensors(cls, layers, emb, pe, final_ln, seq, nhead, ffn_act, ln_eps, emb_scale, fold=True, key=None):
        """A pre-norm Transformer decoder given as plain tensors: ``layers`` = dicts with ln1 / sa_in / sa_out / ln2 /
        ca_in / ca_out / ln3 / ff1 / ff2 -> (weight, bias) (in_proj stacked [3d,d] as torch.nn.MultiheadAttention);
        ``emb_scale`` multiplies the token embedding (0 = sqrt(d_model)); ``seq`` = output projection (weight, bias)."""
        self = cls.__new__(cls)
        self._build(layers, emb, pe, final_ln, seq, nhead, ffn_act, ln_eps, emb_scale, fold)
        self.key = key
        return self

    def _build(self, layer_specs, emb, pe, final_ln, seq, nhead, ffn_act, ln_eps, emb_scale, fold):
        self.keep = []
        fold = fold and final_ln[0].shape[0] in (128, 256, 512)  # shapes the fused kernel takes
        layers = (DecoderLayer * len(layer_specs))()

        def ptr(t):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            _dev_ok(t)
            _f32(t)
            self.keep.append(t)
            return t.data_ptr()

        dm = emb.shape[1]

        def split3(w):  # sbk_split_bf16x3's image (the big contractions of the search run on the bf16 matrix pipe)
            if not F32X3 or w.shape[1] % 32 != 0 or w.shape[1] < 64 or w.shape[0] % 4 != 0:
                return None
            w2 = w.detach().contiguous()
            _dev_ok(w2)
            _f32(w2)
            out = torch.empty(w2.shape[0], w2.shape[1] // 32, 3, 32, dtype=torch.int16, device=w2.device)
            _chk(load().sbk_split_bf16x3(_p(w2), w2.shape[1], _p(out), w2.shape[0], w2.shape[1], _stream(w2)), "sbk_split_bf16x3")
            self.keep.append(out)
            return out.data_ptr()

        def panel(w):  # sbk_split_x3p's image: the decode step's projections as sbk_gemm_nt_x3r (csrc/gemm.hip)
            if not (F32X3 and X3R) or w.shape[1] % 256 != 0 or w.shape[0] % 4 != 0:
                return None
            w2 = w.detach().contiguous()
            _dev_ok(w2)
            _f32(w2)
            N, K = w2.shape
            out = torch.empty(((N + 63) // 64) * 64 * K * 3, dtype=torch.int16, device=w2.device)
            _chk(load().sbk_split_x3p(_p(w2), K, _p(out), N, K, _stream(w2)), "sbk_split_x3p")
            self.keep.append(out)
            return out.data_ptr()

        for l, S in enumerate(layer_specs):
            o = layers[l]
            o.ln1_g, o.ln1_b = map(ptr, S["ln1"])
            o.sa_in_w, o.sa_in_b = map(ptr, S["sa_in"])
            o.sa_out_w, o.sa_out_b = map(ptr, S["sa_out"])
            o.ln2_g, o.ln2_b = map(ptr, S["ln2"])
            o.ca_in_w, o.ca_in_b = map(ptr, S["ca_in"])
            o.ca_out_w, o.ca_out_b = map(ptr, S["ca_out"])
            o.ln3_g, o.ln3_b = map(ptr, S["ln3"])
            o.ff1_w, o.ff1_b = map(ptr, S["ff1"])
            o.ff2_w, o.ff2_b = map(ptr, S["ff2"])
            o.ca_kv_w3 = split3(S["ca_in"][0][dm:])
            o.sa_in_wp, o.sa_out_wp = panel(S["sa_in"][0]), panel(S["sa_out"][0])
            o.ca_q_wp, o.ca_out_wp = panel(S["ca_in"][0][:dm]), panel(S["ca_out"][0])
            o.ff1_wp, o.ff2_wp = panel(S["ff1"][0]), panel(S["ff2"][0])
            if fold:  # LayerNorm folded into the projection it feeds (fused kernels, csrc/gemm.hip)
                for name, (wf, bf) in (("sa_in", _fold_ln(*S["sa_in"], *S["ln1"])),
                                       ("ca_q", _fold_ln(S["ca_in"][0][:dm], S["ca_in"][1][:dm], *S["ln2"])),
                                       ("ff1", _fold_ln(*S["ff1"], *S["ln3"]))):
                    setattr(o, name + "_wf", ptr(wf))
                    setattr(o, name + "_bf", ptr(bf))
                    setattr(o, name + "_wfp", panel(wf))  # (ABI 9: the same fold for the rows sbk_gemm_ln_nt_x3r takes)
        self.layers = layers
        W = DecoderWeights()
        W.layers = ctypes.cast(layers, POINTER(DecoderLayer))
        W.emb, W.pe = ptr(emb), ptr(pe)
        W.final_ln_g, W.final_ln_b = map(ptr, final_ln)
        if seq is not None:
            W.seq_w, W.seq_b = map(ptr, seq)
            W.seq_w3 = split3(seq[0])
            W.seq_wp = panel(seq[0])
            if fold:
                wf, bf = _fold_ln(seq[0], seq[1], *final_ln)
                W.seq_wf, W.seq_bf, W.seq_wfp = ptr(wf), ptr(bf), panel(wf)
        W.d_model, W.nhead = dm, nhead
        W.d_ffn, W.n_layers = layer_specs[0]["ff1"][0].shape[0], len(layer_specs)
        W.vocab = seq[0].shape[0] if seq is not None else emb.shape[0]
        W.max_len, W.ffn_act, W.ln_eps, W.emb_scale = pe.shape[-2], ffn_act, ln_eps, float(emb_scale)
        self.W = W
        self.device = emb.device
        self.ready = _Ready(emb.device)  # (the folded / split tables above were written on this thread's stream)

    @staticmethod
    def source_key(model, seq_lin=None):
        """(data_ptr, _version) of every tensor the handle was derived from -- the folded LayerNorm copies go
        stale when a source parameter is updated IN PLACE (load_state_dict, Pretrainer, copy_), which keeps
        data_ptr and bumps _version.  Cheap: no handle is built."""
        src = list(model.decoder.parameters())
        src.append(model.custom_tgt_module.layers[0].emb.Embedding.weight)
        src.append(model.positional_encoding_decoder.pe)
        if seq_lin is not None:
            src.extend(seq_lin.parameters())
        return tuple((t.data_ptr(), t._version) for t in src)

    def stale(self, model, seq_lin=None):
        return self.source_key(model, seq_lin) != self.key


class LMHandle:
    """Device pointers of a TransformerLM laid out as sbk_lm_weights (keeps the tensors alive)."""

    def __init__(self, lm):
        self.keep = []

        def ptr(t):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            _dev_ok(t)
            _f32(t)
            self.keep.append(t)
            return t.data_ptr()

        enc = lm.encoder
        layers = (LMLayer * len(enc.layers))()
        for l, L in enumerate(enc.layers):
            o = layers[l]
            o.in_w, o.in_b = ptr(L.self_att.att.in_proj_weight), ptr(L.self_att.att.in_proj_bias)
            o.out_w, o.out_b = ptr(L.self_att.att.out_proj.weight), ptr(L.self_att.att.out_proj.bias)
            o.ln1_g, o.ln1_b = ptr(L.norm1.norm.weight), ptr(L.norm1.norm.bias)
            o.ff1_w, o.ff1_b = ptr(L.pos_ffn.ffn[0].weight), ptr(L.pos_ffn.ffn[0].bias)
            o.ff2_w, o.ff2_b = ptr(L.pos_ffn.ffn[3].weight), ptr(L.pos_ffn.ffn[3].bias)
            o.ln2_g, o.ln2_b = ptr(L.norm2.norm.weight), ptr(L.norm2.norm.bias)
        self.layers = layers
        W = LMWeights()
        W.layers = ctypes.cast(layers, POINTER(LMLayer))
        emb = lm.custom_src_module.emb.Embedding.weight
        pe = lm.positional_encoding.pe
        W.emb, W.pe = ptr(emb), ptr(pe.reshape(pe.shape[-2], pe.shape[-1]))
        W.final_ln_g, W.final_ln_b = ptr(enc.norm.norm.weight), ptr(enc.norm.norm.bias)
        o0, oln, o2 = lm.output_proj.layers[0], lm.output_proj.layers[1], lm.output_proj.layers[2]
        W.out0_w, W.out0_b = ptr(o0.w.weight), ptr(o0.w.bias)
        W.out_ln_g, W.out_ln_b = ptr(oln.norm.weight), ptr(oln.norm.bias)
        W.out2_w, W.out2_b = ptr(o2.w.weight), ptr(o2.w.bias)
        W.d_model, W.nhead = emb.shape[1], enc.layers[0].nhead
        W.d_ffn, W.n_layers, W.vocab = enc.layers[0].pos_ffn.ffn[0].weight.shape[0], len(enc.layers), o2.w.weight.shape[0]
        W.max_len, W.ffn_act = pe.shape[-2], enc.layers[0].pos_ffn.act_code
        W.normalize_before, W.pad_idx, W.ln_eps = int(enc.layers[0].normalize_before), 0, enc.norm.eps
        self.W = W
        self.device = emb.device
        self.ready = _Ready(emb.device)
        self.key = self.source_key(lm)

    @staticmethod
    def source_key(lm):
        """(data_ptr, _version) of every source tensor (see DecoderHandle.source_key)."""
        return tuple((t.data_ptr(), t._version) for t in list(lm.parameters()) + [lm.positional_encoding.pe])

    def stale(self, lm):
        return self.source_key(lm) != self.key


def lm_prefix(handle: "LMHandle", tokens):
    """TransformerLM.forward: tokens [n,L] int32 -> logits [n,L,V] (KV-cached, causal, pad-0 keys masked)."""
    lib = load()
    _dev_ok(tokens)
    n, L = tokens.shape
    handle.ready.wait(tokens.device)
    nbytes = lib.sbk_lm_prefix_workspace_bytes(ctypes.byref(handle.W), n, L)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=tokens.device)
    off = (-ws.data_ptr()) % 256
    out = torch.empty(n, L, handle.W.vocab, dtype=torch.float32, device=tokens.device)
    _chk(lib.sbk_lm_prefix_f32(ctypes.byref(handle.W), _p(tokens), c_void_p(ws.data_ptr() + off), nbytes, _p(out), n, L,
                               _stream(tokens)), "sbk_lm_prefix_f32")
    return out


def _fold_ln(w, b, gamma, beta):
    """(Wf, bf) with a preceding LayerNorm's affine folded in: Wf = W*gamma[k], bf = b + W.beta (fp64, rounded once)."""
    w64, g64, be64 = w.detach().double(), gamma.detach().double(), beta.detach().double()
    wf = (w64 * g64[None, :]).float().contiguous()
    bf = (b.detach().double() + w64 @ be64).float().contiguous()
    return wf, bf


def _host_flag(device):
    """A host int32 the C ABI takes as its "stop-rule polling allowed" argument (the library polls through its own
    pinned ring, csrc/search.hip AsyncPoll).  One per host thread, allocated once: pinning memory per search call costs
    a driver round trip per call."""
    t = getattr(_tls, "host_flag", None)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32)
        _tls.host_flag = t
    return t


def beam_search(handle: DecoderHandle, cfg: SearchConfig, enc, enc_len, ctc_w=None, ctc_b=None, utt_min_steps=None,
                utt_max_steps=None, want_longest=False):
    """Returns (tokens [B,max_steps] int32, lens [B] int32, scores [B], log_probs [B,max_steps],
    max_len [1] int32 (longest finished hypothesis), steps_run) -- plus longest [B] int32 with ``want_longest``.
    ``utt_min_steps`` / ``utt_max_steps`` (int32 [B] on the device): per-utterance step limits of a grouped search."""
    lib = load()
    _dev_ok(enc, enc_len, ctc_w, ctc_b, utt_min_steps, utt_max_steps)
    cfg.utt_min_steps = utt_min_steps.data_ptr() if utt_min_steps is not None else None
    cfg.utt_max_steps = utt_max_steps.data_ptr() if utt_max_steps is not None else None
    _f32(enc)
    B, T, _ = enc.shape
    dev = enc.device
    L = max(int(cfg.max_steps), 1)
    K = max(int(cfg.topk), 1)  # rows per utterance (return_topk)
    handle.ready.wait(dev)
    nbytes = lib.sbk_beam_search_workspace_bytes(ctypes.byref(handle.W), ctypes.byref(cfg), B, T)
    ws, ws_key = _search_workspace(nbytes + 256, dev)
    off = (-ws.data_ptr()) % 256
    out_tok = torch.zeros(B * K, L, dtype=torch.int32, device=dev)
    out_len = torch.zeros(B * K, dtype=torch.int32, device=dev)
    out_score = torch.zeros(B * K, dtype=torch.float32, device=dev)
    out_lp = torch.zeros(B * K, L, dtype=torch.float32, device=dev)
    out_max = torch.zeros(1, dtype=torch.int32, device=dev)
    out_longest = torch.zeros(B, dtype=torch.int32, device=dev) if want_longest else None
    flag = _host_flag(dev)
    steps = c_int32(0)
    cfg.ctc_w3 = None
    if ctc_w is not None and F32X3 and ctc_w.dim() == 2 and ctc_w.is_contiguous() and ctc_w.shape[1] % 32 == 0 \
            and ctc_w.shape[1] >= 64 and ctc_w.shape[0] % 4 == 0:
        cfg.ctc_w3 = lp_weight(ctc_w, "x3").data_ptr()  # (cached with the parameter: the CTC head over B*T frames)
    try:
        _chk(lib.sbk_beam_search_f32(ctypes.byref(handle.W), ctypes.byref(cfg), _p(enc), _p(enc_len), _p(ctc_w), _p(ctc_b),
                                     c_void_p(ws.data_ptr() + off), nbytes, _p(out_tok), _p(out_len), _p(out_score),
                                     _p(out_lp), _p(out_max), _p(out_longest), c_void_p(flag.data_ptr()),
                                     ctypes.byref(steps), B, T, _stream(enc)),
             "sbk_beam_search_f32")
    finally:
        _search_workspace_done(ws_key)
    if want_longest:
        return out_tok, out_len, out_score, out_lp, out_max, steps.value, out_longest
    return out_tok, out_len, out_score, out_lp, out_max, steps.value


# SBK_WS_TRIM=1: an outgrown multi-GB search buffer is handed back to the driver (torch.cuda.empty_cache) instead of staying in
# its stream's pool.  OFF by default: on one box (profiles/r05_i_*) it took the bench's reserved memory from 81.5 to 67 GB but
# the headline from 11.7 K to 8.7-11.1 K audio-s/s -- the emptied pools are refilled by hipMalloc inside the next job.
WS_TRIM = os.environ.get("SBK_WS_TRIM", "0") != "0"
_SEARCH_WS = {}  # (device index, stream handle) -> [buffer, a host thread is enqueueing a search into it]
_SEARCH_WS_LOCK = threading.RLock()  # (re-entrant: a transcriber finalised by the cyclic GC inside an allocation made under the lock)


def _search_workspace(nbytes: int, dev):
    """The search workspace of the current stream: ONE grow-only buffer per (device, stream), reused by the searches that
    follow each other on that stream (stream order makes the reuse safe; the library joins its helper stream before it
    returns) and kept ACROSS the jobs of a transcriber (``release_search_workspaces``: when its streams retire).
    Rounds 3-4 kept it per (thread, stream) and dropped it after every job: the caching allocator then served a later, smaller
    request by splitting the cached block and a still later, larger one from a fresh segment -- 106-133 GB reserved for 62-71 GB
    allocated.  Returns (buffer, key); ``_search_workspace_done(key)`` when the library call has
    returned.  A second host thread that starts a search on the same stream meanwhile gets a private buffer (key None)."""
    if dev.type != "cuda":
        return torch.empty(nbytes, dtype=torch.uint8, device=dev), None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    with _SEARCH_WS_LOCK:
        ent = _SEARCH_WS.get(key)
        if ent is not None and ent[1]:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev), None
        if ent is None or ent[0].numel() < nbytes:
            grown = WS_TRIM and ent is not None and ent[0].numel() >= (256 << 20)
            _SEARCH_WS.pop(key, None)
            ent = None  # (the old block goes back to the allocator before the larger one is requested)
            if grown:
                # (opt-in, see WS_TRIM: a multi-GB block that has just been outgrown sits in this stream's pool for good -- no
                #  later request of the pool is that large, and another stream's pool cannot take it)
                torch.cuda.empty_cache()
            ent = _SEARCH_WS[key] = [torch.empty(nbytes + (nbytes >> 3), dtype=torch.uint8, device=dev), False]
        ent[1] = True
        return ent[0], key


def _search_workspace_done(key):
    if key is None:
        return
    with _SEARCH_WS_LOCK:
        ent = _SEARCH_WS.get(key)
        if ent is not None:
            ent[1] = False


def release_search_workspaces(streams=None):
    """Drop the cached search workspaces of the given ``torch.cuda.Stream`` objects (a ConcurrentTranscriber calls it for its
    worker streams when it is closed), or -- ``None`` -- of every stream no search is running on."""
    with _SEARCH_WS_LOCK:
        if streams is None:
            keys = [k for k, ent in _SEARCH_WS.items() if not ent[1]]
        else:
            keys = [(s.device.index, s.cuda_stream) for s in streams]
        for k in keys:
            ent = _SEARCH_WS.get(k)
            if ent is not None and not ent[1]:  # (a search of another owner of the same pooled handle is running on it: kept)
                _SEARCH_WS.pop(k, None)


def greedy_search(handle: DecoderHandle, enc, enc_len, min_steps, max_steps, bos, eos, check_every=8):
    """Returns (tokens [B,max_steps] int32 (EOS-latched), scores [B,max_steps], steps_run)."""
    lib = load()
    _dev_ok(enc, enc_len)
    B, T, _ = enc.shape
    dev = enc.device
    L = max(int(max_steps), 1)
    handle.ready.wait(dev)
    nbytes = lib.sbk_greedy_search_workspace_bytes(ctypes.byref(handle.W), B, T, max_steps)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    out_tok = torch.zeros(B, L, dtype=torch.int32, device=dev)
    out_sc = torch.zeros(B, L, dtype=torch.float32, device=dev)
    flag = _host_flag(dev)
    steps = c_int32(0)
    _chk(lib.sbk_greedy_search_f32(ctypes.byref(handle.W), _p(enc), _p(enc_len), c_void_p(ws.data_ptr() + off), nbytes,
                                   _p(out_tok), _p(out_sc), c_void_p(flag.data_ptr()), ctypes.byref(steps), B, T,
                                   int(min_steps), int(max_steps), int(bos), int(eos), int(check_every), _stream(enc)),
         "sbk_greedy_search_f32")
    return out_tok, out_sc, steps.value


def prompted_greedy_search(handle: DecoderHandle, enc, enc_len, prompt, max_new, eos, logit_bias=None, first_bias=None,
                           probe=None, check_every=8):
    """S2SWhisperGreedySearcher on the device: prompt [B,P] int32 primes the KV cache, then arg-max decoding with the
    additive [V] masks.  Returns (tokens [B,max_new] (EOS-latched), scores [B,max_new], steps_run, probe [B] | None);
    ``probe`` = (prompt position, token): softmax probability of the token at that position."""
    lib = load()
    _dev_ok(enc, enc_len, prompt, logit_bias, first_bias)
    B, T, _ = enc.shape
    P = prompt.shape[1]
    dev = enc.device
    L = max(int(max_new), 1)
    handle.ready.wait(enc.device)
    nbytes = lib.sbk_prompted_greedy_search_workspace_bytes(ctypes.byref(handle.W), B, T, P, int(max_new))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    out_tok = torch.zeros(B, L, dtype=torch.int32, device=dev)
    out_sc = torch.zeros(B, L, dtype=torch.float32, device=dev)
    out_probe = torch.zeros(B, dtype=torch.float32, device=dev) if probe is not None else None
    flag = _host_flag(dev)
    steps = c_int32(0)
    _chk(lib.sbk_prompted_greedy_search_f32(
        ctypes.byref(handle.W), _p(enc), _p(enc_len), _p(prompt), P, _p(logit_bias), _p(first_bias),
        c_void_p(ws.data_ptr() + off), nbytes, _p(out_tok), _p(out_sc), int(probe[0]) if probe else 0,
        int(probe[1]) if probe else 0, _p(out_probe), c_void_p(flag.data_ptr()), ctypes.byref(steps), B, T, int(max_new),
        int(eos), int(check_every), _stream(enc)), "sbk_prompted_greedy_search_f32")
    return out_tok, out_sc, steps.value, out_probe


def decoder_prefix(handle: DecoderHandle, tokens, enc, enc_len):
    """tokens [n,L] int32, enc [n,T,d] -> decoder.norm output [n,L,d] (KV-cached, teacher forced)."""
    lib = load()
    _dev_ok(tokens, enc, enc_len)
    n, L = tokens.shape
    T = enc.shape[1]
    handle.ready.wait(enc.device)
    nbytes = lib.sbk_decoder_prefix_workspace_bytes(ctypes.byref(handle.W), n, T, L)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=enc.device)
    off = (-ws.data_ptr()) % 256
    pred = torch.empty(n, L, handle.W.d_model, dtype=torch.float32, device=enc.device)
    _chk(lib.sbk_decoder_prefix_f32(ctypes.byref(handle.W), _p(tokens), _p(enc), _p(enc_len),
                                    c_void_p(ws.data_ptr() + off), nbytes, _p(pred), n, T, L, _stream(enc)),
         "sbk_decoder_prefix_f32")
    return pred


class CTCStepScorer:
    """The CTC prefix scorer as a per-step object (sbk_ctc_scorer_*: decoders/ctc.py:79-295 CTCPrefixScore).  ``log_probs``
    [B,T,V] = log_softmax(ctc_fc(enc)) -- the scorer takes ownership and converts it in place --, ``enc_len`` [B] int32 absolute
    lengths.  ``score(inp_tokens, step)`` -> psi - psi_prev [B*beam, V]; ``permute(parent, token, inp_tokens, step)`` moves the
    state to the hypotheses chosen by the beam update.  The beam width is fixed by the first ``score`` call."""

    def __init__(self, log_probs, enc_len, blank, eos, ctc_window_size=0):
        _dev_ok(log_probs, enc_len)
        _f32(log_probs)
        self.x, self.enc_len = log_probs, enc_len.to(torch.int32)
        self.B, self.T, self.V = log_probs.shape
        self.blank, self.eos, self.window = int(blank), int(eos), int(ctc_window_size)
        self.beam, self.ws, self.nbytes = 0, None, 0

    def _ready(self, beam):
        if self.beam == beam:
            return
        if self.beam:
            raise SbkError(f"CTCStepScorer: the beam width changed from {self.beam} to {beam} inside one utterance batch")
        lib = load()
        self.beam = beam
        self.nbytes = lib.sbk_ctc_scorer_workspace_bytes(self.B, self.T, self.V, beam)
        self.ws = torch.empty(self.nbytes + 256, dtype=torch.uint8, device=self.x.device)
        self._off = (-self.ws.data_ptr()) % 256
        _chk(lib.sbk_ctc_scorer_reset_f32(_p(self.x), _p(self.enc_len), c_void_p(self.ws.data_ptr() + self._off), self.nbytes, self.B,
                                          self.T, self.V, beam, self.blank, _stream(self.x)), "sbk_ctc_scorer_reset_f32")

    def score(self, inp_tokens, step, attn_window=None):
        n = inp_tokens.shape[0]
        if n % self.B:
            raise SbkError(f"CTCStepScorer: {n} hypotheses for {self.B} utterances")
        self._ready(n // self.B)
        tok = inp_tokens.to(torch.int32).contiguous()
        out = torch.empty(n, self.V, dtype=torch.float32, device=self.x.device)
        _chk(load().sbk_ctc_scorer_score_f32(_p(self.x), _p(self.enc_len), c_void_p(self.ws.data_ptr() + self._off), self.nbytes, _p(tok),
                                             int(step), _p(attn_window), self.window if attn_window is not None else 0, _p(out), self.B,
                                             self.T, self.V, self.beam, self.blank, self.eos, _stream(self.x)), "sbk_ctc_scorer_score_f32")
        return out

    def permute(self, parent, token, parent_last_tok, step, attn_window=None):
        par, tk, last = (t.to(torch.int32).contiguous() for t in (parent, token, parent_last_tok))
        _chk(load().sbk_ctc_scorer_permute_f32(_p(self.x), c_void_p(self.ws.data_ptr() + self._off), self.nbytes, _p(par), _p(tk), _p(last),
                                               int(step), _p(attn_window), self.window if attn_window is not None else 0, self.B, self.T,
                                               self.V, self.beam, self.blank, _stream(self.x)), "sbk_ctc_scorer_permute_f32")


def log_softmax(x, temperature=1.0, weight=1.0):
    lib = load()
    V = x.shape[-1]
    x2 = x.reshape(-1, V)
    _dev_ok(x2)
    out = torch.empty_like(x)
    _chk(lib.sbk_log_softmax_f32(_p(x2), _p(out), x2.shape[0], V, float(temperature), float(weight), _stream(x2)),
         "sbk_log_softmax_f32")
    return out


# ------------------------------------------------------------------ HIP-event profiler
def prof_enable(on: bool):
    load().sbk_prof_enable(1 if on else 0)


def prof_reset():
    load().sbk_prof_reset()


def prof_report():
    """{kernel: dict(count, ms, flops, bytes)} for everything launched since prof_reset()."""
    lib = load()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.sbk_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, count, ms, flops, nbytes = line.split()
        out[name] = dict(count=int(count), ms=float(ms), flops=float(flops), bytes=float(nbytes))
    return out
