"""ctypes binding of libsbk_hip.so (include/sbk.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every
numeric op on the hot path is one of the C-ABI entry points below.  There is NO
CPU fallback: a CPU tensor, or a missing library, raises.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsbk_hip.so")

ACT_NONE, ACT_SWISH, ACT_GELU, ACT_RELU, ACT_LEAKY_RELU = 0, 1, 2, 3, 4

_lib = None
_host_tensors_ok = False  # flipped only by tests that attach the kernel emulator


class SbkError(RuntimeError):
    pass


def _declare(lib):
    p, i, f = c_void_p, c_int, c_float
    sig = {
        "sbk_abi_version": ([], c_int),
        "sbk_last_error": ([], c_char_p),
        "sbk_fbank_f32": ([p, p, p, POINTER(c_int32), i, p, p, p, p, p, i, i, i, i, i, i, f, f, p, p, f, p], c_int),
        "sbk_input_norm_global_f32": ([p, p, p, p, i, i, f, p], c_int),
        "sbk_gemm_nt_f32": ([p, i, p, i, p, p, i, p, i, i, i, i, i, f, p, i, p], c_int),
        "sbk_conv_block_f32": ([p, p, p, p, p, p, i, i, i, i, i, f, f, p], c_int),
        "sbk_relpos_attention_f32": ([p, p, p, p, p, p, p, i, i, i, i, f, p], c_int),
        "sbk_glu_dwconv_f32": ([p, p, p, p, i, i, i, i, p], c_int),
        "sbk_layernorm_f32": ([p, p, p, p, i, i, f, i, p], c_int),
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
    if lib.sbk_abi_version() != 1:
        raise SbkError(f"ABI version mismatch: {lib.sbk_abi_version()}")
    _lib = lib
    return lib


def _attach_for_tests(path: str):
    """Tests only: point the binding at the CPU kernel emulator build of the same sources."""
    global _lib, _host_tensors_ok
    _lib = None
    load(path)
    _host_tensors_ok = True


def _detach_for_tests():
    global _lib, _host_tensors_ok
    _lib = None
    _host_tensors_ok = False


def _chk(rc: int, what: str):
    if rc != 0:
        raise SbkError(f"{what} failed (rc={rc}): {_lib.sbk_last_error().decode()}")


def _dev_ok(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda and not _host_tensors_ok:
            raise SbkError(
                "speechbrain_amd ops run on an MI355X (HIP) device only; got a CPU tensor. "
                "Move the module and inputs to 'cuda' (there is no CPU fallback)."
            )
        if not t.is_contiguous():
            raise SbkError("non-contiguous tensor passed to a kernel")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _f32(t):
    if t.dtype != torch.float32:
        raise SbkError(f"expected float32, got {t.dtype}")
    return t


# ------------------------------------------------------------------ ops
def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act=ACT_NONE, alpha=1.0, out=None,
            seq_len=None, rows_per_seq=0):
    """out[M,N] = residual + alpha * act(a[M,K] @ w[N,K]^T + bias).  `a` may have leading dims.

    With ``seq_len`` (int32 [batch]) rows are [batch][rows_per_seq] and the rows past each
    sequence's length contribute 0 before the residual is added."""
    lib = load()
    K = a.shape[-1]
    a2 = a.reshape(-1, K)
    M, N = a2.shape[0], w.shape[0]
    _dev_ok(a2, w, bias, residual)
    _f32(a2), _f32(w)
    if out is None:
        out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _dev_ok(seq_len)
    _chk(lib.sbk_gemm_nt_f32(_p(a2), K, _p(w), w.stride(0), _p(bias), _p(r2), N, _p(out), N, M, N, K, act,
                             float(alpha), _p(seq_len), int(rows_per_seq), _stream(a2)), "sbk_gemm_nt_f32")
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


def input_norm_global(x, mean, std, eps):
    lib = load()
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    _dev_ok(x2, mean, std)
    out = torch.empty_like(x)
    _chk(lib.sbk_input_norm_global_f32(_p(x2), _p(mean), _p(std), _p(out), x2.shape[0], C, float(eps), _stream(x2)),
         "sbk_input_norm_global_f32")
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


def relpos_attention(qkv, pos, bias_u, bias_v, key_len, H, scale, want_attn=False):
    """qkv [B,T,3*d] (per-head interleaved), pos [2T-1,d] -> context [B,T,d] (+ weights [B,H,T,T])."""
    lib = load()
    _dev_ok(qkv, pos, bias_u, bias_v, key_len)
    _f32(qkv)
    B, T, d3 = qkv.shape
    d = d3 // 3
    out = torch.empty(B, T, d, dtype=torch.float32, device=qkv.device)
    attn = torch.empty(B, H, T, T, dtype=torch.float32, device=qkv.device) if want_attn else None
    _chk(lib.sbk_relpos_attention_f32(_p(qkv), _p(pos), _p(bias_u), _p(bias_v), _p(key_len), _p(out), _p(attn), B, T,
                                      H, d // H, float(scale), _stream(qkv)), "sbk_relpos_attention_f32")
    return out, attn


def glu_dwconv(h, w, bias, ksize):
    """h [B,T,2d] -> depthwise_conv(GLU(h)) [B,T,d]; w [d,ksize]."""
    lib = load()
    _dev_ok(h, w, bias)
    _f32(h)
    B, T, d2 = h.shape
    y = torch.empty(B, T, d2 // 2, dtype=torch.float32, device=h.device)
    _chk(lib.sbk_glu_dwconv_f32(_p(h), _p(w), _p(bias), _p(y), B, T, d2 // 2, int(ksize), _stream(h)),
         "sbk_glu_dwconv_f32")
    return y
