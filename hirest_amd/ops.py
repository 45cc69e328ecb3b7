"""Tensor-level wrappers over the C ABI (one function per kernel entry point).

torch is used for device memory and the current stream only; every computation below is a
hand-written gfx950 kernel inside libhirest_hip.so.  All wrappers validate device / dtype /
contiguity and raise RuntimeError on violations (the reference's ATen ops raise likewise).
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BIAS_BF16, EPI_BIAS_F32, EPI_BIAS_GELU_BF16, EPI_BIAS_QGELU_BF16, EPI_BIAS_RESID_F32,
                   EPI_PATCH_POS_F32)

__all__ = ["gemm", "layernorm", "attention", "patchify", "write_cls_rows", "embed_tokens", "to_bf16",
           "pool_l2norm", "similarity", "topk", "stream_ptr", "on_tensor_device"]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream on the CURRENT device (wrappers switch to their tensors' device first).  Called
    once per kernel launch — the training step issues ~300 per iteration from Python — so it takes torch's raw accessor when the
    build has it (no Stream object per call)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def on_tensor_device(fn):
    """Run `fn` with the device of its first GPU tensor argument current.  The reference's drivers place models with
    ``.to(device)`` and never call ``set_device`` (run.py:61-65, inference_video_retrieval.py:195-201); a kernel must
    then launch on the stream — and with the per-device kernel attributes — of the tensors' device, not of cuda:0."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kw)
                break
        return fn(*args, **kw)
    return wrapped


def _dev(t: torch.Tensor, dtype, name: str) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], dtype, name: str) -> Optional[int]:
    return None if t is None else _dev(t, dtype, name)


@on_tensor_device
def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epilogue: int,
         pos: Optional[torch.Tensor] = None, patches_per_frame: int = 0,
         aux0: Optional[torch.Tensor] = None, aux1: Optional[torch.Tensor] = None, flags: int = 0) -> torch.Tensor:
    """out <- epilogue(a @ w.T + bias); a [M,K] bf16, w [N,K] bf16, bias [N] f32.  aux0 / aux1: the LN-fold epilogues'
    extra operands (include/hirest_hip.h)."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f"gemm: K mismatch {a.shape} x {w.shape}")
    out_dtype = torch.bfloat16 if epilogue in (EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_QGELU_BF16, _lib.EPI_LNFOLD_BF16,
                                              _lib.EPI_LNFOLD_GELU_BF16, _lib.EPI_BIAS_RESID2_LNSTATS) else torch.float32
    args = _lib.GemmArgs.make(_dev(a, torch.bfloat16, "gemm.a"), K, _dev(w, torch.bfloat16, "gemm.w"), K,
                         _opt(bias, torch.float32, "gemm.bias"), _dev(out, out_dtype, "gemm.out"), out.shape[-1],
                         M, N, K, epilogue, _opt(pos, torch.float32, "gemm.pos"), patches_per_frame,
                         None if aux0 is None else _dev(aux0, aux0.dtype, "gemm.aux0"),
                         None if aux1 is None else _dev(aux1, aux1.dtype, "gemm.aux1"), int(flags))
    _lib.check(lib.hirest_gemm_bf16(C.byref(args), stream_ptr()), "hirest_gemm_bf16")
    return out


def gemm_select_kernel(which: int):
    """0 = automatic, 1 = force the 128x128 kernel, 2 = force the 256x256 ping-pong kernel."""
    _lib.check(_lib.load().hirest_gemm_select_kernel(int(which)), "hirest_gemm_select_kernel")


ATTENTION_DEFAULT_KERNEL = 7     # persistent kernel + producer wave (include/hirest_hip.h: hirest_attention_select_kernel)


def attention_select_kernel(which: int):
    _lib.check(_lib.load().hirest_attention_select_kernel(int(which)), "hirest_attention_select_kernel")


def attention_set_mapping(by_head: Optional[bool] = None):
    """Persistent attention kernel: one head per workgroup over frames (True), one frame per workgroup (False), or automatic (None, the
    default: by head below 256 frames)."""
    _lib.check(_lib.load().hirest_attention_set_mapping(2 if by_head is None else int(bool(by_head))), "hirest_attention_set_mapping")


@on_tensor_device
def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor,
              row_index: Optional[torch.Tensor] = None, ldx: Optional[int] = None, rows: Optional[int] = None):
    lib = _lib.load()
    D = gamma.numel()
    rows = rows if rows is not None else (row_index.numel() if row_index is not None else x.numel() // D)
    ldx = ldx if ldx is not None else D
    out_f32 = 1 if out.dtype == torch.float32 else 0
    _lib.check(lib.hirest_layernorm(_dev(x, torch.float32, "ln.x"), ldx, _opt(row_index, torch.int32, "ln.row_index"),
                                    _dev(gamma, torch.float32, "ln.gamma"), _dev(beta, torch.float32, "ln.beta"),
                                    float(eps), _dev(out, out.dtype, "ln.out"), D, out_f32, rows, D, stream_ptr()),
               "hirest_layernorm")
    return out


@on_tensor_device
def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, N: int, H: int, dh: int, causal: bool,
              scale: Optional[float] = None):
    lib = _lib.load()
    scale = dh ** -0.5 if scale is None else scale
    _lib.check(lib.hirest_attention_bf16(_dev(qkv, torch.bfloat16, "attn.qkv"), _dev(out, torch.bfloat16, "attn.out"),
                                         B, N, H, dh, float(scale), int(bool(causal)), stream_ptr()),
               "hirest_attention_bf16")
    return out


_IN_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.uint8: 2}


@on_tensor_device
def patchify(frames: torch.Tensor, patch: int, kpad: int, out: torch.Tensor, mean=None, std=None):
    lib = _lib.load()
    if frames.dtype not in _IN_DTYPES:
        raise RuntimeError(f"patchify: unsupported frame dtype {frames.dtype}")
    code = _IN_DTYPES[frames.dtype]
    B = frames.shape[0]
    S = frames.shape[-1] if code != 2 else frames.shape[1]
    _lib.check(lib.hirest_patchify(_dev(frames, frames.dtype, "patchify.frames"), code, B, S, patch,
                                   _opt(mean, torch.float32, "mean"), _opt(std, torch.float32, "std"),
                                   _dev(out, torch.bfloat16, "patchify.out"), kpad, stream_ptr()), "hirest_patchify")
    return out


@on_tensor_device
def write_cls_rows(x: torch.Tensor, cls: torch.Tensor, pos0: torch.Tensor, B: int, T: int, D: int):
    lib = _lib.load()
    _lib.check(lib.hirest_write_cls_rows(_dev(x, torch.float32, "x"), D, _dev(cls, torch.float32, "cls"),
                                         _dev(pos0, torch.float32, "pos"), B, T, D, stream_ptr()), "hirest_write_cls_rows")
    return x


@on_tensor_device
def embed_tokens(tokens: torch.Tensor, tok_emb: torch.Tensor, pos: torch.Tensor, x: torch.Tensor, eot_row: torch.Tensor):
    lib = _lib.load()
    B, L = tokens.shape
    V, D = tok_emb.shape
    _lib.check(lib.hirest_embed_tokens(_dev(tokens, torch.int64, "tokens"), _dev(tok_emb, torch.float32, "tok_emb"),
                                       _dev(pos, torch.float32, "pos"), _dev(x, torch.float32, "x"),
                                       _dev(eot_row, torch.int32, "eot_row"), B, L, D, V, stream_ptr()), "hirest_embed_tokens")
    return x


@on_tensor_device
def to_bf16(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (RNE) on device through the library's cast kernel."""
    lib = _lib.load()
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    n = t.numel()
    if n % 4 != 0:
        raise RuntimeError("to_bf16: element count must be a multiple of 4")
    _lib.check(lib.hirest_f32_to_bf16(_dev(t, torch.float32, "to_bf16.in"), out.data_ptr(), n, stream_ptr()), "hirest_f32_to_bf16")
    return out


@on_tensor_device
def split2(x: torch.Tensor, gelu: bool = False) -> torch.Tensor:
    """[rows, D] fp32 -> [rows, 2D] bf16 in the bf16x3 operand format (hirest_split2_bf16): every 64-column block holds the bf16 hi
    parts of 32 consecutive columns, then their lo parts (lo = bf16(x - hi)).  D % 32 == 0."""
    lib = _lib.load()
    rows, D = x.shape
    out = torch.empty((rows, 2 * D), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.hirest_split2_bf16(_dev(x, torch.float32, "split2.in"), D, out.data_ptr(), 2 * D, rows, D, int(bool(gelu)), stream_ptr()),
               "hirest_split2_bf16")
    return out


@on_tensor_device
def gemm_x3(a2: torch.Tensor, w2: torch.Tensor, bias: Optional[torch.Tensor] = None, resid_out: Optional[torch.Tensor] = None,
            gelu_split: bool = False, stream=None) -> torch.Tensor:
    """fp32 [M, N] = A W^T (+ bias) from split operands a2 [M, 2K], w2 [N, 2K] (ops.split2): HIREST_GEMM_X3.  With `resid_out` (fp32 [M, N])
    the product is added into it (x += ...); with `gelu_split` the result is nn.GELU()(A W^T + bias) as a split operand [M, 2N] bf16
    (HIREST_EPI_BIAS_GELU_SPLIT2: the next layer's A operand, no fp32 round trip)."""
    lib = _lib.load()
    M, K2 = a2.shape
    N = w2.shape[0]
    if gelu_split:
        out = torch.empty((M, 2 * N), dtype=torch.bfloat16, device=a2.device)
        args = _lib.GemmArgs.make(_dev(a2, torch.bfloat16, "gemm_x3.a"), K2, _dev(w2, torch.bfloat16, "gemm_x3.w"), K2,
                                  _opt(bias, torch.float32, "gemm_x3.bias"), out.data_ptr(), 2 * N, M, N, K2, _lib.EPI_BIAS_GELU_SPLIT2, None, 0,
                                  None, None, _lib.GEMM_X3)
        _lib.check(lib.hirest_gemm_bf16(C.byref(args), stream_ptr()), "hirest_gemm_bf16 (x3, gelu + split)")
        return out
    out = resid_out if resid_out is not None else torch.empty((M, N), dtype=torch.float32, device=a2.device)
    epi = _lib.EPI_BIAS_RESID_F32 if resid_out is not None else _lib.EPI_BIAS_F32
    args = _lib.GemmArgs.make(_dev(a2, torch.bfloat16, "gemm_x3.a"), K2, _dev(w2, torch.bfloat16, "gemm_x3.w"), K2,
                              _opt(bias, torch.float32, "gemm_x3.bias"), out.data_ptr(), N, M, N, K2, epi, None, 0, None, None, _lib.GEMM_X3)
    _lib.check(lib.hirest_gemm_bf16(C.byref(args), stream_ptr() if stream is None else stream), "hirest_gemm_bf16 (x3)")
    return out


@on_tensor_device
def pool_l2norm(frame_embeds: torch.Tensor, normalize_frames_first: bool = False) -> torch.Tensor:
    """[V,F,E] f32 -> [V,E]: mean over frames then L2 (inference_video_retrieval.py:283-285)."""
    lib = _lib.load()
    V, F, E = frame_embeds.shape
    out = torch.empty((V, E), dtype=torch.float32, device=frame_embeds.device)
    _lib.check(lib.hirest_pool_l2norm(_dev(frame_embeds, torch.float32, "pool.in"), out.data_ptr(), V, F, E,
                                      int(bool(normalize_frames_first)), stream_ptr()), "hirest_pool_l2norm")
    return out


@on_tensor_device
def similarity(text_n: torch.Tensor, video_n: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    Q, E = text_n.shape
    V = video_n.shape[0]
    out = torch.empty((Q, V), dtype=torch.float32, device=text_n.device)
    _lib.check(lib.hirest_similarity_f32(_dev(text_n, torch.float32, "sim.text"), _dev(video_n, torch.float32, "sim.video"),
                                         out.data_ptr(), Q, V, E, stream_ptr()), "hirest_similarity_f32")
    return out


def gemm_f32_strided(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """alpha * a @ b.T in exact fp32 (hirest_gemm_f32_strided; a [M, K], b [N, K], last dim contiguous)."""
    lib = _lib.load()
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(lib.hirest_gemm_f32_strided(_dev(a, torch.float32, "gemm.a"), a.stride(0), 1, _dev(b, torch.float32, "gemm.b"), b.stride(0), 1,
                                           out.data_ptr(), N, M, N, K, float(alpha), stream_ptr()), "hirest_gemm_f32_strided")
    return out


@on_tensor_device
def topk(scores: torch.Tensor, k: int, tie_rank: Optional[torch.Tensor] = None):
    lib = _lib.load()
    Q, V = scores.shape
    idx = torch.empty((Q, k), dtype=torch.int32, device=scores.device)
    val = torch.empty((Q, k), dtype=torch.float32, device=scores.device)
    if V >= 16384:                                   # long rows (beam search over beams * vocab): chunked two-pass selection
        need = lib.hirest_topk_workspace_bytes(Q, V, k)
        ws = torch.empty(need, dtype=torch.uint8, device=scores.device)
        _lib.check(lib.hirest_topk_f32_ws(_dev(scores, torch.float32, "topk.scores"), _opt(tie_rank, torch.int32, "topk.tie_rank"),
                                          Q, V, k, idx.data_ptr(), val.data_ptr(), ws.data_ptr(), need, stream_ptr()),
                   "hirest_topk_f32_ws")
        return val, idx
    _lib.check(lib.hirest_topk_f32(_dev(scores, torch.float32, "topk.scores"), _opt(tie_rank, torch.int32, "topk.tie_rank"),
                                   Q, V, k, idx.data_ptr(), val.data_ptr(), stream_ptr()), "hirest_topk_f32")
    return val, idx


def to_device(t: torch.Tensor, device) -> torch.Tensor:
    """t.to(device) that does not stall the host when it need not: a pinned CPU tensor (what DataLoader(pin_memory=True) delivers:
    hirest_dataset.py:614,624) is copied asynchronously on the current stream — the kernels that read it are behind it on the same
    stream — everything else as t.to(device).  As with any non_blocking copy, the source must not be overwritten before the copy has
    run (DataLoader hands out fresh pinned tensors per batch; a caller that refills one pinned buffer in place must synchronise first)."""
    if t.device.type == "cpu" and t.is_pinned():
        return t.to(device, non_blocking=True)
    return t.to(device)


_F32_GEMM_WS = {}
_F32_GEMM_RETIRED = []


def f32_gemm_workspace(device, nbytes: int, tag: int = 0, stream: Optional[int] = None):
    """Scratch for hirest_gemm_f32_ws (the split form of few-tile fp32 GEMMs): one buffer per (device, tag, stream), so two
    streams never share scratch (tag 1: the training step's side stream; `stream` = the raw stream pointer the GEMM is
    enqueued on, default the current torch stream).  Grown geometrically on demand; an outgrown buffer is RETIRED, not
    freed: torch's caching allocator only orders a block's reuse against the stream it was allocated on, and the split
    GEMMs still in flight on this key's stream may be reading it — the retired blocks (their sizes sum to less than the
    live buffer) stay referenced for the life of the process.  Returns (pointer, bytes) — (None, 0) when the problem
    wants none."""
    if nbytes <= 0:
        return None, 0
    key = (device.type, device.index, tag, int(stream) if stream is not None else stream_ptr())
    buf = _F32_GEMM_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _F32_GEMM_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 2 * (buf.numel() if buf is not None else 0), 32 << 20), dtype=torch.uint8, device=device)
        _F32_GEMM_WS[key] = buf
    return buf.data_ptr(), buf.numel()
