"""Thin torch-tensor front end over the C ABI (device pointers + the current HIP stream).

torch is used for device memory and stream plumbing only; every computation below is a launch into
libudt_kernels.so.  All functions raise if a tensor is not on a GPU — there is no CPU path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import NamedTuple, Optional

import torch

from . import lib as L

# ---------------------------------------------------------------------------------------- launch context
# The stream-K kernels need (a) a workspace whose first 4 KiB (slab flags + error word) are zero between launches
# and that no concurrent stream shares, and (b) the number of launch streams sharing the device (``cu_share``: all
# their workgroups must be resident).  Both are properties of the CALLER's launch stream, so they travel in a
# thread-local context that ``gemm_desc`` / ``run_gemm`` read — the C ABI itself keeps no such state
# (udt_gemm_desc.cu_share, caller-owned workspace).
WORKSPACE_BYTES = 96 << 20      # >= 256 workgroups x (256 x 160) fp32 slab + header: covers every plan of the library


class Workspace:
    """A stream-K workspace owned by whoever launches on one stream (a sampling runner, a captured graph): allocated
    with it, passed explicitly, freed with it, never regrown while launches or captured graphs may reference it."""

    def __init__(self, device, nbytes: int = WORKSPACE_BYTES):
        if torch.device(device).type != "cuda":
            raise L.UdtError("udifftext_amd workspaces live on the GPU (no CPU fallback)")
        # zero-initialised: the kernels keep 'slab ready' flags in the first 4 KiB and restore them to 0
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)

    def check(self, stream: Optional[int] = None) -> None:
        """synchronise ``stream`` (default: current) and raise UdtError if a launch on this workspace timed out"""
        L.check(L.load().udt_check_async_error(self.buf.data_ptr(), self.buf.numel(), _stream() if stream is None else stream),
                "udt_check_async_error")


class _Ctx(threading.local):
    cu_share = 1
    workspace: Optional[Workspace] = None


_ctx = _Ctx()
_default_ws: dict = {}


@contextlib.contextmanager
def launch_context(cu_share: Optional[int] = None, workspace: Optional[Workspace] = None):
    """launches issued inside run with this CU share / on this workspace (thread-local, re-entrant)"""
    prev = (_ctx.cu_share, _ctx.workspace)
    if cu_share is not None:
        _ctx.cu_share = max(1, int(cu_share))
    if workspace is not None:
        _ctx.workspace = workspace
    try:
        yield
    finally:
        _ctx.cu_share, _ctx.workspace = prev


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise L.UdtError("udifftext_amd ops need device tensors (no CPU fallback)")
    return t.data_ptr()


def _ws(nbytes: int, device) -> torch.Tensor:
    ws = _ctx.workspace
    if ws is None:
        # eager callers without a context: one default workspace per (device, stream) — concurrent streams must not
        # share the flags and slabs.  Fixed size, so a buffer a captured graph references is never replaced.
        key = (device.type, device.index, _stream())
        ws = _default_ws.get(key)
        if ws is None:
            ws = _default_ws[key] = Workspace(device)
    if ws.buf.numel() < nbytes:
        raise L.UdtError(f"stream-K workspace of {ws.buf.numel()} bytes is too small for this launch ({nbytes})")
    return ws.buf


def check_async_errors() -> None:
    """check every default (per-stream) workspace — tests and eager callers; runners check their own"""
    for (_, _, stream), ws in list(_default_ws.items()):
        ws.check(stream)


# optional algorithmic work accounting for bench.py's roofline objects: {"gemm": flops, "gemm_bytes": bytes, "attn": ...}
# (2*M*N*K per GEMM launch, 4*B*H*Nq*Nk*D per flash-attention launch; operands + result once for the bytes)
WORK_COUNTER: Optional[dict] = None


def count_work(kind: str, amount: float) -> None:
    if WORK_COUNTER is not None:
        WORK_COUNTER[kind] = WORK_COUNTER.get(kind, 0.0) + amount


def _bf16(t: torch.Tensor) -> None:
    if t.dtype != torch.bfloat16:
        raise ValueError(f"expected bf16 tensor, got {t.dtype}")


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_desc(**kw) -> L.GemmDesc:
    d = L.GemmDesc()
    d.batch = 1
    d.alpha = 1.0
    d.cu_share = _ctx.cu_share
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def run_gemm(d: L.GemmDesc, device, fused_gn: bool = False) -> None:
    lib = L.load()
    need = lib.udt_gemm_workspace_bytes(C.byref(d))
    ws_ptr, ws_bytes = None, 0
    if need:
        ws = _ws(need, device)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    if fused_gn:
        L.check(lib.udt_gn_silu_conv3x3_fwd(C.byref(d), ws_ptr, ws_bytes, _stream()), "udt_gn_silu_conv3x3_fwd")
    else:
        L.check(lib.udt_gemm(C.byref(d), ws_ptr, ws_bytes, _stream()), "udt_gemm")


GN_STRIP = True       # one-launch strip GroupNorm where the shape allows (tests switch it off to compare with the two-kernel path)


class GnStats(NamedTuple):
    """column statistics a producer's epilogue emitted for its output (udt_gemm_desc.colstats): fp32
    [slots, C, 2] = per-(row slot, channel) (sum, sum of squares); a sample owns ``slots_per_sample`` consecutive slots"""
    data: torch.Tensor
    slots_per_sample: int


def gn_stats_of(t: Optional[torch.Tensor]) -> Optional[GnStats]:
    return getattr(t, "gn_stats", None) if t is not None else None


def _attach_colstats(d: L.GemmDesc, out: torch.Tensor, n_cols: int, rows_per_batch: int) -> bool:
    """ask the library whether this problem can emit column statistics; if so allocate them and point the
    descriptor at them (the tensor rides on ``out.gn_stats``)"""
    lib = L.load()
    rows = lib.udt_gemm_colstats_rows(C.byref(d))
    if rows <= 0:
        return False
    slots = lib.udt_gemm_colstats_slots(C.byref(d))
    st = torch.empty((slots, n_cols, 2), dtype=torch.float32, device=out.device)
    d.colstats = st.data_ptr()
    out.gn_stats = GnStats(st, rows_per_batch // rows)
    return True


def _drop_stale_stats(out: Optional[torch.Tensor]) -> None:
    """a caller-provided ``out=`` tensor may still carry the statistics of an earlier launch into it: this launch emits none,
    so a GroupNorm reading ``out.gn_stats`` afterwards must not find the old ones"""
    if out is not None and getattr(out, "gn_stats", None) is not None:
        del out.gn_stats


class Mx8Act(NamedTuple):
    """an activation in the MX8 form the UDT_GEMM_MX8 consumers read (BASELINE config #5): OCP e4m3 elements + one E8M0 scale
    per 32 consecutive columns of a row, written by the PRODUCING kernel's epilogue (udt_gemm_desc.q8_out, udt_tattn_fused_q8,
    udt_attn_rowv_q8_fwd, udt_gn_apply_scsh_q8) — never by a pass of its own.
      data  uint8 [M, K]            e4m3 bytes
      scale int32 [K / 128, M]      byte j of dword (t, m) = scale of columns [128 t + 32 j, + 32) of row m
      stats fp32 [P, M, 2] or None  partial (sum, sum of squares) of every row — for a LayerNorm-folded consumer"""
    data: torch.Tensor
    scale: torch.Tensor
    stats: Optional[torch.Tensor] = None


def mx8_of(t: Optional[torch.Tensor]) -> Optional[Mx8Act]:
    """the MX8 twin a producer attached to its bf16 result (``out.mx8``), if any"""
    return getattr(t, "mx8", None) if t is not None else None


def _mx8_alloc(M: int, cols: int, device) -> Mx8Act:
    return Mx8Act(torch.empty((M, cols), dtype=torch.uint8, device=device),
                  torch.empty(((cols + 127) // 128, M), dtype=torch.int32, device=device))


def _attach_q8(d: L.GemmDesc, M: int, cols: int, device, rowstats: bool, fixed: Optional[tuple] = None) -> Optional[Mx8Act]:
    """ask the library whether this launch can also emit its result as an MX8 activation (+ partial row statistics); if so
    allocate the buffers and point the descriptor at them.  fixed = (first column, multiplier): columns from there on are written
    with that fixed scale instead of block scales (udt_gemm_desc.q8_fixed_col)"""
    q = _mx8_alloc(M, cols, device)
    d.q8_out, d.q8_scale, d.ld_q8 = q.data.data_ptr(), q.scale.data_ptr(), cols
    if fixed is not None:
        d.q8_fixed_col, d.q8_fixed_mul = int(fixed[0]), float(fixed[1])
    lib = L.load()
    parts = lib.udt_gemm_rowstat_parts(C.byref(d)) if rowstats else 0
    if not lib.udt_gemm_q8_ok(C.byref(d)) or (rowstats and parts <= 0):
        d.q8_out, d.q8_scale, d.ld_q8, d.q8_fixed_col = None, None, 0, 0
        return None
    if rowstats:
        st = torch.empty((parts, M, 2), dtype=torch.float32, device=device)
        d.rowstat_out = st.data_ptr()
        q = Mx8Act(q.data, q.scale, st)
    return q


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, out: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, rowvec: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
           flags: int = 0, alpha: float = 1.0, n_out: Optional[int] = None, colstats: bool = False,
           emit_q8: bool = False, emit_rowstats: bool = False) -> torch.Tensor:
    """out[M, N] = epilogue(x[M, K] @ w[N, K]^T).  x may be a strided row view (last dim contiguous).
    colstats: also emit the per-column partial sums of the output (``out.gn_stats``, None when the plan cannot) —
    the GroupNorm statistics of the next layer; needs rows_per_batch (rows of one sample)."""
    _bf16(x); _bf16(w)
    K = x.shape[-1]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x
    assert x2.dim() == 2 and x2.stride(1) == 1
    M = x2.shape[0]
    N = w.shape[0] if n_out is None else n_out
    assert w.shape[1] == K and w.is_contiguous()
    n_cols = N // 2 if (flags & L.GEMM_GEGLU) else N
    if out is None:
        dt = torch.float32 if (flags & L.GEMM_OUT_F32) else torch.bfloat16
        if flags & L.GEMM_TRANSPOSED:
            assert rows_per_batch > 0
            out = torch.empty((M // rows_per_batch, N, rows_per_batch), dtype=dt, device=x.device)
        else:
            out = torch.empty((M, n_cols), dtype=dt, device=x.device)
    ldo = out.stride(0) if not (flags & L.GEMM_TRANSPOSED) else 0
    d = gemm_desc(a=_ptr(x2), w=_ptr(w), bias=_ptr(bias), residual=_ptr(residual), rowvec=_ptr(rowvec), out=_ptr(out),
                  M=M, N=N, K=K, lda=x2.stride(0), ldo=ldo, ldr=(residual.stride(0) if residual is not None else 0),
                  rows_per_batch=rows_per_batch, ld_rowvec=(rowvec.stride(0) if rowvec is not None else 0), flags=flags,
                  alpha=alpha)
    if not (colstats and rows_per_batch > 0 and out.is_contiguous() and _attach_colstats(d, out, n_cols, rows_per_batch)):
        _drop_stale_stats(out)
    q8 = None
    if emit_q8 and not (flags & (L.GEMM_OUT_F32 | L.GEMM_TRANSPOSED | L.GEMM_GEGLU)) and d.colstats is None and rowvec is None:
        q8 = _attach_q8(d, M, n_cols, x.device, emit_rowstats)      # (None: the plan has no emitting epilogue)
    run_gemm(d, x.device)
    if getattr(out, "mx8", None) is not None:
        del out.mx8
    if q8 is not None:
        out.mx8 = q8
    if WORK_COUNTER is not None:
        count_work("gemm", 2.0 * M * N * K)
        count_work("gemm_bytes", 2.0 * (M * K + N * K) + out.numel() * out.element_size()
                   + (2.0 * M * n_cols if residual is not None else 0.0))
        count_work("gemm_launches", 1.0)
    return out


def linear_mx8(x: Mx8Act, wq: torch.Tensor, colscale: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
               out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
               flags: int = 0, n_out: Optional[int] = None, colstats: bool = False, emit_q8: bool = False,
               emit_rowstats: bool = False, ln_c: Optional[torch.Tensor] = None, ln_s: Optional[torch.Tensor] = None,
               eps: float = 1e-5, want_bf16: bool = True, q8_fixed: Optional[tuple] = None):
    """out[M, N] (bf16) = epilogue(dequant(x) @ dequant(wq)^T) on the MX8 path (UDT_GEMM_MX8): x an Mx8Act, wq e4m3 [N, K]
    with per-output-channel fp32 scales.  ln_c / ln_s: the GEMM is LayerNorm-folded (packing.pack_ln_linear_mx8; x.stats holds
    the row statistics its producer emitted).  emit_q8: the result again as an Mx8Act (``out.mx8``; with GEGLU and
    want_bf16=False ONLY that — the function then returns the Mx8Act)."""
    xq = x.data
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and xq.is_contiguous() and wq.is_contiguous()
    M, K = xq.shape
    N = wq.shape[0] if n_out is None else n_out
    assert wq.shape[1] == K and K % 128 == 0 and colscale.dtype == torch.float32 and colscale.numel() >= N
    assert x.scale.dtype == torch.int32 and x.scale.shape == (K // 128, M) and x.scale.is_contiguous()
    geglu = bool(flags & L.GEMM_GEGLU)
    n_cols = N // 2 if geglu else N
    only_q8 = emit_q8 and not want_bf16
    if out is None and not only_q8:
        out = torch.empty((M, n_cols), dtype=torch.bfloat16, device=xq.device)
    d = gemm_desc(a=_ptr(xq), w=_ptr(wq), bias=_ptr(ln_c if ln_c is not None else bias), residual=_ptr(residual),
                  out=_ptr(out), M=M, N=N, K=K, lda=K, ldo=(out.stride(0) if out is not None else 0),
                  ldr=(residual.stride(0) if residual is not None else 0), rows_per_batch=rows_per_batch,
                  flags=flags | L.GEMM_MX8, colscale=_ptr(colscale), a_scale=_ptr(x.scale))
    if ln_s is not None:
        if x.stats is None:
            raise L.UdtError("linear_mx8: a LayerNorm-folded MX8 GEMM needs the row statistics of its input (Mx8Act.stats)")
        assert x.stats.dtype == torch.float32 and x.stats.is_contiguous() and x.stats.shape[1:] == (M, 2)
        d.ln_colsum, d.ln_eps = _ptr(ln_s), eps
        d.rowstat_in, d.rowstat_in_parts = _ptr(x.stats), x.stats.shape[0]
    if out is not None and not (colstats and rows_per_batch > 0 and out.is_contiguous()
                                and _attach_colstats(d, out, n_cols, rows_per_batch)):
        _drop_stale_stats(out)
    q8 = None
    if emit_q8:
        q8 = _attach_q8(d, M, n_cols, xq.device, emit_rowstats and not geglu, fixed=q8_fixed)
        if q8 is None:
            raise L.UdtError(f"udt_gemm has no MX8-emitting plan for M={M} N={N} K={K} flags={flags:#x}")
    run_gemm(d, xq.device)
    if out is not None and getattr(out, "mx8", None) is not None:
        del out.mx8
    if WORK_COUNTER is not None:
        count_work("gemm_fp8", 2.0 * M * N * K)
        count_work("gemm_fp8_launches", 1.0)
    if only_q8:
        return q8
    if q8 is not None:
        out.mx8 = q8
    return out


def ln_linear(x: torch.Tensor, w_folded: torch.Tensor, c: torch.Tensor, s: torch.Tensor, *, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, flags: int = 0,
              n_out: Optional[int] = None, emit_q8: bool = False, want_bf16: bool = True, q8_fixed: Optional[tuple] = None,
              q8_or_none: bool = False):
    """out = epilogue(LayerNorm(x) @ W^T + b) in ONE launch (udt_ln_gemm_fwd): x holds the raw rows, (w_folded, c, s) come
    from packing.pack_ln_linear; the row statistics are taken inside the GEMM (reference attention.py:310-339).
    q8_or_none: with emit_q8, return None (nothing launched) instead of raising when the library has no emitting plan for the shape
    (row-resident kernel switched off through udt_debug_set / UDT_LEAN, unaligned rows) — the caller then takes its bf16 path."""
    _bf16(x); _bf16(w_folded)
    K = x.shape[-1]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x
    assert x2.dim() == 2 and x2.stride(1) == 1 and w_folded.shape[1] == K and w_folded.is_contiguous()
    M = x2.shape[0]
    N = w_folded.shape[0] if n_out is None else n_out
    n_cols = N // 2 if (flags & L.GEMM_GEGLU) else N
    only_q8 = emit_q8 and not want_bf16
    if out is None and not only_q8:
        out = torch.empty((M, n_cols), dtype=torch.bfloat16, device=x.device)
    d = gemm_desc(a=_ptr(x2), w=_ptr(w_folded), bias=_ptr(c), residual=_ptr(residual), out=_ptr(out), M=M, N=N, K=K,
                  lda=x2.stride(0), ldo=(out.stride(0) if out is not None else 0),
                  ldr=(residual.stride(0) if residual is not None else 0), flags=flags, ln_colsum=_ptr(s), ln_eps=eps)
    q8 = None
    if emit_q8:
        # (the row-resident K = 320 kernel's emitting epilogue: q|k|v for the e4m3 self-attention of config #5 at the 64x64 level)
        q8 = _attach_q8(d, M, n_cols, x.device, False, fixed=q8_fixed)
        if q8 is None:
            if q8_or_none:
                return None
            raise L.UdtError(f"udt_ln_gemm_fwd has no MX8-emitting plan for M={M} N={N} K={K}")
    lib = L.load()
    need = lib.udt_gemm_workspace_bytes(C.byref(d))
    ws_ptr, ws_bytes = None, 0
    if need:
        ws = _ws(need, x.device)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    rc = lib.udt_ln_gemm_fwd(C.byref(d), ws_ptr, ws_bytes, _stream())
    if rc in (-1, -2) and x.is_cuda:            # UDT_ERR_BAD_SHAPE / UDT_ERR_BAD_ARG
        # the LayerNorm-folded form exists on the lean kernels only: a problem their plan declines (lean kernels switched off at
        # run time through udt_debug_set, unaligned out / residual pointers, > 2 GiB operands) says so loudly instead of
        # surfacing as a bare status code from inside a transformer block
        raise L.UdtError(f"udt_ln_gemm_fwd declined M={M} N={N} K={K} flags={flags:#x} (status {rc}): the LayerNorm-folded GEMM "
                         "needs the lean kernel family (UDT_LEAN / udt_debug_set('lean') != 0) and 16-byte aligned operands; "
                         "set UDT_LN_GEMM=0 to run layernorm + GEMM instead")
    L.check(rc, "udt_ln_gemm_fwd")
    if WORK_COUNTER is not None:
        count_work("gemm", 2.0 * M * N * K)
        count_work("gemm_bytes", 2.0 * (M * K + N * K) + (2.0 if out is not None else 0.0) * M * n_cols
                   + (1.0 * M * n_cols if q8 is not None else 0.0) + (2.0 * M * n_cols if residual is not None else 0.0))
        count_work("gemm_launches", 1.0)
    if only_q8:
        return q8
    if q8 is not None:
        out.mx8 = q8
    return out


def bmm_nt(a: torch.Tensor, w: torch.Tensor, *, out: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """Batched out[b] = a[b] @ w[b]^T with a [B, M, K], w [B, N, K] bf16 (row-strided views allowed)."""
    _bf16(a); _bf16(w)
    B, M, K = a.shape
    N = w.shape[1]
    assert a.stride(2) == 1 and w.stride(2) == 1 and w.shape[0] == B and w.shape[2] == K
    if out is None:
        out = torch.empty((B, M, N), dtype=torch.bfloat16, device=a.device)
    d = gemm_desc(a=_ptr(a), w=_ptr(w), out=_ptr(out), M=M, N=N, K=K, lda=a.stride(1), ldw=w.stride(1),
                  ldo=out.stride(1), batch=B, stride_a=a.stride(0), stride_w=w.stride(0), stride_out=out.stride(0),
                  alpha=alpha)
    run_gemm(d, a.device)
    if WORK_COUNTER is not None:
        count_work("gemm", 2.0 * B * M * N * K)
        count_work("gemm_bytes", 2.0 * B * (M * K + N * K + M * N))
        count_work("gemm_launches", 1.0)
    return out


def conv2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, ksize: int = 3, stride: int = 1,
           pad: Optional[tuple] = None, upsample: bool = False, x2: Optional[torch.Tensor] = None,
           out_hw: Optional[tuple] = None, residual: Optional[torch.Tensor] = None,
           rowvec: Optional[torch.Tensor] = None, flags: int = 0, n_out: Optional[int] = None,
           out: Optional[torch.Tensor] = None, colstats: bool = False, in_scsh: Optional[torch.Tensor] = None,
           in_act: int = 0, probe_in_scsh: bool = False):
    """NHWC convolution as implicit GEMM.  x [B,H,W,C1] (+ optional x2 [B,H,W,C2], channel concat) bf16;
    w [N, ksize*ksize*(C1+C2)] packed tap-major.  Returns [B,Hout,Wout,N].
    in_scsh / in_act: GroupNorm scale/shift table (gn_finalize) + activation applied to the input on the staged patch
    (udt_gn_silu_conv3x3_fwd); colstats: emit the output's column statistics (``out.gn_stats``);
    probe_in_scsh: no launch — returns whether the library would accept in_scsh for this problem."""
    _bf16(x); _bf16(w)
    assert x.is_contiguous() and w.is_contiguous()
    B, H, W_, C1 = x.shape
    C2 = 0
    if x2 is not None:
        _bf16(x2)
        assert x2.is_contiguous() and x2.shape[:3] == x.shape[:3]
        C2 = x2.shape[3]
    Hv, Wv = (H * 2, W_ * 2) if upsample else (H, W_)
    if pad is None:
        pad = (ksize // 2, ksize // 2)
    if out_hw is None:
        out_hw = ((Hv + 2 * pad[0] - ksize) // stride + 1, (Wv + 2 * pad[1] - ksize) // stride + 1)
    Ho, Wo = out_hw
    N = w.shape[0] if n_out is None else n_out
    K = ksize * ksize * (C1 + C2)
    assert w.shape[1] == K, (w.shape, K)
    M = B * Ho * Wo
    if out is None and not probe_in_scsh:
        dt = torch.float32 if (flags & L.GEMM_OUT_F32) else torch.bfloat16
        out = torch.empty((B, Ho, Wo, N), dtype=dt, device=x.device)
    d = gemm_desc(a=_ptr(x), a2=_ptr(x2), w=_ptr(w), bias=_ptr(bias), residual=_ptr(residual), rowvec=_ptr(rowvec),
                  out=_ptr(out), M=M, N=N, K=K, lda=0, ldo=(out.stride(2) if out is not None else N),
                  ldr=(residual.stride(2) if residual is not None else 0),
                  Hin=H, Win=W_, C1=C1, C2=C2, Hout=Ho, Wout=Wo, ksize=ksize, stride=stride, pad_t=pad[0], pad_l=pad[1],
                  upsample=1 if upsample else 0, rows_per_batch=Ho * Wo,
                  ld_rowvec=(rowvec.stride(0) if rowvec is not None else 0), flags=flags | L.GEMM_CONV)
    if probe_in_scsh:
        return bool(L.load().udt_gemm_in_scsh_ok(C.byref(d)))
    if in_scsh is not None:                       # (before the statistics probe: the plan depends on it)
        assert in_scsh.dtype == torch.float32 and in_scsh.is_contiguous() and in_scsh.numel() == B * (C1 + C2) * 2
        d.in_scsh = in_scsh.data_ptr()
        d.in_act = int(in_act)
    if not (colstats and _attach_colstats(d, out, N, Ho * Wo)):
        _drop_stale_stats(out)
    if in_scsh is not None:
        run_gemm(d, x.device, fused_gn=True)
    else:
        run_gemm(d, x.device)
    return out


# ------------------------------------------------------------------------------------------- attention
def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, scale: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q,k: [B, N, *] row views whose first heads*64 columns are the head-major projections (last dim
    contiguous); vt: [B, heads*64, Nk].  Returns o [B, Nq, heads*64]."""
    _bf16(q); _bf16(k); _bf16(vt)
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1
    if out is None:
        out = torch.empty((B, Nq, heads * 64), dtype=torch.bfloat16, device=q.device)
    L.check(L.load().udt_attn_fwd(_ptr(q), _ptr(k), _ptr(vt), _ptr(out), B, heads, Nq, Nk,
                                  q.stride(1), k.stride(1), vt.stride(1), out.stride(1),
                                  q.stride(0), k.stride(0), vt.stride(0), out.stride(0), scale, _stream()),
            "udt_attn_fwd")
    if WORK_COUNTER is not None:
        count_work("attn", 4.0 * B * heads * Nq * Nk * 64)
        count_work("attn_bytes", 2.0 * B * heads * 64 * (2 * Nq + 2 * Nk))
        count_work("attn_launches", 1.0)
    return out


def attention_rowv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float,
                   out: Optional[torch.Tensor] = None, emit_q8: bool = False) -> torch.Tensor:
    """flash attention with V row-major like K: q, k, v [B, N, *] row views (e.g. the column ranges of one q|k|v
    projection) whose first heads*64 columns are the head-major projections.  Returns o [B, Nq, heads*64].
    emit_q8: o ALSO as an MX8 activation (``out.mx8``) for an e4m3 ``to_out`` (heads even)."""
    _bf16(q); _bf16(k); _bf16(v)
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1 and v.shape[1] == Nk
    if out is None:
        out = torch.empty((B, Nq, heads * 64), dtype=torch.bfloat16, device=q.device)
    if getattr(out, "mx8", None) is not None:
        del out.mx8
    if emit_q8 and heads % 2 == 0 and out.is_contiguous():
        q8 = _mx8_alloc(B * Nq, heads * 64, q.device)
        L.check(L.load().udt_attn_rowv_q8_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Nq, Nk,
                                              q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                              q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale,
                                              _ptr(q8.data), _ptr(q8.scale), heads * 64, _stream()), "udt_attn_rowv_q8_fwd")
        out.mx8 = q8
    else:
        L.check(L.load().udt_attn_rowv_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Nq, Nk,
                                           q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                           q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, _stream()),
                "udt_attn_rowv_fwd")
    if WORK_COUNTER is not None:
        count_work("attn", 4.0 * B * heads * Nq * Nk * 64)
        count_work("attn_bytes", 2.0 * B * heads * 64 * (2 * Nq + 2 * Nk))
        count_work("attn_launches", 1.0)
    return out


def attention_mx8(qkv: Mx8Act, batch: int, heads: int, scale: float, v_mul: float, out: Optional[torch.Tensor] = None,
                  emit_q8: bool = False) -> torch.Tensor:
    """the same self-attention on e4m3 operands (BASELINE config #5's "fp8 attention", udt_attn_mx8_fwd): qkv is the MX8 form of
    one q|k|v projection [B * N, 3 C] whose emitting epilogue wrote the v third with the fixed multiplier v_mul
    (q8_fixed = (2 C, v_mul)).  Returns o [B, N, C] bf16 (``out.mx8``: o again as an MX8 activation, emit_q8)."""
    C_ = heads * 64
    M, ld8 = qkv.data.shape
    assert qkv.data.dtype == torch.uint8 and qkv.data.is_contiguous() and ld8 >= 3 * C_ and M % batch == 0
    assert qkv.scale.dtype == torch.int32 and qkv.scale.is_contiguous() and qkv.scale.shape == ((ld8 + 127) // 128, M)
    N = M // batch
    if out is None:
        out = torch.empty((batch, N, C_), dtype=torch.bfloat16, device=qkv.data.device)
    assert out.is_contiguous()
    if getattr(out, "mx8", None) is not None:
        del out.mx8
    q8 = _mx8_alloc(M, C_, qkv.data.device) if emit_q8 else None
    L.check(L.load().udt_attn_mx8_fwd(_ptr(qkv.data), _ptr(qkv.scale), _ptr(out), batch, heads, N, ld8, out.stride(1), scale,
                                      1.0 / v_mul, _ptr(q8.data) if q8 else None, _ptr(q8.scale) if q8 else None, C_, _stream()),
            "udt_attn_mx8_fwd")
    if q8 is not None:
        out.mx8 = q8
    if WORK_COUNTER is not None:
        count_work("attn_fp8", 4.0 * batch * heads * N * N * 64)
        count_work("attn_fp8_bytes", 1.0 * batch * heads * 64 * 3 * N + 2.0 * batch * heads * 64 * N)
        count_work("attn_fp8_launches", 1.0)
    return out


def attention_d512(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
                   out: Optional[torch.Tensor] = None, key_split: bool = True) -> torch.Tensor:
    """single-head flash attention with head_dim 512 (the VAE mid-block attention): q, k, v [B, N, 512] row views, e.g. the
    column ranges of one q|k|v projection.  Returns o [B, Nq, 512]; no [Nq, Nk] score tensor is materialised.
    ``key_split``: grids too small to fill the CUs (a single image) split the keys over workgroups and merge
    (udt_attn512_split_fwd; the library plans the split, the scratch for the partial results is allocated here)."""
    _bf16(q); _bf16(k); _bf16(v)
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    assert q.shape[2] == 512 and k.shape[2] == 512 and v.shape[2] == 512 and v.shape[1] == Nk
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty((B, Nq, 512), dtype=torch.bfloat16, device=q.device)
    lib = L.load()
    need = int(lib.udt_attn512_workspace_bytes(B, Nq, Nk)) if key_split else 0
    if need:
        scratch = torch.empty((need,), dtype=torch.uint8, device=q.device)
        L.check(lib.udt_attn512_split_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Nq, Nk,
                                          q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                          q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, _ptr(scratch), need, _stream()),
                "udt_attn512_split_fwd")
    else:
        L.check(lib.udt_attn512_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Nq, Nk,
                                    q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                    q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, _stream()),
                "udt_attn512_fwd")
    if WORK_COUNTER is not None:
        count_work("attn", 4.0 * B * Nq * Nk * 512)
        count_work("attn_bytes", 2.0 * B * 512 * (2 * Nq + 2 * Nk))
        count_work("attn_launches", 1.0)
    return out


def xattention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, head_dim: int, scale: float,
               probs: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B, Nq, heads*head_dim]; k, v row views [B, L, *] sharing one row stride; probs fp32 [B*heads, Nq, L]."""
    _bf16(q); _bf16(k); _bf16(v)
    B, Nq = q.shape[0], q.shape[1]
    Lc = k.shape[1]
    assert q.is_contiguous() or (q.stride(2) == 1 and q.stride(0) == Nq * q.stride(1))
    assert k.stride(2) == 1 and v.stride(2) == 1 and k.stride(1) == v.stride(1)
    assert k.stride(0) == Lc * k.stride(1) and v.stride(0) == Lc * v.stride(1)
    if out is None:
        out = torch.empty((B, Nq, heads * head_dim), dtype=torch.bfloat16, device=q.device)
    if probs is not None:
        assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.numel() == B * heads * Nq * Lc
    L.check(L.load().udt_xattn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(probs), B, heads, head_dim, Nq, Lc,
                                   q.stride(1), k.stride(1), out.stride(1), scale, _stream()), "udt_xattn_fwd")
    return out


def masked_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float,
                     mask: Optional[torch.Tensor] = None, key_padding_mask: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MultiheadAttention's core for short sequences (udt_mattn_fwd): q [B, Nq, heads*D], k / v [B, Lk, *] row views
    whose first heads*D columns are the head-major projections; mask fp32 [Nq, Lk] additive; key_padding_mask
    uint8 / bool [B, Lk] (true = ignore).  D = q.shape[-1] // heads, a multiple of 8 up to 64."""
    _bf16(q); _bf16(k); _bf16(v)
    B, Nq, Cq = q.shape
    Lk = k.shape[1]
    D = Cq // heads
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty((B, Nq, heads * D), dtype=torch.bfloat16, device=q.device)
    if mask is not None:
        mask = mask.float().contiguous()
        assert mask.shape == (Nq, Lk)
    if key_padding_mask is not None:
        key_padding_mask = key_padding_mask.to(torch.uint8).contiguous()
        assert key_padding_mask.shape == (B, Lk)
    L.check(L.load().udt_mattn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(mask), _ptr(key_padding_mask), B, heads, D,
                                   Nq, Lk, q.stride(1), k.stride(1), v.stride(1), out.stride(1), Lk,
                                   q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, _stream()), "udt_mattn_fwd")
    return out


class TattnTables(NamedTuple):
    """per-sample tables of the fused text cross-attention (udt_tattn_prepare)"""
    A: torch.Tensor          # bf16 [B, hp, C]
    sc: torch.Tensor         # fp32 [B, hp, 2]
    BmT: torch.Tensor        # bf16 [B, C, hp]

    def rows(self, begin: int, end: Optional[int] = None) -> "TattnTables":
        return TattnTables(self.A[begin:end], self.sc[begin:end], self.BmT[begin:end])


def tattn_prepare(kv: torch.Tensor, wq: torch.Tensor, wo: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, heads: int,
                  scale: float, out: Optional[TattnTables] = None) -> TattnTables:
    """kv bf16 [B, L, 2*C] (hoisted k|v of the context), wq / wo packed bf16 [C, >=C] -> tables (in place into ``out``)"""
    _bf16(kv); _bf16(wq); _bf16(wo)
    B, Lc, two_c = kv.shape
    Cc = heads * 64
    assert two_c == 2 * Cc and kv.is_contiguous() and wq.is_contiguous() and wo.is_contiguous()
    hp = L.load().udt_tattn_hp(heads)
    if out is None:
        out = TattnTables(torch.empty((B, hp, Cc), dtype=torch.bfloat16, device=kv.device),
                          torch.empty((B, hp, 2), dtype=torch.float32, device=kv.device),
                          torch.empty((B, Cc, hp), dtype=torch.bfloat16, device=kv.device))
    L.check(L.load().udt_tattn_prepare(_ptr(kv), two_c, _ptr(wq), wq.stride(0), _ptr(wo), wo.stride(0), _ptr(gamma), _ptr(beta),
                                       _ptr(out.A), _ptr(out.sc), _ptr(out.BmT), B, Lc, Cc, heads, scale, _stream()),
            "udt_tattn_prepare")
    return out


def tattn_fused(x: torch.Tensor, tables: Optional[TattnTables], bias: torch.Tensor, heads: int, zero_samples: int, eps: float,
                out: Optional[torch.Tensor] = None, emit_q8: bool = False) -> torch.Tensor:
    """x bf16 [B, N, C] -> x + t_attn(LayerNorm(x)) (+ bias); the first zero_samples samples see a zero context.
    emit_q8: the result ALSO as an MX8 activation with partial row statistics (``out.mx8``) where the library has the instance."""
    _bf16(x)
    assert x.is_contiguous()
    B, N, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous()
    if tables is not None:
        assert tables.A.is_contiguous() and tables.sc.is_contiguous() and tables.BmT.is_contiguous() and tables.A.shape[0] == B
    if getattr(out, "mx8", None) is not None:
        del out.mx8
    parts = L.load().udt_tattn_rowstat_parts(B, N, Cc) if emit_q8 else 0
    if parts > 0:
        q = _mx8_alloc(B * N, Cc, x.device)
        q = Mx8Act(q.data, q.scale, torch.empty((parts, B * N, 2), dtype=torch.float32, device=x.device))
        L.check(L.load().udt_tattn_fused_q8(_ptr(x), _ptr(out), _ptr(tables.A) if tables else None, _ptr(tables.sc) if tables else None,
                                            _ptr(tables.BmT) if tables else None, _ptr(bias), B, N, Cc, heads, zero_samples, eps,
                                            _ptr(q.data), _ptr(q.scale), _ptr(q.stats), _stream()), "udt_tattn_fused_q8")
        out.mx8 = q
        return out
    L.check(L.load().udt_tattn_fused(_ptr(x), _ptr(out), _ptr(tables.A) if tables else None, _ptr(tables.sc) if tables else None,
                                     _ptr(tables.BmT) if tables else None, _ptr(bias), B, N, Cc, heads, zero_samples, eps, _stream()),
            "udt_tattn_fused")
    return out


def softmax_rows_(x: torch.Tensor) -> torch.Tensor:
    _bf16(x)
    assert x.is_contiguous()
    cols = x.shape[-1]
    L.check(L.load().udt_softmax_rows(_ptr(x), x.numel() // cols, cols, cols, _stream()), "udt_softmax_rows")
    return x


# --------------------------------------------------------------------------------------- normalisation
def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
               out: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 NHWC [B, ..., C1] (+ optional x2 [B, ..., C2] concatenated on channels) -> [B, ..., C1+C2];
    statistics in fp32 over (pixels, channels/groups)."""
    _bf16(x)
    assert x.is_contiguous()
    B, C1 = x.shape[0], x.shape[-1]
    C2 = 0
    if x2 is not None:
        _bf16(x2)
        assert x2.is_contiguous() and x2.shape[:-1] == x.shape[:-1]
        C2 = x2.shape[-1]
    HW = x.numel() // (B * C1)
    lib = L.load()
    if out is None:
        out = torch.empty(x.shape[:-1] + (C1 + C2,), dtype=torch.bfloat16, device=x.device)
    if GN_STRIP and lib.udt_gn_strip_ok(B, HW, C1, C2, groups):      # one launch, second read out of L2
        L.check(lib.udt_gn_strip(_ptr(x), _ptr(x2), _ptr(out), _ptr(gamma), _ptr(beta), B, HW, C1, C2, groups, eps,
                                 1 if silu else 0, _stream()), "udt_gn_strip")
        return out
    nch = lib.udt_gn_nchunks(HW, C1 + C2)
    part = torch.empty((B, nch, groups, 2), dtype=torch.float32, device=x.device)
    L.check(lib.udt_gn_stats(_ptr(x), _ptr(x2), _ptr(part), B, HW, C1, C2, groups, _stream()), "udt_gn_stats")
    L.check(lib.udt_gn_apply(_ptr(x), _ptr(x2), _ptr(out), _ptr(part), _ptr(gamma), _ptr(beta), B, HW, C1, C2, groups,
                             eps, 1 if silu else 0, _stream()), "udt_gn_apply")
    return out


def gn_finalize(st1: GnStats, C1: int, st2: Optional[GnStats], C2: int, gamma: torch.Tensor, beta: torch.Tensor, B: int,
                HW: int, groups: int, eps: float) -> torch.Tensor:
    """producer-epilogue statistics (one or two concatenated sources) -> per-(sample, channel) GroupNorm scale/shift,
    fp32 [B, (C1+C2)/64, 2, 64] (the table udt_gn_silu_conv3x3_fwd reads)"""
    Ct = C1 + C2
    scsh = torch.empty((B, Ct // 64, 2, 64), dtype=torch.float32, device=st1.data.device)
    L.check(L.load().udt_gn_finalize(_ptr(st1.data), st1.slots_per_sample, C1, _ptr(st2.data) if st2 is not None else None,
                                     st2.slots_per_sample if st2 is not None else 0, C2, _ptr(gamma), _ptr(beta), _ptr(scsh),
                                     B, HW, groups, eps, _stream()), "udt_gn_finalize")
    return scsh


def gn_strip_ok(B: int, HW: int, C1: int, C2: int, groups: int) -> bool:
    """would group_norm run this shape as ONE strip launch (statistics + apply)?"""
    return bool(GN_STRIP and L.load().udt_gn_strip_ok(B, HW, C1, C2, groups))


def group_norm_from_stats(x: torch.Tensor, st1: GnStats, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                          silu: bool, x2: Optional[torch.Tensor] = None, st2: Optional[GnStats] = None,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm (+ SiLU) whose statistics came out of the producers' epilogues (``x.gn_stats``): udt_gn_finalize turns them
    into a per-(sample, channel) scale / shift table, udt_gn_apply_scsh is the one read + write left of the norm"""
    _bf16(x)
    assert x.is_contiguous()
    B, C1 = x.shape[0], x.shape[-1]
    C2 = 0
    if x2 is not None:
        _bf16(x2)
        assert x2.is_contiguous() and x2.shape[:-1] == x.shape[:-1] and st2 is not None
        C2 = x2.shape[-1]
    HW = x.numel() // (B * C1)
    if out is None:
        out = torch.empty(x.shape[:-1] + (C1 + C2,), dtype=torch.bfloat16, device=x.device)
    if gn_strip_ok(B, HW, C1, C2, groups):          # small levels: one launch, the strip kernel without its statistics pass
        L.check(L.load().udt_gn_strip_stats(_ptr(x), _ptr(x2), _ptr(out), _ptr(st1.data), st1.slots_per_sample,
                                            _ptr(st2.data) if st2 is not None else None,
                                            st2.slots_per_sample if st2 is not None else 0, _ptr(gamma), _ptr(beta), B, HW, C1, C2,
                                            groups, eps, 1 if silu else 0, _stream()), "udt_gn_strip_stats")
        return out
    scsh = gn_finalize(st1, C1, st2, C2, gamma, beta, B, HW, groups, eps)
    L.check(L.load().udt_gn_apply_scsh(_ptr(x), _ptr(x2), _ptr(out), _ptr(scsh), B, HW, C1, C2, 1 if silu else 0, _stream()),
            "udt_gn_apply_scsh")
    return out


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _bf16(x)
    assert x.is_contiguous()
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().udt_layernorm(_ptr(x), _ptr(out), _ptr(gamma), _ptr(beta), x.numel() // Cc, Cc, eps, _stream()),
            "udt_layernorm")
    return out


# ---------------------------------------------------------------------------------------- elementwise
def unet_input(x: torch.Tensor, xin: torch.Tensor, c_in: float) -> None:
    B, _, h, w = x.shape
    L.check(L.load().udt_unet_input(_ptr(x), _ptr(xin), B, h * w, xin.shape[-1], c_in, _stream()), "udt_unet_input")


def cfg_euler_step(x: torch.Tensor, eps: torch.Tensor, sigma: float, sigma_next: float, scale: float,
                   denoised: Optional[torch.Tensor] = None, c_out: Optional[float] = None) -> None:
    """c_out defaults to -sigma (EpsScaling); pass the quantised value when it differs from sigma."""
    B, _, h, w = x.shape
    assert x.dtype == torch.float32 and eps.dtype == torch.float32 and x.is_contiguous() and eps.is_contiguous()
    L.check(L.load().udt_cfg_euler_step(_ptr(x), _ptr(eps), _ptr(denoised), B, h * w, eps.shape[-1],
                                        -sigma if c_out is None else c_out, sigma, sigma_next, scale, _stream()),
            "udt_cfg_euler_step")


def posterior_sample(moments: torch.Tensor, noise: torch.Tensor, scale: float) -> torch.Tensor:
    """moments fp32 NHWC [B, h, w, ld>=8]; noise fp32 NCHW [B,4,h,w] -> z fp32 NCHW."""
    B, h, w, ld = moments.shape
    assert moments.dtype == torch.float32 and noise.dtype == torch.float32 and moments.is_contiguous()
    z = torch.empty((B, 4, h, w), dtype=torch.float32, device=moments.device)
    L.check(L.load().udt_posterior_sample(_ptr(moments), _ptr(noise.contiguous()), _ptr(z), B, h * w, ld, scale,
                                          _stream()), "udt_posterior_sample")
    return z


def nchw_to_nhwc(x: torch.Tensor, cpad: int, scale: float = 1.0) -> torch.Tensor:
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, Cc, H, W_ = x.shape
    y = torch.empty((B, H, W_, cpad), dtype=torch.bfloat16, device=x.device)
    L.check(L.load().udt_nchw_to_nhwc(_ptr(x), _ptr(y), B, Cc, H * W_, cpad, scale, _stream()), "udt_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, channels: int) -> torch.Tensor:
    assert x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    B, H, W_, ld = x.shape
    y = torch.empty((B, channels, H, W_), dtype=torch.float32, device=x.device)
    L.check(L.load().udt_nhwc_to_nchw(_ptr(x), _ptr(y), B, channels, H * W_, ld, 1 if x.dtype == torch.float32 else 0,
                                      _stream()), "udt_nhwc_to_nchw")
    return y


def nhwc_set_channels(src: torch.Tensor, dst: torch.Tensor, c0: int) -> None:
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.dtype == torch.bfloat16 and dst.is_contiguous()
    B, Cc, H, W_ = src.shape
    L.check(L.load().udt_nhwc_set_channels(_ptr(src), _ptr(dst), B, Cc, H * W_, dst.shape[-1], c0, _stream()),
            "udt_nhwc_set_channels")


def embed_tokens(idx: torch.Tensor, table: torch.Tensor, pe: torch.Tensor) -> torch.Tensor:
    assert idx.dtype == torch.int32 and idx.is_contiguous()
    n_tok = idx.numel()
    Lc, D = pe.shape
    out = torch.empty((n_tok, D), dtype=torch.bfloat16, device=idx.device)
    L.check(L.load().udt_embed_tokens(_ptr(idx), _ptr(table), _ptr(pe), _ptr(out), n_tok, Lc, D, _stream()),
            "udt_embed_tokens")
    return out


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    assert t.dtype == torch.float32 and t.is_contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.bfloat16, device=t.device)
    L.check(L.load().udt_timestep_embedding(_ptr(t), _ptr(out), t.numel(), dim, _stream()), "udt_timestep_embedding")
    return out


def mask_downsample(mask: torch.Tensor) -> torch.Tensor:
    assert mask.dtype == torch.float32 and mask.is_contiguous()
    B, _, H, W_ = mask.shape
    out = torch.empty((B, 1, H // 8, W_ // 8), dtype=torch.float32, device=mask.device)
    L.check(L.load().udt_mask_downsample(_ptr(mask), _ptr(out), B, H, W_, _stream()), "udt_mask_downsample")
    return out


def local_loss_accumulate(probs: torch.Tensor, mask: torch.Tensor, seg_mask: torch.Tensor, gk9: torch.Tensor,
                          loss: torch.Tensor, heads: int, size: int) -> None:
    """loss [n] += per-layer local-loss term of n = probs.shape[0] / heads samples; sample i is scored against
    mask[i % B] / seg_mask[i % B] (B = mask.shape[0]; n a multiple of B: tiled candidates, or the uncond ‖ cond halves)"""
    B = mask.shape[0]
    n = probs.shape[0] // heads
    Lc = probs.shape[-1]
    assert probs.is_contiguous() and loss.is_contiguous() and loss.numel() == n and n % B == 0
    L.check(L.load().udt_local_loss_tiled(_ptr(probs), _ptr(mask), _ptr(seg_mask), _ptr(gk9), _ptr(loss), n, B, heads, size, Lc,
                                          seg_mask.shape[1], mask.shape[2], mask.shape[3], _stream()), "udt_local_loss_tiled")


def add_(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    _bf16(x); _bf16(y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    L.check(L.load().udt_add_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "udt_add_bf16")
    return x


def bias_add(x: torch.Tensor, bias: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[..., c] = x[..., c] + bias[c]  (bf16 rows, fp32 bias)"""
    _bf16(x)
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    assert x.is_contiguous() and out.is_contiguous() and bias.dtype == torch.float32 and bias.numel() >= Cc
    L.check(L.load().udt_bias_add_bf16(_ptr(x), _ptr(bias), _ptr(out), x.numel() // Cc, Cc, _stream()), "udt_bias_add_bf16")
    return out


# ------------------------------------------------------------------------------------------ backward (dX only, csrc/backward.hip)
def attention_bwd(qkv: torch.Tensor, o: torch.Tensor, d_o: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """flash-attention backward of ``attention_rowv(qkv[..., :C], qkv[..., C:2C], qkv[..., 2C:], heads, scale)``:
    qkv bf16 [B, N, 3 C], o / d_o bf16 [B, N, C] -> d(qkv) bf16 [B, N, 3 C] (udt_attn_bwd: two launches, deterministic)"""
    _bf16(qkv); _bf16(o); _bf16(d_o)
    B, N, C3 = qkv.shape
    Cc = heads * 64
    assert C3 == 3 * Cc and qkv.is_contiguous() and o.is_contiguous() and d_o.is_contiguous() and o.shape == (B, N, Cc) == d_o.shape
    dqkv = torch.empty_like(qkv)
    ws = torch.empty((2, B * heads * N), dtype=torch.float32, device=qkv.device)
    es = qkv.element_size()
    L.check(L.load().udt_attn_bwd(qkv.data_ptr(), qkv.data_ptr() + Cc * es, qkv.data_ptr() + 2 * Cc * es, _ptr(o), _ptr(d_o),
                                  dqkv.data_ptr(), dqkv.data_ptr() + Cc * es, dqkv.data_ptr() + 2 * Cc * es, ws[0].data_ptr(),
                                  ws[1].data_ptr(), B, heads, N, C3, Cc, C3, scale, _stream()), "udt_attn_bwd")
    return dqkv


def xattention_bwd(k: torch.Tensor, v: torch.Tensor, probs: torch.Tensor, d_probs: Optional[torch.Tensor],
                   d_o: Optional[torch.Tensor], heads: int, scale: float) -> torch.Tensor:
    """dq of the text cross-attention: k, v row views [B, L, *] (as ``xattention``), probs / d_probs fp32 [B * heads, Nq, L],
    d_o bf16 [B, Nq, heads * 64] -> dq bf16 [B, Nq, heads * 64]"""
    B, Lc = k.shape[0], k.shape[1]
    Nq = probs.shape[1]
    assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.shape == (B * heads, Nq, Lc)
    assert k.stride(2) == 1 and v.stride(2) == 1 and k.stride(1) == v.stride(1) and k.stride(0) == Lc * k.stride(1) == v.stride(0)
    if d_probs is not None:
        assert d_probs.dtype == torch.float32 and d_probs.is_contiguous() and d_probs.shape == probs.shape
    if d_o is not None:
        _bf16(d_o)
        assert d_o.is_contiguous() and d_o.shape == (B, Nq, heads * 64)
    dq = torch.empty((B, Nq, heads * 64), dtype=torch.bfloat16, device=probs.device)
    L.check(L.load().udt_xattn_bwd(_ptr(k), _ptr(v), _ptr(probs), _ptr(d_probs), _ptr(d_o), _ptr(dq), B, heads, 64, Nq, Lc,
                                   k.stride(1), heads * 64, heads * 64, scale, _stream()), "udt_xattn_bwd")
    return dq


def local_loss_bwd(probs: torch.Tensor, mask: torch.Tensor, seg_mask: torch.Tensor, gk9: torch.Tensor, d_probs: torch.Tensor,
                   loss: Optional[torch.Tensor], heads: int, size: int, weight: float) -> None:
    """d_probs += weight * d(local-loss term of this layer) / d probs (and loss += the term): udt_local_loss_bwd"""
    B = mask.shape[0]
    n = probs.shape[0] // heads
    Lc = probs.shape[-1]
    assert probs.is_contiguous() and d_probs.is_contiguous() and d_probs.shape == probs.shape and n % B == 0
    assert d_probs.dtype == torch.float32 and (loss is None or (loss.is_contiguous() and loss.numel() == n))
    scratch = torch.empty((n, seg_mask.shape[1], 2), dtype=torch.float32, device=probs.device)
    L.check(L.load().udt_local_loss_bwd(_ptr(probs), _ptr(mask), _ptr(seg_mask), _ptr(gk9), _ptr(d_probs), _ptr(loss), _ptr(scratch), n, B,
                                        heads, size, Lc, seg_mask.shape[1], mask.shape[2], mask.shape[3], weight, _stream()),
            "udt_local_loss_bwd")


def layer_norm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float = 1e-5,
                   add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX of LayerNorm (statistics recomputed from x) + ``add`` (the gradient arriving over the residual connection)"""
    _bf16(x); _bf16(dy)
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and (add is None or (add.is_contiguous() and add.shape == x.shape))
    Cc = x.shape[-1]
    dx = torch.empty_like(x)
    L.check(L.load().udt_layernorm_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(add), _ptr(dx), x.numel() // Cc, Cc, eps, _stream()),
            "udt_layernorm_bwd")
    return dx


GN_BWD_CHUNKED = os.environ.get("UDT_GN_BWD_CHUNKED", "1") != "0"     # 0: one workgroup per (sample, group) (A/B, tests)


def group_norm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
                   add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX of GroupNorm (+ SiLU) on bf16 NHWC [B, ..., C] (statistics recomputed from x) + ``add``"""
    _bf16(x); _bf16(dy)
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and (add is None or (add.is_contiguous() and add.shape == x.shape))
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    dx = torch.empty_like(x)
    lib = L.load()
    part = torch.empty((2, B, lib.udt_gn_nchunks(HW, Cc), groups, 2), dtype=torch.float32, device=x.device) if GN_BWD_CHUNKED else None
    L.check(lib.udt_gn_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), _ptr(add), _ptr(dx), _ptr(part), B, HW, Cc, groups,
                           eps, 1 if silu else 0, _stream()), "udt_gn_bwd")
    return dx


def geglu(ag: torch.Tensor) -> torch.Tensor:
    """x * gelu(gate) of stored pre-activations ag bf16 [rows, 2 * inner] = [x | gate]"""
    _bf16(ag)
    assert ag.is_contiguous() and ag.dim() == 2
    inner = ag.shape[1] // 2
    out = torch.empty((ag.shape[0], inner), dtype=torch.bfloat16, device=ag.device)
    L.check(L.load().udt_geglu_fwd(_ptr(ag), _ptr(out), ag.shape[0], inner, _stream()), "udt_geglu_fwd")
    return out


def geglu_bwd(ag: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _bf16(ag); _bf16(dy)
    inner = ag.shape[1] // 2
    assert ag.is_contiguous() and dy.is_contiguous() and dy.shape == (ag.shape[0], inner)
    dag = torch.empty_like(ag)
    L.check(L.load().udt_geglu_bwd(_ptr(ag), _ptr(dy), _ptr(dag), ag.shape[0], inner, _stream()), "udt_geglu_bwd")
    return dag


def sum2x2(dy: torch.Tensor) -> torch.Tensor:
    """nearest x2 upsampling backward: bf16 NHWC [B, 2H, 2W, C] -> [B, H, W, C]"""
    _bf16(dy)
    assert dy.is_contiguous() and dy.shape[1] % 2 == 0 and dy.shape[2] % 2 == 0
    B, H2, W2, Cc = dy.shape
    dx = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=torch.bfloat16, device=dy.device)
    L.check(L.load().udt_sum2x2_bf16(_ptr(dy), _ptr(dx), B, H2 // 2, W2 // 2, Cc, _stream()), "udt_sum2x2_bf16")
    return dx


def center_tokens(ctx: torch.Tensor) -> torch.Tensor:
    """fp32 [B, L, D] -> bf16 [B, L, D]: every token minus the mean over the sample's L tokens (udt_center_tokens)"""
    assert ctx.dtype == torch.float32 and ctx.is_contiguous() and ctx.dim() == 3
    B, Lc, D = ctx.shape
    out = torch.empty((B, Lc, D), dtype=torch.bfloat16, device=ctx.device)
    L.check(L.load().udt_center_tokens(_ptr(ctx), _ptr(out), B, Lc, D, _stream()), "udt_center_tokens")
    return out


def axpy_(x: torch.Tensor, y: torch.Tensor, a: float) -> torch.Tensor:
    """x += a * y (fp32, in place)"""
    assert x.dtype == torch.float32 == y.dtype and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    L.check(L.load().udt_axpy_f32(_ptr(x), _ptr(y), a, x.numel(), _stream()), "udt_axpy_f32")
    return x


# ------------------------------------------------------------------------------------------ training step (csrc/backward.hip)
def transpose(x: torch.Tensor) -> torch.Tensor:
    """bf16 [R, C] (row-strided view allowed) -> [C, Rp], Rp = R rounded up to 64, zero-padded (operand of a dW = dY^T X GEMM)"""
    _bf16(x)
    assert x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    Rp = (R + 63) // 64 * 64
    out = torch.empty((Cc, Rp), dtype=torch.bfloat16, device=x.device)
    L.check(L.load().udt_transpose_bf16(_ptr(x), _ptr(out), R, Cc, x.stride(0), Rp, _stream()), "udt_transpose_bf16")
    return out


WGRAD_KERNEL = os.environ.get("UDT_WGRAD_KERNEL", "1") != "0"     # 0: transposes + the forward GEMM (A/B, tests)


def weight_grad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW fp32 [N, K] = dy^T x for dy bf16 [M, N], x bf16 [M, K] (nn.Linear: y = x W^T), contraction over the M rows with fp32
    accumulation and output: udt_wgrad_bf16 straight from the row-major operands (or, switched off / for shapes it does not take, the
    forward GEMM on the two operands transposed)"""
    _bf16(dy); _bf16(x)
    assert dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.stride(1) == 1 and x.stride(1) == 1
    R, N = dy.shape
    K = x.shape[1]
    lib = L.load()
    if not WGRAD_KERNEL or N % 8 or K % 8 or dy.stride(0) % 8 or x.stride(0) % 8 or (dy.data_ptr() | x.data_ptr()) & 15:
        return linear(transpose(dy), transpose(x), None, flags=L.GEMM_OUT_F32)
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    S = lib.udt_wgrad_splits(R, N, K)
    part = torch.empty((S, N, K), dtype=torch.float32, device=dy.device) if S > 1 else None
    L.check(lib.udt_wgrad_bf16(_ptr(dy), _ptr(x), _ptr(dw), _ptr(part), R, N, K, dy.stride(0), x.stride(0), _stream()), "udt_wgrad_bf16")
    return dw


def colsum(x: torch.Tensor) -> torch.Tensor:
    """fp32 [C] column sums of bf16 [rows, C] (bias gradient)"""
    _bf16(x)
    assert x.is_contiguous() and x.dim() == 2
    rows, Cc = x.shape
    lib = L.load()
    part = torch.empty((lib.udt_colparts(rows), Cc), dtype=torch.float32, device=x.device)
    out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    L.check(lib.udt_colsum_bf16(_ptr(x), _ptr(part), _ptr(out), rows, Cc, _stream()), "udt_colsum_bf16")
    return out


def layer_norm_param_grad(x: torch.Tensor, dy: torch.Tensor, eps: float = 1e-5):
    """(d gamma, d beta) fp32 [C] each of LayerNorm over bf16 rows x with output cotangent dy"""
    _bf16(x); _bf16(dy)
    assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape and x.dim() == 2
    rows, Cc = x.shape
    lib = L.load()
    part = torch.empty((lib.udt_colparts(rows), 2, Cc), dtype=torch.float32, device=x.device)
    out = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    L.check(lib.udt_ln_param_grad(_ptr(x), _ptr(dy), _ptr(part), _ptr(out), rows, Cc, eps, _stream()), "udt_ln_param_grad")
    return out[0], out[1]


def xattention_bwd_kv(q: torch.Tensor, v: torch.Tensor, probs: torch.Tensor, d_probs: Optional[torch.Tensor],
                      d_o: Optional[torch.Tensor], heads: int, scale: float):
    """(dk, dv) bf16 [B, L, heads * 64] of the text cross-attention: q bf16 [B, Nq, C], v a row view [B, L, *]"""
    _bf16(q)
    B, Nq, Cc = q.shape
    Lc = v.shape[1]
    assert q.is_contiguous() and v.stride(2) == 1 and v.stride(0) == Lc * v.stride(1) and probs.shape == (B * heads, Nq, Lc)
    dk = torch.empty((B, Lc, Cc), dtype=torch.bfloat16, device=q.device)
    dv = torch.empty_like(dk)
    lib = L.load()
    part = torch.empty((lib.udt_xattn_kv_splits(Nq), B * Lc, Cc, 2), dtype=torch.float32, device=q.device)
    L.check(lib.udt_xattn_bwd_kv(_ptr(q), _ptr(v), _ptr(probs), _ptr(d_probs), _ptr(d_o), _ptr(dk), _ptr(dv), _ptr(part), B, heads, 64, Nq,
                                 Lc, Cc, v.stride(1), Cc, Cc, scale, _stream()), "udt_xattn_bwd_kv")
    return dk, dv


def local_loss_seg_bwd(probs: torch.Tensor, seg: torch.Tensor, seg_mask: torch.Tensor, gk9: torch.Tensor, d_probs: torch.Tensor,
                       loss: Optional[torch.Tensor], heads: int, size: int, weight: float) -> None:
    """FullLoss.get_local_loss per layer: d_probs += weight * d f / d probs, loss[b] += f_b (udt_local_loss_seg_bwd)"""
    B = seg.shape[0]
    assert probs.is_contiguous() and d_probs.is_contiguous() and d_probs.shape == probs.shape and probs.shape[0] == B * heads
    assert seg.dtype == torch.float32 and seg.is_contiguous() and seg_mask.is_contiguous() and seg.shape[1] == seg_mask.shape[1]
    scratch = torch.empty((B, seg.shape[1]), dtype=torch.float32, device=probs.device)
    L.check(L.load().udt_local_loss_seg_bwd(_ptr(probs), _ptr(seg), _ptr(seg_mask), _ptr(gk9), _ptr(d_probs), _ptr(loss), _ptr(scratch), B,
                                            heads, size, probs.shape[-1], seg.shape[1], seg.shape[2], seg.shape[3], weight, _stream()),
            "udt_local_loss_seg_bwd")


def diff_loss_grad(eps: torch.Tensor, noised: torch.Tensor, target: torch.Tensor, sigma: torch.Tensor, cpad: int = 64):
    """(loss fp32 [B], d_eps bf16 NHWC [B, h, w, cpad]) of the eps-prediction loss: eps fp32 NHWC [B, h, w, ld], noised / target fp32 NCHW"""
    B, h, w, ld = eps.shape
    assert eps.dtype == torch.float32 and eps.is_contiguous() and noised.is_contiguous() and target.is_contiguous()
    assert noised.shape == (B, 4, h, w) == target.shape and sigma.dtype == torch.float32 and sigma.numel() == B
    d_eps = torch.empty((B, h, w, cpad), dtype=torch.bfloat16, device=eps.device)
    loss = torch.empty((B,), dtype=torch.float32, device=eps.device)
    L.check(L.load().udt_diff_loss_grad(_ptr(eps), _ptr(noised), _ptr(target), _ptr(sigma), _ptr(d_eps), _ptr(loss), B, h * w, ld, cpad,
                                        _stream()), "udt_diff_loss_grad")
    return loss, d_eps


def adamw_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
           weight_decay: float = 1e-2, grad_scale: float = 1.0) -> None:
    """torch.optim.AdamW's update of one fp32 tensor, in place (udt_adamw_f32); bumps the tensor's version so that cached packed
    layouts and captured graphs notice"""
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    L.check(L.load().udt_adamw_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step,
                                   grad_scale, _stream()), "udt_adamw_f32")
    torch._C._increment_version(p)


# ------------------------------------------------------------------------------------------ profiling
def prof_enable(mask: int) -> None:
    L.check(L.load().udt_prof_enable(mask), "udt_prof_enable")


def prof_reset() -> None:
    L.check(L.load().udt_prof_reset(), "udt_prof_reset")


def prof_get(op_class: int):
    ms = C.c_double(0.0)
    n = C.c_int64(0)
    L.check(L.load().udt_prof_get(op_class, C.byref(ms), C.byref(n)), "udt_prof_get")
    return ms.value, n.value
