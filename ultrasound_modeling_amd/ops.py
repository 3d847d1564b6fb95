"""Thin torch-tensor front end of the C ABI (include/usseg.h).

torch is used for device memory and the current HIP stream only; every function here enqueues
hand-written gfx950 kernels from libusseg_hip.so and raises if the library is missing.

Activation tensors are bf16 ``[B,H,W,Cphys]`` (NHWC) or channel-slice views of such tensors
(``t[..., a:b]`` with a, b multiples of 8); the channel stride ``ld`` is taken from the view.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import (ACCUMULATE, ACT_ELU, ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, OUT_F32, CardinalDesc, ConvDesc, GemmDesc, LossDesc, NormDesc,
                   SplitAttnDesc, SplitAttnGrads, SplitAttnParams)

BF16 = torch.bfloat16
ACC_FLOATS = 2050   # USSEG_ACC_FLOATS: a reproducible scalar accumulator ([0] result, [1] ticket, [2..] per-workgroup partials)


def roundup(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_REDUCE_WS = {}
_WGRAD_WS = {}
WGRAD_WS_FLOATS = 64 << 20   # 256 MB of the 288 GB: split-K partial slabs of the 3x3 weight-gradient kernel


# ------------------------------------------------------------------------------------------------ side stream
# Weight gradients never feed the backward-data chain, so inside an ``overlap_region`` (the model's backward pass) they
# are enqueued on a second HIP stream and run beside the dgrad / norm-backward kernels of the main stream (which leave
# most CUs under-occupied at batch 16).  Both streams are captured into the same HIP graph (fork by event, join at the
# end of the region).  Workspaces are per stream; the tensors a side launch reads are kept alive until the join so the
# caching allocator cannot hand their memory to a later main-stream tensor.
# Off by default since the conv / wgrad kernels got ~1.5x faster: the fork/join edges of the captured graph now cost more
# than the overlap returns (Arch B 4.17 vs 4.23 ms, Arch A 9.47 vs 9.87 ms per step; USSEG_SIDE_STREAM=1 turns it back on).
class _Side:
    enabled = os.environ.get("USSEG_SIDE_STREAM", "0") != "0"
    defer = os.environ.get("USSEG_DEFER", "1") != "0"
    depth = 0
    stream = None
    keep = []
    dirty = False
    deferring = set()      # stream handles with an open deferral context
    post = []              # callbacks that run after the region's final flush (ops.after_flush)
    bufs = {}


DEFER_REDUCE_FLOATS = 32 << 20    # 128 MB of partial rows per stream
DEFER_WGRAD_FLOATS = 256 << 20    # 1 GB of split-K slabs per stream (of 288 GB): the whole backward pass between flushes


def _defer_begin():
    """Queue the finishing reductions of the current stream (usseg_defer_begin) until _defer_end."""
    if not _Side.defer:
        return
    h = _stream()
    if h in _Side.deferring:
        return
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream() == _Side.stream)
    if key not in _Side.bufs:
        _Side.bufs[key] = (torch.empty(DEFER_REDUCE_FLOATS, dtype=torch.float32, device="cuda"),
                           torch.empty(DEFER_WGRAD_FLOATS, dtype=torch.float32, device="cuda"))
    r, w = _Side.bufs[key]
    L.check(L.load().usseg_defer_begin(h, r.data_ptr(), r.numel(), w.data_ptr(), w.numel()), "defer_begin")
    _Side.deferring.add(h)


def _defer_end():
    h = _stream()
    if h in _Side.deferring:
        _Side.deferring.discard(h)
        L.check(L.load().usseg_defer_end(h), "defer_end")


def defer_flush():
    """Run the queued finishing reductions of the current stream now (their gradients are about to be read)."""
    if _stream() in _Side.deferring:
        L.check(L.load().usseg_defer_flush(_stream()), "defer_flush")


def after_flush(fn):
    """Run ``fn`` once the finishing reductions queued so far have executed: at the end of the enclosing ``overlap_region``
    (after its final flush - no extra finishing launches in the middle of the backward pass), or right away when nothing is
    being deferred."""
    if _Side.depth > 0 and _stream() in _Side.deferring and not (_Side.enabled and _Side.stream is not None and torch.cuda.current_stream() == _Side.stream):
        _Side.post.append(fn)
    else:
        defer_flush()
        fn()


@contextlib.contextmanager
def overlap_region():
    _Side.depth += 1
    if _Side.depth == 1:
        _defer_begin()
    try:
        yield
    finally:
        _Side.depth -= 1
        if _Side.depth == 0:
            side_join()
            _defer_end()
            post, _Side.post = _Side.post, []
            for fn in post:
                fn()


@contextlib.contextmanager
def side_stream(*keep):
    """Enqueue the enclosed launches on the side stream, ordered after everything already on the current stream."""
    if not (_Side.enabled and _Side.depth > 0):
        yield
        return
    if _Side.stream is None:
        _Side.stream = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    _Side.stream.wait_event(ev)
    _Side.keep.extend(keep)
    _Side.dirty = True
    with torch.cuda.stream(_Side.stream):
        _defer_begin()
        yield


# Lazy weight gradients (USSEG_LAZY_WGRAD, default on): inside ``lazy_wgrads()`` the weight-gradient launches handed to
# ``wgrad_later`` are only collected; at the end of the block they all run on the side stream behind ONE fork, beside whatever the
# main stream does next (the encoder's backward pass), and join at the end of the overlap region.  The per-layer fork of
# ``side_stream`` (USSEG_SIDE_STREAM=1) costs a cross-stream edge per launch, which made it slower than a single stream.
_LAZY = os.environ.get("USSEG_LAZY_WGRAD", "1") != "0"
_lazy_q = None


@contextlib.contextmanager
def lazy_wgrads():
    global _lazy_q
    if not _LAZY or _Side.depth == 0 or _lazy_q is not None:
        yield
        return
    _lazy_q = []
    try:
        yield
    finally:
        q, _lazy_q = _lazy_q, None
        if q:
            enabled, _Side.enabled = _Side.enabled, True
            try:
                with side_stream(*[t for _, keep in q for t in keep]):
                    for fn, _ in q:
                        fn()
            finally:
                _Side.enabled = enabled


def wgrad_later(fn, *keep):
    """Run ``fn`` (weight-gradient launches reading the tensors ``keep``) now - on the side stream if that is enabled - or, inside
    ``lazy_wgrads()``, at the end of that block."""
    if _lazy_q is not None:
        _lazy_q.append((fn, keep))
        return
    with side_stream(*keep):
        fn()


def side_join():
    if _Side.dirty:
        with torch.cuda.stream(_Side.stream):
            _defer_end()
        ev = torch.cuda.Event()
        ev.record(_Side.stream)
        torch.cuda.current_stream().wait_event(ev)
        _Side.dirty = False
    _Side.keep.clear()


def _ws_key(device):
    on_side = _Side.stream is not None and torch.cuda.current_stream() == _Side.stream
    return (str(device), on_side)


def wgrad_ws(device) -> torch.Tensor:
    key = _ws_key(device)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = torch.empty(WGRAD_WS_FLOATS, dtype=torch.float32, device=device)
        _WGRAD_WS[key] = ws
    return ws



def reduce_ws(device) -> torch.Tensor:
    """fp32 workspace for the library's two-pass per-channel reductions (one per device and stream; stream order serialises its users)."""
    key = _ws_key(device)
    ws = _REDUCE_WS.get(key)
    if ws is None:
        ws = torch.empty(int(L.load().usseg_reduce_ws_floats()), dtype=torch.float32, device=device)
        _REDUCE_WS[key] = ws
    return ws


def geom(t: torch.Tensor) -> Tuple[int, int, int, int, int]:
    """(B,H,W,C,ld) of an NHWC tensor or channel-slice view; validates the layout."""
    assert t.dim() == 4, f"expected NHWC tensor, got shape {tuple(t.shape)}"
    B, H, W, Cc = t.shape
    assert t.stride(3) == 1 or Cc == 1, "channels must be contiguous"
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if B > 1 else Cc))
    if W > 1 and H > 1:
        assert t.stride(1) == W * ld, "rows must be dense"
    if B > 1:
        assert t.stride(0) == H * W * ld, "images must be dense"
    return B, H, W, Cc, ld


def new_act(B: int, H: int, W: int, Cphys: int, device, zero: bool = False) -> torch.Tensor:
    f = torch.zeros if zero else torch.empty
    return f((B, H, W, Cphys), dtype=BF16, device=device)


def _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation=1, act=ACT_NONE, alpha=0.0, flags=0) -> ConvDesc:
    return ConvDesc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation, act, alpha, flags)


# ------------------------------------------------------------------------------------------------ conv
def conv2d_fwd(x, wp, bias, ksize, dilation, out, act=ACT_NONE, alpha=0.0, residual=None, out_f32=False, scale=None):
    """``scale`` (fp32 per output channel): y = act(scale*conv + bias) - the folded inference BatchNorm (bias = its shift)."""
    B, H, W, Cin, ldx = geom(x)
    Bo, Ho, Wo, Cout, ldy = geom(out)
    assert (Bo, Ho, Wo) == (B, H, W)
    assert x.dtype == BF16 and out.dtype == (torch.float32 if out_f32 else BF16)
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation, act, alpha, OUT_F32 if out_f32 else 0)
    ldr = geom(residual)[4] if residual is not None else 0
    if scale is not None:
        L.check(L.load().usseg_conv2d_fwd_affine(C.byref(d), x.data_ptr(), wp.data_ptr(), scale.data_ptr(), bias.data_ptr(), _ptr(residual),
                                                 ldr, out.data_ptr(), _stream()), "conv2d_fwd_affine")
        return out
    L.check(L.load().usseg_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), _ptr(bias), _ptr(residual), ldr,
                                      out.data_ptr(), _stream()), "conv2d_fwd")
    return out


def conv2d_dgrad(dy, wp_d, ksize, dilation, dx, residual=None, accumulate=False):
    B, H, W, Cout, ldy = geom(dy)
    _, _, _, Cin, ldx = geom(dx)
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation, flags=ACCUMULATE if accumulate else 0)
    ldr = geom(residual)[4] if residual is not None else 0
    L.check(L.load().usseg_conv2d_dgrad(C.byref(d), dy.data_ptr(), wp_d.data_ptr(), _ptr(residual), ldr, dx.data_ptr(),
                                        _stream()), "conv2d_dgrad")
    return dx


def conv2d_fwd_multi(jobs):
    """jobs: list of (x, wp, bias, ksize, dilation, out, act, alpha) of INDEPENDENT convs -> one launch where possible."""
    arr = (L.ConvJob * len(jobs))()
    for j, job in enumerate(jobs):
        x, wp, bias, ksize, dilation, out, act, alpha = job[:8]
        arr[j].scale = _ptr(job[8]) if len(job) > 8 else None      # optional folded-BatchNorm multiplier (bias = its shift)
        B, H, W, Cin, ldx = geom(x)
        Bo, Ho, Wo, Cout, ldy = geom(out)
        assert (Bo, Ho, Wo) == (B, H, W) and x.dtype == BF16 and out.dtype == BF16
        arr[j].desc = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation, act, alpha, 0)
        arr[j].x, arr[j].wp, arr[j].bias, arr[j].residual, arr[j].ldr, arr[j].y = x.data_ptr(), wp.data_ptr(), _ptr(bias), None, 0, out.data_ptr()
    L.check(L.load().usseg_conv2d_fwd_multi(len(jobs), C.addressof(arr), _stream()), "conv2d_fwd_multi")


def conv2d_dgrad_multi(jobs):
    """jobs: list of (dy, wp_d, ksize, dilation, dx, residual, accumulate) writing DISJOINT dx buffers -> one launch where possible."""
    arr = (L.ConvJob * len(jobs))()
    for j, (dy, wp_d, ksize, dilation, dx, residual, accumulate) in enumerate(jobs):
        B, H, W, Cout, ldy = geom(dy)
        _, _, _, Cin, ldx = geom(dx)
        arr[j].desc = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation, flags=ACCUMULATE if accumulate else 0)
        arr[j].x, arr[j].wp, arr[j].bias, arr[j].y = dy.data_ptr(), wp_d.data_ptr(), None, dx.data_ptr()
        arr[j].residual, arr[j].ldr = _ptr(residual), (geom(residual)[4] if residual is not None else 0)
    L.check(L.load().usseg_conv2d_dgrad_multi(len(jobs), C.addressof(arr), _stream()), "conv2d_dgrad_multi")


def conv2d_wgrad(x, dy, ksize, dilation, dw: torch.Tensor):
    """dw (fp32, [ntaps, Cin_phys, Cout_phys], pre-zeroed or holding a running sum) += x^T dy."""
    B, H, W, Cin, ldx = geom(x)
    _, _, _, Cout, ldy = geom(dy)
    assert dw.dtype == torch.float32 and dw.numel() == ksize * ksize * Cin * Cout
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation)
    ws = wgrad_ws(x.device)
    L.check(L.load().usseg_conv2d_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
            "conv2d_wgrad")
    return dw


def wgrad_dst(blocks) -> "L.WgradDst":
    """blocks: list of (dst tensor, sT, sI, sO, i_off, o_off, ni, no) -> destination map of a mapped weight gradient."""
    m = L.WgradDst()
    m.nblocks = len(blocks)
    for b, (dst, sT, sI, sO, i_off, o_off, ni, no) in enumerate(blocks):
        assert dst.dtype == torch.float32
        m.blk[b].dst, m.blk[b].sT, m.blk[b].sI, m.blk[b].sO = dst.data_ptr(), sT, sI, sO
        m.blk[b].i_off, m.blk[b].o_off, m.blk[b].ni, m.blk[b].no = i_off, o_off, ni, no
    return m


def conv2d_wgrad_mapped(x, dy, ksize, dilation, dst_map):
    """dW (physical [ntaps][Cin_phys][Cout_phys]) accumulated straight into the variables named by ``dst_map`` (no scratch pass)."""
    B, H, W, Cin, ldx = geom(x)
    _, _, _, Cout, ldy = geom(dy)
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation)
    ws = wgrad_ws(x.device)
    L.check(L.load().usseg_conv2d_wgrad_mapped(C.byref(d), x.data_ptr(), dy.data_ptr(), C.byref(dst_map), ws.data_ptr(), ws.numel(),
                                               _stream()), "conv2d_wgrad_mapped")


def conv2d_dgrad_branches(dy_cat, wp_cat, ksizes, dilations, ch_offs, Cb, dx, residual=None, accumulate=False):
    """dx = sum over the parallel branches of their backward-data passes (one implicit GEMM over every branch's taps)."""
    B, H, W, _, ldy = geom(dy_cat)
    _, _, _, Cin, ldx = geom(dx)
    d = _conv_desc(B, H, W, Cin, Cb, ldx, ldy, 3, 1, flags=ACCUMULATE if accumulate else 0)
    n = len(ksizes)
    arr = lambda v: (C.c_int32 * n)(*v)
    ldr = geom(residual)[4] if residual is not None else 0
    L.check(L.load().usseg_conv2d_dgrad_branches(C.byref(d), n, arr(ksizes), arr(dilations), arr(ch_offs), Cb, dy_cat.data_ptr(),
                                                 wp_cat.data_ptr(), _ptr(residual), ldr, dx.data_ptr(), _stream()), "conv2d_dgrad_branches")
    return dx


def conv2d_wgrad_multi(jobs):
    """jobs: list of (x, dy, ksize, dilation, dw or None, dst_map or None): independent weight gradients, one launch where possible."""
    arr = (L.WgradJob * len(jobs))()
    for j, (x, dy, ksize, dilation, dw, dst_map) in enumerate(jobs):
        B, H, W, Cin, ldx = geom(x)
        _, _, _, Cout, ldy = geom(dy)
        arr[j].desc = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, dilation)
        arr[j].x, arr[j].dy, arr[j].dw = x.data_ptr(), dy.data_ptr(), _ptr(dw)
        arr[j].dst = C.pointer(dst_map) if dst_map is not None else None
    ws = wgrad_ws(jobs[0][0].device)
    L.check(L.load().usseg_conv2d_wgrad_multi(len(jobs), C.addressof(arr), ws.data_ptr(), ws.numel(), _stream()), "conv2d_wgrad_multi")


def tconv2d_wgrad_mapped(x, dy, ksize, dst_map):
    B, H, W, Cin, ldx = geom(x)
    _, _, _, Cout, ldy = geom(dy)
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, 1)
    ws = wgrad_ws(x.device)
    L.check(L.load().usseg_tconv2d_wgrad_mapped(C.byref(d), x.data_ptr(), dy.data_ptr(), C.byref(dst_map), ws.data_ptr(), ws.numel(),
                                                _stream()), "tconv2d_wgrad_mapped")


def tconv2d_fwd(x, wp, bias, ksize, out, act=ACT_NONE, alpha=0.0, out_f32=False):
    B, H, W, Cin, ldx = geom(x)
    Bo, Ho, Wo, Cout, ldy = geom(out)
    assert (Bo, Ho, Wo) == (B, 2 * H, 2 * W)
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, 1, act, alpha, OUT_F32 if out_f32 else 0)
    L.check(L.load().usseg_tconv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), _ptr(bias), out.data_ptr(), _stream()),
            "tconv2d_fwd")
    return out


def tconv2d_dgrad(dy, wp_d, ksize, dx, residual=None, accumulate=False):
    B, H2, W2, Cout, ldy = geom(dy)
    Bx, H, W, Cin, ldx = geom(dx)
    assert (H2, W2) == (2 * H, 2 * W)
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, 1, flags=ACCUMULATE if accumulate else 0)
    ldr = geom(residual)[4] if residual is not None else 0
    L.check(L.load().usseg_tconv2d_dgrad(C.byref(d), dy.data_ptr(), wp_d.data_ptr(), _ptr(residual), ldr, dx.data_ptr(),
                                         _stream()), "tconv2d_dgrad")
    return dx


def tconv2d_wgrad(x, dy, ksize, dw: torch.Tensor):
    """dw (fp32, [ntaps, Cin_phys, Cout_phys]) += sum_i x[i] (x) dy[2i+k-pad]."""
    B, H, W, Cin, ldx = geom(x)
    _, _, _, Cout, ldy = geom(dy)
    assert dw.dtype == torch.float32 and dw.numel() == ksize * ksize * Cin * Cout
    d = _conv_desc(B, H, W, Cin, Cout, ldx, ldy, ksize, 1)
    L.check(L.load().usseg_tconv2d_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _stream()), "tconv2d_wgrad")
    return dw


def pack_weight(src: torch.Tensor, sT, sN, sK, T, Nn, Kk, dst: torch.Tensor, Kw, tap_stride, n_off=0, k_off=0):
    assert src.dtype == torch.float32 and dst.dtype == BF16
    L.check(L.load().usseg_pack_weight(src.data_ptr(), sT, sN, sK, T, Nn, Kk, dst.data_ptr(), Kw, tap_stride, n_off, k_off,
                                       _stream()), "pack_weight")


PACK_JOB_DTYPE = [("src", "<u8"), ("dst", "<u8"), ("sT", "<i8"), ("sN", "<i8"), ("sK", "<i8"), ("T", "<i4"), ("Nn", "<i4"), ("Kk", "<i4"),
                  ("Kw", "<i4"), ("tap_stride", "<i4"), ("n_off", "<i4"), ("k_off", "<i4"), ("reserved", "<i4"), ("nscale", "<u8")]


def pack_job(src, sT, sN, sK, T, Nn, Kk, dst, Kw, tap_stride, n_off=0, k_off=0, nscale=None):
    """Descriptor tuple of one pack (same arguments as pack_weight) for pack_weights_batched.  ``nscale``: optional fp32 vector,
    one multiplier per row n (an inference BatchNorm folded into the forward operand)."""
    assert src.dtype == torch.float32 and dst.dtype == BF16
    assert nscale is None or (nscale.dtype == torch.float32 and nscale.numel() >= Nn)
    return (src.data_ptr(), dst.data_ptr(), sT, sN, sK, T, Nn, Kk, Kw, tap_stride, n_off, k_off, 0, 0 if nscale is None else nscale.data_ptr())


def make_pack_table(jobs, device) -> torch.Tensor:
    import numpy as np
    arr = np.array(jobs, dtype=PACK_JOB_DTYPE)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


BN_FOLD_JOB_DTYPE = [("gamma", "<u8"), ("beta", "<u8"), ("mean", "<u8"), ("var", "<u8"), ("bias", "<u8"), ("scale", "<u8"), ("shift", "<u8"),
                     ("C", "<i4"), ("Cp", "<i4"), ("eps", "<f4"), ("reserved", "<i4")]


def bn_fold_job(gamma, beta, mean, var, bias, scale, shift, C_logical, eps):
    """Descriptor of one folded inference BatchNorm: scale = gamma*rsqrt(var+eps), shift = beta - mean*scale + scale*bias."""
    return (gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), bias.data_ptr() if bias is not None else 0,
            scale.data_ptr(), shift.data_ptr(), C_logical, scale.numel(), eps, 0)


def make_bn_fold_table(jobs, device) -> torch.Tensor:
    import numpy as np
    arr = np.array(jobs, dtype=BN_FOLD_JOB_DTYPE)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


def bn_fold_batched(table: torch.Tensor, njobs: int):
    L.check(L.load().usseg_bn_fold_batched(table.data_ptr(), njobs, _stream()), "bn_fold_batched")


def make_pack_tilemap(jobs, device, tiles_per_block: int = 4):
    """(job, first tile, count) triples covering every 32x32 tile of every pack job once -> (int32 device tensor, nblocks)."""
    import numpy as np
    trip = []
    for ji, job in enumerate(jobs):
        T, Nn, Kk = job[5], job[6], job[7]
        n = T * ((Nn + 31) // 32) * ((Kk + 31) // 32)
        for t0 in range(0, n, tiles_per_block):
            trip.append((ji, t0, min(tiles_per_block, n - t0)))
    arr = np.array(trip, dtype=np.int32).reshape(-1)
    return torch.from_numpy(arr).to(device), len(trip)


def pack_weights_batched(table: torch.Tensor, njobs: int, tilemap=None):
    if tilemap is not None:
        L.check(L.load().usseg_pack_weights_flat(table.data_ptr(), tilemap[0].data_ptr(), tilemap[1], _stream()), "pack_weights_flat")
        return
    L.check(L.load().usseg_pack_weights_batched(table.data_ptr(), njobs, _stream()), "pack_weights_batched")


def unpack_wgrad(scratch, Mrows, Ncols, T, Nn, Kk, n_off, k_off, dst, sT, sN, sK, scale=1.0, accumulate=True):
    L.check(L.load().usseg_unpack_wgrad(scratch.data_ptr(), Mrows, Ncols, T, Nn, Kk, n_off, k_off, dst.data_ptr(), sT, sN, sK,
                                        scale, 1 if accumulate else 0, _stream()), "unpack_wgrad")


UNPACK_JOB_DTYPE = [("scratch", "<u8"), ("dst", "<u8"), ("sT", "<i8"), ("sN", "<i8"), ("sK", "<i8"), ("Mrows", "<i4"), ("Ncols", "<i4"), ("T", "<i4"),
                    ("Nn", "<i4"), ("Kk", "<i4"), ("n_off", "<i4"), ("k_off", "<i4"), ("accumulate", "<i4"), ("scale", "<f4"), ("reserved", "<i4")]


def unpack_job(scratch, Mrows, Ncols, T, Nn, Kk, n_off, k_off, dst, sT, sN, sK, scale=1.0, accumulate=True):
    return (scratch.data_ptr(), dst.data_ptr(), sT, sN, sK, Mrows, Ncols, T, Nn, Kk, n_off, k_off, 1 if accumulate else 0, scale, 0)


def make_unpack_table(jobs, device):
    import numpy as np
    arr = np.array(jobs, dtype=UNPACK_JOB_DTYPE)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(jobs), int(max(j[7] * j[8] * j[9] for j in jobs))


def unpack_wgrad_batched(table):
    t, n, mx = table
    L.check(L.load().usseg_unpack_wgrad_batched(t.data_ptr(), n, mx, _stream()), "unpack_wgrad_batched")


# ------------------------------------------------------------------------------------------------ norm / act / pool
def _norm_desc(x, C_logical, out_ld, G, mode, eps, act, alpha) -> NormDesc:
    B, H, W, Cphys, ldx = geom(x)
    return NormDesc(B * H * W, C_logical, Cphys, ldx, out_ld, G, mode, eps, act, alpha)


def _readable(t: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    """The norm kernels read their per-channel vectors in 16-byte pieces up to Cphys floats (include/usseg.h).  The model's vectors
    are views of the flat parameter buffer, where every variable is padded to 8 floats; a stand-alone C-float tensor (tests) is
    copied into a padded one so the kernel never reads past an allocation."""
    if t is None or t.numel() >= n or t.storage_offset() + n <= t.untyped_storage().nbytes() // t.element_size():
        return t
    out = torch.zeros(n, dtype=t.dtype, device=t.device)
    out[:t.numel()] = t.reshape(-1)
    return out


def norm_act_fwd(x, C_logical, gamma, beta, out, mode, G=1, eps=1e-3, act=ACT_NONE, alpha=0.0, mean=None, var=None, mask=None):
    d = _norm_desc(x, C_logical, geom(out)[4], G, mode, eps, act, alpha)
    gamma, beta, mean, var = (_readable(t, d.Cphys) for t in (gamma, beta, mean, var))
    ldm = geom(mask)[4] if mask is not None else 0
    L.check(L.load().usseg_norm_act_fwd(C.byref(d), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(mean), _ptr(var),
                                        _ptr(mask), ldm, out.data_ptr(), _stream()), "norm_act_fwd")
    return out


def norm_act_fwd_gap(x, C_logical, gamma, beta, out, mode, G=1, eps=1e-3, act=ACT_NONE, alpha=0.0, mean=None, var=None):
    """norm + activation that also emits the partial rows of the global average pool of its output (split attention,
    ResNest.py:179): -> (out, (rows [B,nb,Cphys] fp32, nb, Cphys)) for ``splitattn_fwd(..., gap=...)``."""
    B, H, W, Cphys, _ = geom(x)
    d = _norm_desc(x, C_logical, geom(out)[4], G, mode, eps, act, alpha)
    gamma, beta, mean, var = (_readable(t, Cphys) for t in (gamma, beta, mean, var))
    nb = max(1, min(32, (H * W) // 32))
    rows = torch.empty((B, nb, Cphys), dtype=torch.float32, device=x.device)
    L.check(L.load().usseg_norm_act_fwd_gap(C.byref(d), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(mean), _ptr(var), out.data_ptr(),
                                            B, nb, rows.data_ptr(), _stream()), "norm_act_fwd_gap")
    return out, (rows, nb, Cphys)


def norm_act_bwd_sa(x, dout, C_logical, gamma, beta, dx, dgamma, dbeta, mode, G, eps, act, alpha, sa_s, sa_dg, sa_mult, mean=None, var=None,
                    dbias=None):
    """Norm backward whose incoming gradient is the split attention re-weighting's backward mult*s[b][c]*dout + dg[b][c]
    (formed in registers: no usseg_splitattn_apply_bwd_dy pass, no dy tensor)."""
    B, H, W, Cphys, ldx = geom(x)
    d = NormDesc(B * H * W, C_logical, Cphys, ldx, geom(dout)[4], G, mode, eps, act, alpha, geom(dx)[4])
    gamma, beta, mean, var = (_readable(t, Cphys) for t in (gamma, beta, mean, var))
    L.check(L.load().usseg_norm_act_bwd_sa(C.byref(d), x.data_ptr(), dout.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(mean), _ptr(var), B,
                                           sa_s.data_ptr(), sa_dg.data_ptr(), sa_mult, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                           _ptr(dbias), reduce_ws(x.device).data_ptr(), _stream()), "norm_act_bwd_sa")
    return dx


def norm_act_bwd_pair(xa, dya, Ca, gamma_a, beta_a, dxa, dgamma_a, dbeta_a, dbias_a, xb, doutb, Cb, Gb, gamma_b, beta_b, dxb, dgamma_b, dbeta_b,
                      dbias_b, sa_s, sa_dg, sa_mult, eps, alpha) -> bool:
    """A plain LayerNormalization + LeakyReLU backward (A: one group) and one with the split-attention re-weighting folded in (B: ``Gb`` groups, as
    ``norm_act_bwd_sa``) as ONE launch; False - nothing launched - when the pair has no fused instantiation (the caller makes the two calls)."""
    Ba, Ha, Wa, Cpa, ldxa = geom(xa)
    Bb, Hb, Wb, Cpb, ldxb = geom(xb)
    da = NormDesc(Ba * Ha * Wa, Ca, Cpa, ldxa, geom(dya)[4], 1, 0, eps, ACT_LRELU, alpha, geom(dxa)[4])
    db = NormDesc(Bb * Hb * Wb, Cb, Cpb, ldxb, geom(doutb)[4], Gb, 0, eps, ACT_LRELU, alpha, geom(dxb)[4])
    gamma_a, beta_a, gamma_b, beta_b = (_readable(t, n) for t, n in ((gamma_a, Cpa), (beta_a, Cpa), (gamma_b, Cpb), (beta_b, Cpb)))
    rc = L.load().usseg_norm_act_bwd_pair(C.byref(da), xa.data_ptr(), dya.data_ptr(), gamma_a.data_ptr(), beta_a.data_ptr(), dxa.data_ptr(),
                                          dgamma_a.data_ptr(), dbeta_a.data_ptr(), dbias_a.data_ptr(), C.byref(db), xb.data_ptr(), doutb.data_ptr(),
                                          gamma_b.data_ptr(), beta_b.data_ptr(), Bb, sa_s.data_ptr(), sa_dg.data_ptr(), float(sa_mult), dxb.data_ptr(),
                                          dgamma_b.data_ptr(), dbeta_b.data_ptr(), dbias_b.data_ptr(), reduce_ws(xa.device).data_ptr(), _stream())
    if rc == -2:        # USSEG_ERR_UNSUPPORTED
        return False
    L.check(rc, "norm_act_bwd_pair")
    return True


def norm_act_bwd(x, dy, C_logical, gamma, beta, dx, dgamma, dbeta, mode, G=1, eps=1e-3, act=ACT_NONE, alpha=0.0, mean=None,
                 var=None, dbias=None, mask=None):
    B, H, W, Cphys, ldx = geom(x)
    lddy = geom(dy)[4]
    d = NormDesc(B * H * W, C_logical, Cphys, ldx, lddy, G, mode, eps, act, alpha, geom(dx)[4])
    gamma, beta, mean, var = (_readable(t, Cphys) for t in (gamma, beta, mean, var))
    ldm = geom(mask)[4] if mask is not None else 0
    L.check(L.load().usseg_norm_act_bwd(C.byref(d), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(mean),
                                        _ptr(var), _ptr(mask), ldm, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dbias),
                                        reduce_ws(x.device).data_ptr(), _stream()), "norm_act_bwd")
    return dx


def accuracy(probs, y_true, acc):
    """acc[0] = mean over the pixels of argmax(probs) == argmax(y_true) (TBI_ResNest.py:48-51); ``acc``: fp32 [ACC_FLOATS], zeroed once."""
    assert acc.numel() >= ACC_FLOATS and probs.is_contiguous() and y_true.is_contiguous() and probs.dtype == y_true.dtype == torch.float32
    C_ = probs.shape[-1]
    L.check(L.load().usseg_accuracy(probs.data_ptr(), y_true.data_ptr(), probs.numel() // C_, C_, acc.data_ptr(), _stream()), "accuracy")
    return acc[0]


def norm_act_bwd_res(x, dy, C_logical, gamma, beta, dres, dx, dgamma, dbeta, eps, dbias=None):
    """Plain LayerNormalization backward + the residual branch of a pre-norm block: dx = bf16(bf16(LN'(dy)) + dres) in one pass;
    ``dbias`` += the column sums of the stored dx."""
    B, H, W, Cphys, ldx = geom(x)
    d = NormDesc(B * H * W, C_logical, Cphys, ldx, geom(dy)[4], 1, 0, eps, ACT_NONE, 0.0, geom(dx)[4])
    L.check(L.load().usseg_norm_act_bwd_res(C.byref(d), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dres.data_ptr(),
                                            geom(dres)[4], dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dbias),
                                            reduce_ws(x.device).data_ptr(), _stream()), "norm_act_bwd_res")
    return dx


def bn_act_pool_fwd(x, C_logical, gamma, beta, mean, var, eps, act, alpha, out=None):
    """inference BatchNorm + activation + AveragePooling2D(2,2) in one pass: x [B,H,W,Cp] -> [B,H/2,W/2,Cp]."""
    B, H, W, Cp, ldx = geom(x)
    out = out if out is not None else new_act(B, H // 2, W // 2, Cp, x.device)
    L.check(L.load().usseg_bn_act_pool_fwd(x.data_ptr(), B, H, W, C_logical, Cp, ldx, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                           eps, act, alpha, out.data_ptr(), geom(out)[4], _stream()), "bn_act_pool_fwd")
    return out


def bn_act_pool_bwd(x, dy, C_logical, gamma, beta, mean, var, eps, act, alpha, dx, dgamma, dbeta, dbias=None):
    B, H, W, Cp, ldx = geom(x)
    L.check(L.load().usseg_bn_act_pool_bwd(x.data_ptr(), dy.data_ptr(), B, H, W, C_logical, Cp, ldx, geom(dy)[4], gamma.data_ptr(), beta.data_ptr(),
                                           mean.data_ptr(), var.data_ptr(), eps, act, alpha, dx.data_ptr(), geom(dx)[4], dgamma.data_ptr(),
                                           dbeta.data_ptr(), _ptr(dbias), reduce_ws(x.device).data_ptr(), _stream()), "bn_act_pool_bwd")
    return dx


def dropout_mask(mask, seed: int, rate: float, step_dev=None):
    """``step_dev``: int32 device counter mixed into the seed (fresh masks when the launch is replayed from a HIP graph)."""
    B, H, W, Cc, ld = geom(mask)
    if step_dev is not None:
        L.check(L.load().usseg_dropout_mask_step(mask.data_ptr(), B * H * W, Cc, ld, seed, step_dev.data_ptr(), rate, _stream()), "dropout_mask_step")
        return mask
    L.check(L.load().usseg_dropout_mask(mask.data_ptr(), B * H * W, Cc, ld, seed, rate, _stream()), "dropout_mask")
    return mask


def channel_stats(x, C_logical, s, s2):
    B, H, W, _, ldx = geom(x)
    L.check(L.load().usseg_channel_stats(x.data_ptr(), B * H * W, C_logical, ldx, s.data_ptr(), s2.data_ptr(),
                                         reduce_ws(x.device).data_ptr(), _stream()), "channel_stats")


def bn_finalize_stats(s, s2, M, C_logical, momentum, mean, var, moving_mean=None, moving_var=None):
    L.check(L.load().usseg_bn_finalize_stats(s.data_ptr(), s2.data_ptr(), M, C_logical, momentum, mean.data_ptr(), var.data_ptr(),
                                             _ptr(moving_mean), _ptr(moving_var), _stream()), "bn_finalize_stats")


def bn_train_bwd_fix(x, dx, C_logical, gamma, mean, var, eps, tg, tb):
    B, H, W, Cphys, ldx = geom(x)
    L.check(L.load().usseg_bn_train_bwd_fix(x.data_ptr(), dx.data_ptr(), B * H * W, C_logical, Cphys, ldx, geom(dx)[4], gamma.data_ptr(),
                                            mean.data_ptr(), var.data_ptr(), eps, tg.data_ptr(), tb.data_ptr(), _stream()), "bn_train_bwd_fix")


def act_fwd(x, out, act, alpha):
    B, H, W, Cc, ldx = geom(x)
    L.check(L.load().usseg_act_fwd(x.data_ptr(), B * H * W, Cc, ldx, geom(out)[4], act, alpha, out.data_ptr(), _stream()), "act_fwd")
    return out


def act_bwd(x, dy, dx, act, alpha):
    B, H, W, Cc, ldx = geom(x)
    L.check(L.load().usseg_act_bwd(x.data_ptr(), dy.data_ptr(), B * H * W, Cc, ldx, geom(dy)[4], geom(dx)[4], act, alpha,
                                   dx.data_ptr(), _stream()), "act_bwd")
    return dx


def act_bwd_colsum(x, dy, dx, act, alpha, db, C_logical):
    """act_bwd + the column sums of its output into ``db`` (the producing conv's bias gradient) in one pass."""
    B, H, W, Cc, ldx = geom(x)
    L.check(L.load().usseg_act_bwd_colsum(x.data_ptr(), dy.data_ptr(), B * H * W, C_logical, ldx, geom(dy)[4], geom(dx)[4], act, alpha, dx.data_ptr(),
                                          db.data_ptr(), reduce_ws(x.device).data_ptr(), _stream()), "act_bwd_colsum")
    return dx


def avgpool2_fwd(x, out):
    B, H, W, Cc, ldx = geom(x)
    L.check(L.load().usseg_avgpool2_fwd(x.data_ptr(), B, H, W, Cc, ldx, geom(out)[4], out.data_ptr(), _stream()), "avgpool2_fwd")
    return out


def avgpool2_bwd(dy, dx, add=None, db=None):
    """``db``: optional fp32 vector that receives the column sums of dx (the bias gradient of the conv in front of the pool)."""
    B, H, W, Cc, lddx = geom(dx)
    ldadd = geom(add)[4] if add is not None else 0
    if db is not None and Cc <= 512:
        L.check(L.load().usseg_avgpool2_bwd_colsum(dy.data_ptr(), B, H, W, Cc, geom(dy)[4], lddx, _ptr(add), ldadd, dx.data_ptr(), db.data_ptr(),
                                                   reduce_ws(dx.device).data_ptr(), _stream()), "avgpool2_bwd_colsum")
        return dx
    L.check(L.load().usseg_avgpool2_bwd(dy.data_ptr(), B, H, W, Cc, geom(dy)[4], lddx, _ptr(add), ldadd, dx.data_ptr(),
                                        _stream()), "avgpool2_bwd")
    return dx


def copy_channels(src, dst, accumulate=False):
    B, H, W, Cc, lds = geom(src)
    Bd, Hd, Wd, Cd, ldd = geom(dst)
    assert B * H * W == Bd * Hd * Wd and Cc == Cd
    L.check(L.load().usseg_copy_channels(src.data_ptr(), B * H * W, Cc, lds, dst.data_ptr(), ldd, 1 if accumulate else 0,
                                         _stream()), "copy_channels")
    return dst


def cast_input(x: torch.Tensor, Cphys: int) -> torch.Tensor:
    """fp32/fp64 NHWC device tensor -> bf16 [B,H,W,Cphys] (zero pad channels)."""
    assert x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, torch.float64)
    B, H, W, Cc = x.shape
    out = new_act(B, H, W, Cphys, x.device)
    L.check(L.load().usseg_cast_input(x.data_ptr(), 1 if x.dtype == torch.float64 else 0, B * H * W, Cc, out.data_ptr(), Cphys,
                                      _stream()), "cast_input")
    return out


def to_f32(x: torch.Tensor, C_logical: Optional[int] = None) -> torch.Tensor:
    B, H, W, Cc, ld = geom(x)
    Cl = C_logical or Cc
    out = torch.empty((B, H, W, Cl), dtype=torch.float32, device=x.device)
    L.check(L.load().usseg_cast_bf16_to_f32(x.data_ptr(), B * H * W, Cl, ld, out.data_ptr(), _stream()), "cast_bf16_to_f32")
    return out


def colsum(dy, db: torch.Tensor, C_logical: Optional[int] = None):
    B, H, W, Cc, ld = geom(dy)
    L.check(L.load().usseg_colsum(dy.data_ptr(), B * H * W, C_logical or Cc, ld, db.data_ptr(), reduce_ws(dy.device).data_ptr(),
                                  _stream()), "colsum")


# ------------------------------------------------------------------------------------------------ optimiser helpers
def fill_f32(t: torch.Tensor, value: float = 0.0):
    assert t.dtype == torch.float32 and t.is_contiguous()
    L.check(L.load().usseg_fill_f32(t.data_ptr(), t.numel(), value, _stream()), "fill_f32")


def sumsq(g: torch.Tensor, out: torch.Tensor):
    assert out.numel() >= ACC_FLOATS, "sumsq accumulates through a USSEG_ACC_FLOATS buffer"
    L.check(L.load().usseg_sumsq(g.data_ptr(), g.numel(), out.data_ptr(), _stream()), "sumsq")


def sum_f32(x: torch.Tensor, out: torch.Tensor):
    """out[0] = sum(x) (fp32, contiguous) through an ACC_FLOATS accumulator (zeroed once): fixed-order, no framework reduction."""
    assert out.numel() >= ACC_FLOATS and x.dtype == torch.float32 and x.is_contiguous()
    L.check(L.load().usseg_sum_f32(x.data_ptr(), x.numel(), out.data_ptr(), _stream()), "sum_f32")
    return out[0]


def sumsq_advance(g: torch.Tensor, out: torch.Tensor, step_dev, lr_t_dev, lr, b1, b2):
    """sumsq + adam_advance in ONE launch (the last workgroup of the ordered sum moves the step counter on)."""
    assert out.numel() >= ACC_FLOATS
    L.check(L.load().usseg_sumsq_advance(g.data_ptr(), g.numel(), out.data_ptr(), step_dev.data_ptr(), lr_t_dev.data_ptr(), lr, b1, b2, _stream()),
            "sumsq_advance")


def reinject_hidden(hidden: torch.Tensor, views, backward: bool):
    """Decoder.py:140-141 for all scales at once.  ``hidden``: contiguous bf16 tensor; ``views``: channel-slice views
    [B,h_i,w_i,c0_i] of the concat buffers (forward: written; backward: their gradients, summed into ``hidden``)."""
    assert hidden.is_contiguous() and hidden.dtype == BF16 and 1 <= len(views) <= 4
    n = len(views)
    bufs = (C.c_void_p * n)(*[v.data_ptr() for v in views])
    c0 = (C.c_int32 * n)(*[v.shape[3] for v in views])
    ld = (C.c_int32 * n)(*[geom(v)[4] for v in views])
    for v in views:
        assert v.numel() == hidden.numel()
    L.check(L.load().usseg_reinject_hidden(hidden.data_ptr(), hidden.numel(), n, bufs, c0, ld, 1 if backward else 0, _stream()), "reinject_hidden")


def scale_by_clip(g: torch.Tensor, sumsq_t: torch.Tensor, clip_norm: float):
    L.check(L.load().usseg_scale_f32(g.data_ptr(), g.numel(), sumsq_t.data_ptr(), clip_norm, _stream()), "scale_f32")


def adam_clip_step(p, g, m, v, sumsq_t, clip_norm, lr_t_dev, b1, b2, eps):
    L.check(L.load().usseg_adam_clip_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), _ptr(sumsq_t),
                                          clip_norm, lr_t_dev.data_ptr(), b1, b2, eps, _stream()), "adam_clip_step")


def adam_advance(step_dev, lr_t_dev, lr, b1, b2):
    L.check(L.load().usseg_adam_advance(step_dev.data_ptr(), lr_t_dev.data_ptr(), lr, b1, b2, _stream()), "adam_advance")


# ------------------------------------------------------------------------------------------------ fused stem
def stem_fwd(x, w1, b1, w2, b2, w3, b3, gamma, beta, mean, var, eps, alpha):
    """ResNest.py:39-47 in one launch -> (y1 [B,H,W,16], t1 [B,H,W,32], c2 [B,H,W,32] pre-norm, pooled [B,H/2,W/2,32])."""
    B, H, W, Cin, ldx = geom(x)
    assert Cin == 8 and x.dtype == BF16
    dev = x.device
    y1, t1, c2 = new_act(B, H, W, 16, dev), new_act(B, H, W, 32, dev), new_act(B, H, W, 32, dev)
    pooled = new_act(B, H // 2, W // 2, 32, dev)
    L.check(L.load().usseg_stem_fwd(B, H, W, x.data_ptr(), ldx, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(),
                                    b3.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, alpha,
                                    y1.data_ptr(), t1.data_ptr(), c2.data_ptr(), pooled.data_ptr(), _stream()), "stem_fwd")
    return y1, t1, c2, pooled


def conv3_dgrad_actbwd(dy, wd, yact, dx, mode, alpha, dbias, gamma=None, beta=None, var=None, eps=1e-3, dgamma=None, dbeta=None) -> bool:
    """3x3 backward-data of a 32-output-channel stem conv + the backward of the layer in front of it in one launch: mode 2 = folded inference
    BatchNorm + LeakyReLU from the activated tensor ``yact`` (32 channels), mode 0 = LeakyReLU + column sums (16 channels).  False - nothing
    launched - when there is no fused kernel for the shape."""
    B, H, W, Cdy, lddy = geom(dy)
    Co = yact.shape[-1]
    if Cdy != 32 or tuple(wd.shape) != (roundup(Co, 16), 9 * 32):
        return False
    rc = L.load().usseg_conv3_dgrad_actbwd(B, H, W, dy.data_ptr(), lddy, wd.data_ptr(), Co, yact.data_ptr(), geom(yact)[4], mode, _ptr(gamma), _ptr(beta),
                                           _ptr(var), eps, alpha, dx.data_ptr(), geom(dx)[4], _ptr(dgamma), _ptr(dbeta), dbias.data_ptr(),
                                           reduce_ws(dy.device).data_ptr(), _stream())
    if rc == -2:
        return False
    L.check(rc, "conv3_dgrad_actbwd")
    return True


# ------------------------------------------------------------------------------------------------ fused cardinal group (K3)
def cardinal_desc(B, H, W, Cin, P, cv11, cvkk, Up, Vp, Oc, ldx, ldu, ldv, ldsc, eps, alpha) -> CardinalDesc:
    return CardinalDesc(B, H, W, Cin, P, cv11, cvkk, Up, Vp, Oc, ldx, ldu, ldv, ldsc, eps, alpha)


def cardinal_supported(Cin, P, cv11, cvkk, Up, Vp, Oc) -> bool:
    d = cardinal_desc(1, 8, 8, Cin, P, cv11, cvkk, Up, Vp, Oc, Cin, Up, Vp, Oc, 1e-3, 0.3)
    return bool(L.load().usseg_cardinal_supported(C.byref(d)))


def cardinal_fwd(x, w1, b1, g1, be1, w2, b2, g2, be2, wsc, bsc, gsc, besc, P, cv11, cvkk, Up, Vp, Oc, eps, alpha):
    """One launch for a residual_S stage's first half (ResNest.py:136-147 for all paths, :99-101): grouped 1x1 -> LN -> LeakyReLU ->
    grouped 3x3 -> LN -> LeakyReLU (+ pooled partial rows) and the shortcut 1x1 -> LN -> LeakyReLU.
    -> (u_raw, u, v_raw, y, gap, sc_raw, sc); gap = (rows [B,tiles,Vp], tiles, Vp) for ``splitattn_fwd(..., gap=gap)``."""
    B, H, W, Cin, ldx = geom(x)
    dev = x.device
    u_raw, u = new_act(B, H, W, Up, dev), new_act(B, H, W, Up, dev)
    v_raw, y = new_act(B, H, W, Vp, dev), new_act(B, H, W, Vp, dev)
    sc_raw, sc = new_act(B, H, W, Oc, dev), new_act(B, H, W, Oc, dev)
    tiles = ((H + 7) // 8) * ((W + 7) // 8)
    rows = torch.empty((B, tiles, Vp), dtype=torch.float32, device=dev)
    d = cardinal_desc(B, H, W, Cin, P, cv11, cvkk, Up, Vp, Oc, ldx, Up, Vp, Oc, eps, alpha)
    g1, be1, g2, be2, gsc, besc = (_readable(t, n) for t, n in ((g1, P * cv11), (be1, P * cv11), (g2, P * cvkk), (be2, P * cvkk), (gsc, Oc), (besc, Oc)))
    L.check(L.load().usseg_cardinal_fwd(C.byref(d), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), g1.data_ptr(), be1.data_ptr(), w2.data_ptr(),
                                        b2.data_ptr(), g2.data_ptr(), be2.data_ptr(), wsc.data_ptr(), bsc.data_ptr(), gsc.data_ptr(), besc.data_ptr(),
                                        u_raw.data_ptr(), u.data_ptr(), v_raw.data_ptr(), y.data_ptr(), rows.data_ptr(), sc_raw.data_ptr(),
                                        sc.data_ptr(), _stream()), "cardinal_fwd")
    return u_raw, u, v_raw, y, (rows, tiles, Vp), sc_raw, sc


def cardinal_bwd(dout, dsc, v_raw, u_raw, sc_raw, w2d, g2, be2, g1, be1, gsc, besc, sa_s, sa_dg, sa_mult, dv, dcat, grads, Cin, P, cv11, cvkk, Up,
                 Vp, Oc, eps, alpha):
    """One launch for the backward pass of ``cardinal_fwd``'s chain (GradientTape through ResNest.py:136-147 and :100-101): the split-attention
    re-weighting's backward (``sa_mult*sa_s*dout + sa_dg``) + conv2_bn backward -> ``dv``; grouped 3x3 backward-data; conv1_bn backward ->
    ``dcat[..., :Up]``; shortcut norm backward of ``dsc`` -> ``dcat[..., Up:]``.  ``grads`` = (dg2, dbe2, db2, dg1, dbe1, db1, dgsc, dbesc,
    dbsc) accumulate.  -> (dv, dcat)"""
    B, H, W, _, ldo = geom(dout)
    d = cardinal_desc(B, H, W, Cin, P, cv11, cvkk, Up, Vp, Oc, Cin, geom(u_raw)[4], geom(v_raw)[4], geom(sc_raw)[4], eps, alpha)
    assert geom(dv)[4] == d.ldv and dcat.shape[-1] >= Up + Oc
    g2, be2, g1, be1, gsc, besc = (_readable(t, n) for t, n in ((g2, P * cvkk), (be2, P * cvkk), (g1, P * cv11), (be1, P * cv11), (gsc, Oc), (besc, Oc)))
    L.check(L.load().usseg_cardinal_bwd(C.byref(d), dout.data_ptr(), ldo, dsc.data_ptr(), geom(dsc)[4], v_raw.data_ptr(), u_raw.data_ptr(),
                                        sc_raw.data_ptr(), w2d.data_ptr(), g2.data_ptr(), be2.data_ptr(), g1.data_ptr(), be1.data_ptr(),
                                        gsc.data_ptr(), besc.data_ptr(), sa_s.data_ptr(), sa_dg.data_ptr(), float(sa_mult), dv.data_ptr(),
                                        dcat.data_ptr(), geom(dcat)[4], *[t.data_ptr() for t in grads], reduce_ws(dout.device).data_ptr(),
                                        _stream()), "cardinal_bwd")
    return dv, dcat


# ------------------------------------------------------------------------------------------------ split attention
def splitattn_desc(B, HW, P, R, Cg, Hd, ldy, ldo, Cy_phys, Co_phys, mult, norm_mode, eps, act, alpha, use_sigmoid) -> SplitAttnDesc:
    return SplitAttnDesc(B, HW, P, R, Cg, Hd, ldy, ldo, Cy_phys, Co_phys, mult, norm_mode, eps, act, alpha, 1 if use_sigmoid else 0)


def _sa_params(w1, b1, gamma, beta, mean, var, w2, b2) -> SplitAttnParams:
    return SplitAttnParams(_ptr(w1), _ptr(b1), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), _ptr(w2), _ptr(b2))


def splitattn_fwd(d: SplitAttnDesc, y, params, out, gap=None):
    """params = (w1,b1,gamma,beta,mean,var,w2,b2) fp32 tensors laid out per path.  ``gap``: the pooled partial rows
    (tensor [B,nb,Cphys], nb, Cphys) that ``norm_act_fwd_gap`` produced together with y - without it the pooling pass over y
    runs here.  Returns (out, g, s, ws); g = (tensor, rows, stride) is what ``splitattn_bwd`` takes back."""
    dev = y.device
    Cy = d.P * d.R * d.Cg
    s = torch.empty((d.B, Cy), dtype=torch.float32, device=dev)
    ws = torch.empty(d.B * d.P * (d.Cg + 2 * d.Hd), dtype=torch.float32, device=dev)
    lib = L.load()
    if gap is None:
        g = torch.empty((d.B, Cy), dtype=torch.float32, device=dev)      # overwritten by usseg_splitattn_gap
        L.check(lib.usseg_splitattn_gap(C.byref(d), y.data_ptr(), g.data_ptr(), reduce_ws(dev).data_ptr(), _stream()), "splitattn_gap")
        gap = (g, 1, Cy)
    sp = _sa_params(*params)
    L.check(lib.usseg_splitattn_mlp_fwd(C.byref(d), gap[0].data_ptr(), gap[1], gap[2], C.byref(sp), s.data_ptr(), ws.data_ptr(), _stream()),
            "splitattn_mlp_fwd")
    L.check(lib.usseg_splitattn_apply_fwd(C.byref(d), y.data_ptr(), s.data_ptr(), out.data_ptr(), _stream()), "splitattn_apply_fwd")
    return out, gap, s, ws


def splitattn_bwd(d: SplitAttnDesc, y, dout, params, grads, g, s, ws, dy):
    """grads = (dw1,db1,dgamma,dbeta,dw2,db2) accumulate (per-image partial rows + an ordered finishing reduction).
    ``dy`` tensor: written (dy = mult*s*dout + dg).  ``dy=None``: the re-weighting's backward is left to the caller's
    ``norm_act_bwd_sa`` (R == 1 only) and (s, dg) are returned for it."""
    dev = y.device
    Cy = d.P * d.R * d.Cg
    dg = torch.empty((d.B, Cy), dtype=torch.float32, device=dev)
    lib = L.load()
    lddo = geom(dout)[4]
    if not isinstance(g, tuple):
        g = (g, 1, Cy)
    sp = _sa_params(*params)
    sg = SplitAttnGrads(*[_ptr(t) for t in grads])
    gws = torch.empty(int(lib.usseg_splitattn_mlp_bwd_ws_floats(C.byref(d))), dtype=torch.float32, device=dev)
    # sum_pixels dout*y per (image, channel) as per-workgroup rows, summed by the MLP kernel itself (no finishing launch between the two)
    L.check(lib.usseg_splitattn_bwd_fused(C.byref(d), y.data_ptr(), dout.data_ptr(), lddo, g[0].data_ptr(), g[1], g[2], C.byref(sp), s.data_ptr(),
                                          ws.data_ptr(), dg.data_ptr(), C.byref(sg), reduce_ws(dev).data_ptr(), gws.data_ptr(), _stream()),
            "splitattn_bwd_fused")
    if dy is None:
        assert d.R == 1, "the fused re-weighting backward needs identical radix branches (R == 1)"
        return s, dg
    L.check(lib.usseg_splitattn_apply_bwd_dy(C.byref(d), dout.data_ptr(), lddo, s.data_ptr(), dg.data_ptr(), dy.data_ptr(), geom(dy)[4],
                                             _stream()), "splitattn_apply_bwd_dy")
    return dy


# ------------------------------------------------------------------------------------------------ head softmax + loss
def softmax_loss(logits, y_true, probs, loss, dlogits, *, HW, C_classes, loss_kind=0, label_smoothing=0.1, clip_eps=1e-7,
                 inv_global_batch=1.0, scale=None, quad_w=0):
    """logits fp32 [..., ldl]; y_true fp32 [..., C] or None; probs fp32 [..., C]; dlogits bf16 [..., 8] or None.
    ``quad_w`` = full-resolution width when logits / dlogits are in the head's space-to-depth layout [B,H/2,W/2,16]."""
    M = probs.numel() // C_classes
    ldl = logits.shape[-1]
    assert loss is None or loss_kind == 1 or loss.numel() >= ACC_FLOATS, "the scalar loss accumulates through a USSEG_ACC_FLOATS buffer"
    d = LossDesc(M, HW, C_classes, ldl, 16 if quad_w else 8, loss_kind, label_smoothing, clip_eps, inv_global_batch, quad_w)
    L.check(L.load().usseg_softmax_loss_fwd_bwd(C.byref(d), logits.data_ptr(), _ptr(y_true), _ptr(scale), probs.data_ptr(), _ptr(loss),
                                                _ptr(dlogits), _stream()), "softmax_loss")


def head_quad_softmax_loss(x, wq, bias, C_classes, y_true, probs, loss, dlogits, *, label_smoothing=0.1, clip_eps=1e-7, inv_global_batch=1.0) -> bool:
    """The quad-form head conv + softmax + loss (loss_kind 0) in one launch; x [B,h,w,Cin_phys] bf16, wq [16][9*Cin_phys], probs [B,2h,2w,C] fp32,
    dlogits bf16 [B,h,w,16] or None.  False - nothing launched - when there is no fused kernel for the size (ordered-sum slots, LDS)."""
    B, h, w, Cx, ldx = geom(x)
    assert tuple(wq.shape) == (16, 9 * Cx) and (loss is None or loss.numel() >= ACC_FLOATS)
    rc = L.load().usseg_head_quad_softmax_loss(x.data_ptr(), B, h, w, Cx, ldx, wq.data_ptr(), bias.data_ptr(), C_classes, _ptr(y_true), probs.data_ptr(),
                                               _ptr(loss), _ptr(dlogits), label_smoothing, clip_eps, inv_global_batch, _stream())
    if rc == -2:
        return False
    L.check(rc, "head_quad_softmax_loss")
    return True


def loss_from_probs(probs, y_true, loss, *, HW, C_classes, loss_kind=0, label_smoothing=0.1, clip_eps=1e-7, inv_global_batch=1.0, scale=None):
    """The reference's public loss methods on PROBABILITIES (compute_loss / my_loss_cat); ``loss`` (pre-zeroed) accumulates."""
    assert probs.dtype == torch.float32 and y_true.dtype == torch.float32 and probs.is_contiguous() and y_true.is_contiguous()
    M = probs.numel() // C_classes
    assert loss_kind == 1 or loss.numel() >= ACC_FLOATS
    d = LossDesc(M, HW, C_classes, C_classes, 8, loss_kind, label_smoothing, clip_eps, inv_global_batch, 0)
    L.check(L.load().usseg_loss_from_probs(C.byref(d), probs.data_ptr(), y_true.data_ptr(), _ptr(scale), loss.data_ptr(), _stream()),
            "loss_from_probs")


def quad_bias_expand(bias, C_logical, out):
    L.check(L.load().usseg_quad_bias_expand(bias.data_ptr(), C_logical, out.numel() // 4, out.data_ptr(), _stream()), "quad_bias_expand")


def space_to_depth2(full, quad, Np, to_quad=True):
    """full [B,2H,2W,C] (channel-slice views allowed) <-> quad [B,H,W,4*Np]: quad[b,i,j,(2a+b')*Np+n] = full[b,2i+a,2j+b',n]."""
    B, H2, W2, Cc, ldf = geom(full)
    _, H, W, _, ldq = geom(quad)
    assert (H2, W2) == (2 * H, 2 * W)
    L.check(L.load().usseg_space_to_depth2(full.data_ptr(), B, H, W, Cc, ldf, quad.data_ptr(), Np, ldq, 1 if to_quad else 0, _stream()),
            "space_to_depth2")


def tconv_quad_fwd(x, wq_f, bias_q, ksize, Np, y4):
    B, H, W, Cin, ldx = geom(x)
    d = _conv_desc(B, H, W, Cin, 4 * Np, ldx, geom(y4)[4], 3, 1)
    L.check(L.load().usseg_tconv_quad_fwd(C.byref(d), ksize, Np, x.data_ptr(), wq_f.data_ptr(), _ptr(bias_q), y4.data_ptr(), _stream()), "tconv_quad_fwd")
    return y4


def tconv_quad_dgrad(dy4, wq_d, ksize, Np, dx):
    B, H, W, _, ldy = geom(dy4)
    _, _, _, Cin, ldx = geom(dx)
    d = _conv_desc(B, H, W, Cin, 4 * Np, ldx, ldy, 3, 1)
    L.check(L.load().usseg_tconv_quad_dgrad(C.byref(d), ksize, Np, dy4.data_ptr(), wq_d.data_ptr(), dx.data_ptr(), _stream()), "tconv_quad_dgrad")
    return dx


def tconv_quad_wgrad(x, dy4, ksize, Np, dq):
    B, H, W, Cin, ldx = geom(x)
    d = _conv_desc(B, H, W, Cin, 4 * Np, ldx, geom(dy4)[4], 3, 1)
    ws = wgrad_ws(x.device)
    L.check(L.load().usseg_tconv_quad_wgrad(C.byref(d), ksize, Np, x.data_ptr(), dy4.data_ptr(), dq.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _stream()), "tconv_quad_wgrad")


def quad_bias_fold(d16, C_classes, dbias):
    L.check(L.load().usseg_quad_bias_fold(d16.data_ptr(), C_classes, dbias.data_ptr(), _stream()), "quad_bias_fold")


def tconv_quad_unpack(dq, Cin_phys, Cin, Cout, Np, ksize, grad):
    L.check(L.load().usseg_tconv_quad_unpack(dq.data_ptr(), Cin_phys, Cin, Cout, Np, ksize, grad.data_ptr(), _stream()), "tconv_quad_unpack")


def quad_head_fold(dq, Cin_phys, Cin, Cout, Np, ksize, grad, d16, dbias):
    """tconv_quad_unpack + quad_bias_fold in one launch (the tail of the quad-form head's backward)."""
    L.check(L.load().usseg_quad_head_fold(dq.data_ptr(), Cin_phys, Cin, Cout, Np, ksize, grad.data_ptr(), d16.data_ptr(), dbias.data_ptr(), _stream()),
            "quad_head_fold")


def loss_cat_scale(y_true, scale):
    B, H, W, Cc = y_true.shape
    L.check(L.load().usseg_loss_cat_scale(y_true.data_ptr(), B, H * W, Cc, scale.data_ptr(), _stream()), "loss_cat_scale")


# ------------------------------------------------------------------------------------------------ ViT attention helpers
def gemm_nt_batched(x, w, y, M, N, K, ldx, ldw, ldy, nb1, nb2, xs, ws, ys, out_f32=False):
    """Y[b1,b2][m][n] = sum_k X[b1,b2][m][k] W[b1,b2][n][k]; x/w/y are tensors whose data_ptr is the (0,0) batch base."""
    d = GemmDesc(M, N, K, ldx, ldw, ldy, nb1, nb2, xs[0], xs[1], ws[0], ws[1], ys[0], ys[1], OUT_F32 if out_f32 else 0)
    L.check(L.load().usseg_gemm_nt_batched(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), _stream()), "gemm_nt_batched")


def gemm_tn_batched(a, b, out, M, N, K, lda, ldb, nb1, nb2, as_, bs, os):
    """OUT[b1,b2][m][n] += sum_r A[b1,b2][r][m] B[b1,b2][r][n] (fp32 out, zero it first)."""
    d = GemmDesc(M, N, K, lda, ldb, N, nb1, nb2, as_[0], as_[1], bs[0], bs[1], os[0], os[1], 0)
    L.check(L.load().usseg_gemm_tn_batched(C.byref(d), a.data_ptr(), b.data_ptr(), out.data_ptr(), _stream()), "gemm_tn_batched")


def softmax_rows_fwd(s, n, scale, p32, pbf):
    L.check(L.load().usseg_softmax_rows_fwd(s.data_ptr(), s.numel() // n, n, scale, p32.data_ptr(), pbf.data_ptr(), _stream()), "softmax_rows_fwd")


def softmax_rows_bwd(p32, dp, n, scale, ds):
    L.check(L.load().usseg_softmax_rows_bwd(p32.data_ptr(), dp.data_ptr(), p32.numel() // n, n, scale, ds.data_ptr(), _stream()), "softmax_rows_bwd")


def flash_attn_fwd(qkv, nh, scale, out, lse, out32=None):
    """qkv [B,N,1,3*hidden] bf16 -> out [B,N,1,hidden] = softmax(scale q k^T) v per (image, head); lse fp32 [B*nh, N] (base 2);
    out32: optional dense fp32 copy of out (for the backward's delta)."""
    B, N, _, C3 = qkv.shape
    hs = C3 // 3
    d = L.FlashDesc(B, N, nh, hs // nh, geom(qkv)[4], geom(out)[4], scale)
    e = qkv.element_size()
    L.check(L.load().usseg_flash_attn_fwd(C.byref(d), qkv.data_ptr(), qkv.data_ptr() + hs * e, qkv.data_ptr() + 2 * hs * e, out.data_ptr(),
                                          _ptr(out32), lse.data_ptr(), _stream()), "flash_attn_fwd")
    return out


def flash_attn_bwd(qkv, nh, scale, out, d_out, lse, delta, dqkv, out32=None):
    """dqkv [B,N,1,3*hidden] bf16 (written): gradients of q, k, v slices; recomputes the probabilities from lse."""
    B, N, _, C3 = qkv.shape
    hs = C3 // 3
    assert geom(dqkv)[4] == geom(qkv)[4] and geom(d_out)[4] == geom(out)[4]
    d = L.FlashDesc(B, N, nh, hs // nh, geom(qkv)[4], geom(out)[4], scale)
    e = qkv.element_size()
    L.check(L.load().usseg_flash_attn_bwd(C.byref(d), qkv.data_ptr(), qkv.data_ptr() + hs * e, qkv.data_ptr() + 2 * hs * e, out.data_ptr(),
                                          _ptr(out32), d_out.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), dqkv.data_ptr() + hs * e,
                                          dqkv.data_ptr() + 2 * hs * e, _stream()), "flash_attn_bwd")
    return dqkv


def transpose_batched(src, R, Cc, lds, nb1, nb2, ss, dst):
    L.check(L.load().usseg_transpose_batched(src.data_ptr(), R, Cc, lds, nb1, nb2, ss[0], ss[1], dst.data_ptr(), _stream()), "transpose_batched")


def cast_f32_to_bf16_batched(src, R, Cc, nb1, nb2, dst, ldd, ds):
    L.check(L.load().usseg_cast_f32_to_bf16_batched(src.data_ptr(), R, Cc, nb1, nb2, dst.data_ptr(), ldd, ds[0], ds[1], _stream()),
            "cast_f32_to_bf16_batched")


# ------------------------------------------------------------------------------------------------ windowed-attention encoder (Swin)
def patchify(x: torch.Tensor, patch: int) -> torch.Tensor:
    """fp32/fp64 [B,H,W,C] -> bf16 [B,H/p,W/p,roundup(p*p*C,8)]: the space-to-depth half of Conv2D(kernel = stride = p)."""
    assert x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, torch.float64)
    B, H, W, Cc = x.shape
    Cp = roundup(patch * patch * Cc, 8)
    out = new_act(B, H // patch, W // patch, Cp, x.device)
    L.check(L.load().usseg_patchify(x.data_ptr(), 1 if x.dtype == torch.float64 else 0, B, H, W, Cc, patch, out.data_ptr(), Cp, _stream()), "patchify")
    return out


def patch_merge(full, merged, backward=False):
    """full [B,H,W,C] <-> merged [B,H/2,W/2,4C] in the reference's order x0=(0,0), x1=(1,0), x2=(0,1), x3=(1,1)."""
    B, H, W, Cc, ldf = geom(full)
    L.check(L.load().usseg_patch_merge(full.data_ptr(), B, H, W, Cc, ldf, merged.data_ptr(), geom(merged)[4], 1 if backward else 0, _stream()), "patch_merge")


def ln_wide_fwd(x, gamma, beta, eps, out):
    B, H, W, Cc, ldx = geom(x)
    L.check(L.load().usseg_ln_wide_fwd(x.data_ptr(), B * H * W, Cc, ldx, gamma.data_ptr(), beta.data_ptr(), eps, out.data_ptr(), geom(out)[4], _stream()),
            "ln_wide_fwd")
    return out


def ln_wide_bwd(x, dy, gamma, eps, dx, dgamma, dbeta):
    B, H, W, Cc, ldx = geom(x)
    ws = reduce_ws(x.device)
    L.check(L.load().usseg_ln_wide_bwd(x.data_ptr(), dy.data_ptr(), B * H * W, Cc, ldx, geom(dy)[4], gamma.data_ptr(), eps, dx.data_ptr(), geom(dx)[4],
                                       dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "ln_wide_bwd")
    return dx


def window_attn_fwd(qkv, table, heads, ws, shift, out):
    B, H, W, C3, ldq = geom(qkv)
    L.check(L.load().usseg_window_attn_fwd(qkv.data_ptr(), ldq, table.data_ptr(), B, H, W, C3 // 3, heads, ws, shift, out.data_ptr(), geom(out)[4], _stream()),
            "window_attn_fwd")
    return out


def window_attn_bwd(qkv, dout, table, heads, ws, shift, dqkv, dtable):
    B, H, W, C3, ldq = geom(qkv)
    lib = L.load()
    rows = torch.empty(int(lib.usseg_window_attn_bwd_ws_floats(B, H, W, heads, ws)), dtype=torch.float32, device=qkv.device)
    L.check(lib.usseg_window_attn_bwd(qkv.data_ptr(), ldq, dout.data_ptr(), geom(dout)[4], table.data_ptr(), B, H, W, C3 // 3, heads, ws, shift,
                                      dqkv.data_ptr(), dtable.data_ptr(), rows.data_ptr(), _stream()), "window_attn_bwd")
    return dqkv


def token_mean_fwd(x):
    B, H, W, Cc, ld = geom(x)
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    L.check(L.load().usseg_token_mean_fwd(x.data_ptr(), B, H * W, Cc, ld, out.data_ptr(), _stream()), "token_mean_fwd")
    return out


def token_mean_bwd(dy, like):
    B, H, W, Cc, ld = geom(like)
    dx = torch.empty_like(like)
    L.check(L.load().usseg_token_mean_bwd(dy.data_ptr(), B, H * W, Cc, ld, dx.data_ptr(), _stream()), "token_mean_bwd")
    return dx
