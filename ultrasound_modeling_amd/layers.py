"""Keras-shaped layers (NHWC, Keras weight layouts and names) on top of the gfx950 kernels.

These mirror the tf.keras.layers the reference instantiates (Conv2D, Conv2DTranspose, LayerNormalization,
BatchNormalization, LeakyReLU/ELU/ReLU, AveragePooling2D) with explicit ``forward`` / ``backward`` methods: the
reference's GradientTape (VisionTransformer.py:237-243) is replaced by each layer saving what its backward needs
and the model calling ``backward`` in reverse order; parameter gradients are accumulated by the kernels directly
into the flat gradient buffer (flat.py).

Activations are bf16 ``[B,H,W,Cphys]`` with Cphys = roundup(C, 8) and zero pad channels.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_ELU, ACT_LRELU, ACT_NONE, ACT_RELU, BF16, roundup

_FUSE_LN_RES = os.environ.get("USSEG_FUSE_LN_RES", "1") != "0"    # LayerNorm backward + residual add in one pass (transformer blocks)

KERAS_LRELU_ALPHA = 0.3   # tf.keras.layers.LeakyReLU() default
KERAS_ELU_ALPHA = 1.0
KERAS_BN_EPS = 1e-3
KERAS_LN_EPS = 1e-3


# ------------------------------------------------------------------------------------------------ workspace
class _Workspace:
    """Grow-only fp32 scratch shared by all layers of a device (wgrad scratch)."""
    _bufs = {}

    @classmethod
    def get(cls, device, nfloats: int) -> torch.Tensor:
        key = ops._ws_key(device)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nfloats:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("workspace grew during graph capture: run one eager step first")
            buf = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=device)
            cls._bufs[key] = buf
        return buf[:nfloats]


# ------------------------------------------------------------------------------------------------ initialisers
def he_normal_(t: torch.Tensor, fan_in: int, gen: Optional[torch.Generator] = None):
    """tf.keras.initializers.HeNormal: truncated normal (+-2 sigma), variance 2/fan_in."""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
    return t.mul_(std)


def glorot_uniform_(t: torch.Tensor, fan_in: int, fan_out: int, gen: Optional[torch.Generator] = None):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-lim, lim, generator=gen)


# ------------------------------------------------------------------------------------------------ activations (markers)
class LeakyReLU(nn.Module):
    """tf.keras.layers.LeakyReLU(): fused into the producing kernel by the parent; standalone call is supported."""
    act, alpha = ACT_LRELU, KERAS_LRELU_ALPHA

    def forward(self, x):
        self._x = x
        return ops.act_fwd(x, torch.empty_like(x), self.act, self.alpha)

    def backward(self, dy):
        return ops.act_bwd(self._x, dy, torch.empty_like(dy), self.act, self.alpha)


class ELU(LeakyReLU):
    act, alpha = ACT_ELU, KERAS_ELU_ALPHA


class ReLU(LeakyReLU):
    act, alpha = ACT_RELU, 0.0


class AveragePooling2D(nn.Module):
    """tf.keras.layers.AveragePooling2D(pool_size=2, strides=2)."""

    def forward(self, x, out=None):
        B, H, W, C, _ = ops.geom(x)
        out = out if out is not None else ops.new_act(B, H // 2, W // 2, C, x.device)
        self._shape = (B, H, W, C)
        return ops.avgpool2_fwd(x, out)

    def backward(self, dy, add=None, dx=None, db=None):
        """``db``: optional bias-gradient vector of the conv whose output this pool consumed (column sums of dx, same pass)."""
        B, H, W, C = self._shape
        dx = dx if dx is not None else ops.new_act(B, H, W, C, dy.device)
        return ops.avgpool2_bwd(dy, dx, add, db)


# ------------------------------------------------------------------------------------------------ convolutions
class Conv2D(nn.Module):
    """tf.keras.layers.Conv2D(filters, k, strides=1, padding='SAME', dilation_rate=d), kernel [k,k,Cin,Cout]."""
    transposed = False

    def __init__(self, in_channels: int, filters: int, kernel_size: int, dilation_rate: int = 1, init: str = "he"):
        super().__init__()
        k = kernel_size
        self.cin, self.cout, self.k, self.dil = in_channels, filters, k, dilation_rate
        self.cin_p, self.cout_p = roundup(in_channels, 8), roundup(filters, 8)
        self.kernel = nn.Parameter(torch.empty(self._kshape()))
        self.bias = nn.Parameter(torch.zeros(filters))
        fan_in, fan_out = self._fans()
        if init == "he":
            he_normal_(self.kernel.data, fan_in)
        else:
            glorot_uniform_(self.kernel.data, fan_in, fan_out)
        self.wp_f = self.wp_d = None

    def _kshape(self):
        return (self.k, self.k, self.cin, self.cout)

    def _fans(self):
        return self.k * self.k * self.cin, self.k * self.k * self.cout

    # strides (in floats) of the Keras kernel seen as [tap][in][out]
    def _strides_tio(self):
        return self.cin * self.cout, self.cout, 1

    def on_finalize(self, device):
        T = self.k * self.k
        self.wp_f = torch.zeros((roundup(self.cout_p, 16), T * self.cin_p), dtype=BF16, device=device)
        self.wp_d = torch.zeros((roundup(self.cin_p, 16), T * self.cout_p), dtype=BF16, device=device)
        self.repack()

    def pack_jobs(self):
        """fp32 master kernel -> bf16 packed operands for the forward and the backward-data GEMMs (job descriptors)."""
        T = self.k * self.k
        sT, sI, sO = self._strides_tio()
        return [
            # forward: rows n = out channel, K = tap*cin_p + ci (times the folded BatchNorm scale of that channel, if any)
            ops.pack_job(self.kernel.data, sT, sO, sI, T, self.cout, self.cin, self.wp_f, T * self.cin_p, self.cin_p,
                         nscale=self.fold_scale()),
            # dgrad: rows n = in channel, K = tap*cout_p + co
            ops.pack_job(self.kernel.data, sT, sI, sO, T, self.cin, self.cout, self.wp_d, T * self.cout_p, self.cout_p)]

    def repack(self):
        jobs = self.pack_jobs()
        ops.pack_weights_batched(ops.make_pack_table(jobs, self.wp_f.device), len(jobs))

    def fold_scale(self):
        """The per-output-channel multiplier folded into the packed FORWARD operand (an inference BatchNormalization that follows
        this conv), or None.  The owner sets ``_fold_scale`` (a device vector kept up to date by usseg_bn_fold_batched) and
        ``_fold_bns`` (the layers it comes from: folding is only valid while they are in inference mode; toggling
        ``training_mode`` needs a ``repack()``)."""
        sc = getattr(self, "_fold_scale", None)
        if sc is None or any(b.training_mode for b in getattr(self, "_fold_bns", ())):
            return None
        return sc

    def forward(self, x, out=None, act=ACT_NONE, alpha=0.0, residual=None, out_f32=False, use_bias=True, scale=None, shift=None, bias=None):
        """``scale`` / ``shift``: a folded inference BatchNormalization applied in the epilogue (y = act(scale*conv + shift); ``shift``
        holds the bias).  ``bias``: replaces the layer's own bias vector (a BatchNorm folded into the packed operand brings its
        own shift: ``_fold_scale`` + ``bias=fold_shift``)."""
        B, H, W, C, _ = ops.geom(x)
        assert C == self.cin_p, f"expected {self.cin_p} physical input channels, got {C}"
        if out is None:
            out = (torch.empty((B, H, W, roundup(self.cout, 4)), dtype=torch.float32, device=x.device)
                   if out_f32 else ops.new_act(B, H, W, self.cout_p, x.device))
        self._x = x
        if scale is not None:
            return ops.conv2d_fwd(x, self.wp_f, shift, self.k, self.dil, out, act, alpha, residual, out_f32, scale=scale)
        b = bias if bias is not None else (self.bias.data if use_bias else None)
        return ops.conv2d_fwd(x, self.wp_f, b, self.k, self.dil, out, act, alpha, residual, out_f32)

    def wgrad_job(self, dy):
        """(x, dy, k, dilation, dw, dst_map) for ops.conv2d_wgrad_multi."""
        plain = self.cin_p == self.cin and self.cout_p == self.cout
        return (self._x, dy, self.k, self.dil, self.kernel.grad if plain else None, None if plain else self._wgrad_map())

    def backward(self, dy, need_dx=True, dx=None, dx_residual=None, accumulate_dx=False, skip_bias=False, skip_wgrad=False):
        """dy: gradient w.r.t. the conv output (before any fused activation).  Accumulates kernel/bias grads.
        ``skip_bias``: the bias gradient was already produced by the following norm layer's backward (its ``dbias``).
        ``skip_wgrad``: the caller runs the weight gradient itself (a multi-job launch with its sibling branches)."""
        x = self._x
        if not (skip_wgrad and skip_bias):
            def params():                         # parameter gradients: beside the backward-data chain
                if not skip_wgrad:
                    self._wgrad(x, dy)
                if not skip_bias:
                    ops.colsum(dy, self.bias.grad, self.cout)
            ops.wgrad_later(params, x, dy)
        if not need_dx:
            return None
        B, H, W, _, _ = ops.geom(x)
        dx = dx if dx is not None else ops.new_act(B, H, W, self.cin_p, dy.device)
        return ops.conv2d_dgrad(dy, self.wp_d, self.k, self.dil, dx, dx_residual, accumulate_dx)

    def _wgrad(self, x, dy):
        T = self.k * self.k
        if self.cin_p == self.cin and self.cout_p == self.cout:
            ops.conv2d_wgrad(x, dy, self.k, self.dil, self.kernel.grad)   # Keras layout == [T][Cin][Cout]: accumulate in place
            return
        ops.conv2d_wgrad_mapped(x, dy, self.k, self.dil, self._wgrad_map())   # padded channels: scatter the logical block, no scratch

    def _wgrad_map(self):
        m = getattr(self, "_wmap", None)
        if m is None or self._wmap_ptr != self.kernel.grad.data_ptr():
            sT, sI, sO = self._strides_tio()
            m = self._wmap = ops.wgrad_dst([(self.kernel.grad, sT, sI, sO, 0, 0, self.cin, self.cout)])
            self._wmap_ptr = self.kernel.grad.data_ptr()
        return m


class Conv2DTranspose(Conv2D):
    """tf.keras.layers.Conv2DTranspose(filters, k, strides=2, padding='same'), kernel [k,k,Cout,Cin], k in (3,4)."""
    transposed = True

    def __init__(self, in_channels: int, filters: int, kernel_size: int, init: str = "he"):
        super().__init__(in_channels, filters, kernel_size, 1, init)

    def _kshape(self):
        return (self.k, self.k, self.cout, self.cin)

    def _fans(self):  # Keras computes fans from shape[-2] (in) / shape[-1] (out) of [k,k,Cout,Cin]
        return self.k * self.k * self.cout, self.k * self.k * self.cin

    def _strides_tio(self):
        return self.cin * self.cout, 1, self.cin

    def forward(self, x, out=None, act=ACT_NONE, alpha=0.0, out_f32=False, use_bias=True):
        B, H, W, C, _ = ops.geom(x)
        assert C == self.cin_p
        if out is None:
            out = (torch.empty((B, 2 * H, 2 * W, roundup(self.cout, 4)), dtype=torch.float32, device=x.device)
                   if out_f32 else ops.new_act(B, 2 * H, 2 * W, self.cout_p, x.device))
        self._x = x
        return ops.tconv2d_fwd(x, self.wp_f, self.bias.data if use_bias else None, self.k, out, act, alpha, out_f32)

    def backward(self, dy, need_dx=True, dx=None, dx_residual=None, accumulate_dx=False):
        x = self._x
        T = self.k * self.k
        def params():
            ops.tconv2d_wgrad_mapped(x, dy, self.k, self._wgrad_map())
            ops.colsum(dy, self.bias.grad, self.cout)
        ops.wgrad_later(params, x, dy)
        if not need_dx:
            return None
        B, H, W, _, _ = ops.geom(x)
        dx = dx if dx is not None else ops.new_act(B, H, W, self.cin_p, dy.device)
        return ops.tconv2d_dgrad(dy, self.wp_d, self.k, dx, dx_residual, accumulate_dx)


class QuadHead:
    """A <= 4-class stride-2 Conv2DTranspose head (Decoder.py:120, k=3; TBI_ResNest.py:124, k=4) run in "quad" form: a
    stride-1 conv at the input resolution whose 16 channels are the four output parities (one HBM-bound launch each way
    instead of a 4-class gather GEMM and a per-tap weight gradient).  Logits / their gradient live in the space-to-depth
    layout [B,h,w,16] that ``ops.softmax_loss(..., quad_w=W)`` indexes."""

    def __init__(self, head: "Conv2DTranspose"):
        assert head.cout <= 4 and head.k in (3, 4)
        self.head = head
        head.on_finalize = lambda device: None      # packed here instead

    def on_finalize(self, device):
        cin_p = self.head.cin_p
        self.wq_f = torch.zeros((16, 9 * cin_p), dtype=BF16, device=device)
        self.wq_d = torch.zeros((roundup(cin_p, 16), 9 * 16), dtype=BF16, device=device)
        self.bias16 = torch.zeros(16, dtype=torch.float32, device=device)
        self._dq16 = torch.zeros(9 * cin_p * 16 + 16, dtype=torch.float32, device=device)      # one buffer: one zero-fill launch
        self.dq, self.d16 = self._dq16[:9 * cin_p * 16], self._dq16[9 * cin_p * 16:]
        jobs = self.pack_jobs()
        ops.pack_weights_batched(ops.make_pack_table(jobs, device), len(jobs))

    def pack_jobs(self):
        """Quad operands from the Keras kernel [k,k,Cout,Cin]: tap kh feeds output parity a = (kh+pad)&1 from source offset
        di = (a+pad-kh)/2 (pad = 1 for k=4); rows = parity class*4 + n, stencil tap = (di+1)*3 + (dj+1)."""
        h, k = self.head, self.head.k
        pad = 1 if k == 4 else 0
        jobs = []
        for kh in range(k):
            for kw in range(k):
                src = h.kernel.data[kh, kw]                       # [Cout][Cin] view
                a, b = (kh + pad) & 1, (kw + pad) & 1
                t = ((a + pad - kh) // 2 + 1) * 3 + ((b + pad - kw) // 2 + 1)
                cls = a * 2 + b
                jobs.append(ops.pack_job(src, 0, h.cin, 1, 1, h.cout, h.cin, self.wq_f, 9 * h.cin_p, h.cin_p, cls * 4, t * h.cin_p))
                jobs.append(ops.pack_job(src, 0, 1, h.cin, 1, h.cin, h.cout, self.wq_d, 9 * 16, 16, 0, t * 16 + cls * 4))
        return jobs

    def forward(self, x):
        B, H, W, _, _ = ops.geom(x)
        ops.quad_bias_expand(self.head.bias.data, self.head.cout, self.bias16)
        logits = torch.empty((B, H, W, 16), dtype=torch.float32, device=x.device)
        self._x = self.head._x = x
        return ops.conv2d_fwd(x, self.wq_f, self.bias16, 3, 1, logits, out_f32=True)

    def forward_loss(self, x, y_true, probs, loss, dlogits, **kw) -> bool:
        """The head conv, its softmax and the loss in ONE launch (csrc/attn_loss_optim.hip head_quad_loss_kernel); leaves the state ``forward``
        leaves.  False - nothing done - when there is no fused kernel for this size (ordered-sum slots, LDS)."""
        if not ops.head_quad_softmax_loss(x, self.wq_f, self.head.bias.data, self.head.cout, y_true, probs, loss, dlogits, **kw):
            return False
        self._x = self.head._x = x
        return True

    def backward(self, dl4, need_dx=True):
        """dl4: bf16 [B,h,w,16] gradient of the quad-form logits -> dx [B,h,w,cin]."""
        h, x = self.head, self._x
        def params():
            ops.fill_f32(self._dq16, 0.0)
            ops.conv2d_wgrad(x, dl4, 3, 1, self.dq)
            ops.colsum(dl4, self.d16, 16)

            def fold():     # dq / d16 are complete once the deferred finishing reductions have run: fold them into the Keras variables then
                ops.quad_head_fold(self.dq, h.cin_p, h.cin, h.cout, 4, h.k, h.kernel.grad, self.d16, h.bias.grad)
            ops.after_flush(fold)
        ops.wgrad_later(params, x, dl4)
        if not need_dx:
            return None
        B, H, W, _, _ = ops.geom(x)
        return ops.conv2d_dgrad(dl4, self.wq_d, 3, 1, ops.new_act(B, H, W, h.cin_p, x.device))


class QuadTConv:
    """Any stride-2 'same' Conv2DTranspose (k = 3 or 4) as tap-masked 3x3 stride-1 convs on space-to-depth tensors: the four
    output parities become 4*Np channels at the INPUT resolution, each parity class uses only its <= 2x2 stencil taps (the
    conv kernels skip the rest), so the work is exactly that of the transposed conv while forward / backward-data run on the
    LDS-DMA conv kernel and the weight gradient on the halo-tile kernel (instead of the 4-class gather GEMM and the per-tap
    weight gradient that re-reads both operands once per tap: 16x for k=4)."""

    def __init__(self, layer: "Conv2DTranspose"):
        assert layer.k in (3, 4)
        self.layer, self.Np = layer, layer.cout_p
        layer.on_finalize = lambda device: None      # packed here instead

    def on_finalize(self, device):
        h, Np = self.layer, self.Np
        self.wq_f = torch.zeros((roundup(4 * Np, 16), 9 * h.cin_p), dtype=BF16, device=device)
        self.wq_d = torch.zeros((roundup(h.cin_p, 16), 9 * 4 * Np), dtype=BF16, device=device)
        self.bias_q = torch.zeros(4 * Np, dtype=torch.float32, device=device)
        self.dq = torch.zeros(9 * h.cin_p * 4 * Np, dtype=torch.float32, device=device)
        jobs = self.pack_jobs()
        ops.pack_weights_batched(ops.make_pack_table(jobs, device), len(jobs))

    def pack_jobs(self):
        h, k, Np = self.layer, self.layer.k, self.Np
        pad = 1 if k == 4 else 0
        jobs = []
        for kh in range(k):
            for kw in range(k):
                src = h.kernel.data[kh, kw]                       # [Cout][Cin] view
                a, b = (kh + pad) & 1, (kw + pad) & 1
                t = ((a + pad - kh) // 2 + 1) * 3 + ((b + pad - kw) // 2 + 1)
                cls = a * 2 + b
                jobs.append(ops.pack_job(src, 0, h.cin, 1, 1, h.cout, h.cin, self.wq_f, 9 * h.cin_p, h.cin_p, cls * Np, t * h.cin_p))
                jobs.append(ops.pack_job(src, 0, 1, h.cin, 1, h.cin, h.cout, self.wq_d, 9 * 4 * Np, 4 * Np, 0, t * 4 * Np + cls * Np))
        return jobs

    def forward(self, x, out=None):
        """x [B,H,W,cin] -> raw tconv output [B,2H,2W,cout_p] (bias added), optionally written into the slice ``out``."""
        B, H, W, _, _ = ops.geom(x)
        h = self.layer
        ops.quad_bias_expand(h.bias.data, h.cout, self.bias_q)
        y4 = ops.tconv_quad_fwd(x, self.wq_f, self.bias_q, h.k, self.Np, ops.new_act(B, H, W, 4 * self.Np, x.device))
        out = out if out is not None else ops.new_act(B, 2 * H, 2 * W, h.cout_p, x.device)
        ops.space_to_depth2(out, y4, self.Np, to_quad=False)
        self._x = h._x = x
        return out

    def backward(self, dy, need_dx=True, bias_grad=True):
        """dy [B,2H,2W,cout_p] (slice views allowed) -> dx; accumulates kernel (and bias) gradients."""
        h, x = self.layer, self._x
        B, H, W, _, _ = ops.geom(x)
        dy4 = ops.new_act(B, H, W, 4 * self.Np, x.device)
        ops.space_to_depth2(dy, dy4, self.Np, to_quad=True)
        def params():
            ops.fill_f32(self.dq, 0.0)
            ops.tconv_quad_wgrad(x, dy4, h.k, self.Np, self.dq)
            ops.defer_flush()                                      # dq is read right away
            ops.tconv_quad_unpack(self.dq, h.cin_p, h.cin, h.cout, self.Np, h.k, h.kernel.grad)
            if bias_grad:
                ops.colsum(dy, h.bias.grad, h.cout)
        ops.wgrad_later(params, x, dy4, dy)
        if not need_dx:
            return None
        return ops.tconv_quad_dgrad(dy4, self.wq_d, h.k, self.Np, ops.new_act(B, H, W, h.cin_p, x.device))


# ------------------------------------------------------------------------------------------------ normalisation
class LayerNormalization(nn.Module):
    """tf.keras.layers.LayerNormalization(axis=-1, epsilon=1e-3): per pixel over channels; optional fused activation."""
    mode = 0

    def __init__(self, channels: int, epsilon: float = KERAS_LN_EPS):
        super().__init__()
        self.C, self.eps = channels, epsilon
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def _stats(self):
        return None, None

    def forward(self, x, act=ACT_NONE, alpha=0.0, out=None):
        out = out if out is not None else torch.empty_like(x)
        self._x, self._act = x, (act, alpha)
        mean, var = self._stats()
        return ops.norm_act_fwd(x, self.C, self.gamma.data, self.beta.data, out, self.mode, 1, self.eps, act, alpha, mean, var)

    def backward(self, dy, dx=None, dbias=None):
        """``dbias``: optional fp32 vector that receives sum_pixels(dx) = the bias gradient of the conv that produced x."""
        x = self._x
        act, alpha = self._act
        dx = dx if dx is not None else torch.empty_like(x)
        mean, var = self._stats()
        return ops.norm_act_bwd(x, dy, self.C, self.gamma.data, self.beta.data, dx, self.gamma.grad, self.beta.grad, self.mode,
                                1, self.eps, act, alpha, mean, var, dbias)


    def backward_residual(self, dy, dres, dbias=None):
        """Backward of x + f(norm(x)) w.r.t. x: LN'(dy) + dres in one pass (the bits of backward() followed by an accumulating copy).
        ``dbias`` (optional) += the column sums of the result: the bias gradient of the Dense layer that produced x's last term."""
        x = self._x
        assert self.mode == 0 and self._act[0] == ACT_NONE
        if not _FUSE_LN_RES:
            dx = self.backward(dy)
            ops.copy_channels(dres, dx, accumulate=True)
            if dbias is not None:
                ops.colsum(dx, dbias, self.C)
            return dx
        return ops.norm_act_bwd_res(x, dy, self.C, self.gamma.data, self.beta.data, dres, torch.empty_like(x), self.gamma.grad, self.beta.grad,
                                    self.eps, dbias)


class BatchNormalization(LayerNormalization):
    """tf.keras.layers.BatchNormalization(axis=-1, momentum=0.99, epsilon=1e-3).

    As the reference is driven the layer runs in INFERENCE mode: a per-channel affine with the
    moving statistics, gamma/beta trainable.  ``training_mode=True`` switches to batch statistics (two-pass) and
    updates the moving statistics with momentum 0.99.
    """
    mode = 1

    def __init__(self, channels: int, epsilon: float = KERAS_BN_EPS, momentum: float = 0.99):
        super().__init__(channels, epsilon)
        self.momentum = momentum
        self.training_mode = False
        cp = roundup(channels, 8)
        self.register_buffer("moving_mean_p", torch.zeros(cp))
        self.register_buffer("moving_variance_p", torch.ones(cp))
        # inference mode folded into the producing conv's epilogue: y = act(fold_scale*conv + fold_shift) (ops.bn_fold_job)
        self.register_buffer("fold_scale", torch.ones(cp))
        self.register_buffer("fold_shift", torch.zeros(cp))

    def fold_job(self, conv_bias):
        return ops.bn_fold_job(self.gamma.data, self.beta.data, self.moving_mean_p, self.moving_variance_p, conv_bias,
                               self.fold_scale, self.fold_shift, self.C, self.eps)

    def backward_folded(self, y, dy, act, alpha, dx=None, dbias=None):
        """Backward of conv-epilogue-folded BN + activation from the ACTIVATED output y (the pre-norm tensor was never stored)."""
        dx = dx if dx is not None else torch.empty_like(y)
        return ops.norm_act_bwd(y, dy, self.C, self.gamma.data, self.beta.data, dx, self.gamma.grad, self.beta.grad, 2, 1, self.eps,
                                act, alpha, self.moving_mean_p, self.moving_variance_p, dbias)

    @property
    def moving_mean(self):
        return self.moving_mean_p[:self.C]

    @property
    def moving_variance(self):
        return self.moving_variance_p[:self.C]

    def _stats(self):
        if self.training_mode:
            return self._bmean, self._bvar
        return self.moving_mean_p, self.moving_variance_p

    def forward(self, x, act=ACT_NONE, alpha=0.0, out=None):
        if self.training_mode:
            # Keras training=True: normalise with the batch mean / biased variance and move the running statistics
            B, H, W, _, _ = ops.geom(x)
            cp, dev = self.moving_mean_p.numel(), x.device
            s, s2 = torch.zeros(cp, device=dev), torch.zeros(cp, device=dev)
            ops.channel_stats(x, self.C, s, s2)
            self._bmean, self._bvar = torch.zeros(cp, device=dev), torch.ones(cp, device=dev)
            ops.bn_finalize_stats(s, s2, B * H * W, self.C, self.momentum, self._bmean, self._bvar, self.moving_mean_p, self.moving_variance_p)
        return super().forward(x, act, alpha, out)

    def forward_pool(self, x, act=ACT_NONE, alpha=0.0, out=None):
        """BN (inference mode) + activation + AveragePooling2D(2,2) in one launch: -> the POOLED tensor [B,H/2,W/2,C].  Only for
        a BN whose activated output feeds nothing but the pool (the stems: ResNest.py:45-47, TBI_ResNest.py:90-92)."""
        assert not self.training_mode
        self._x, self._act = x, (act, alpha)
        return ops.bn_act_pool_fwd(x, self.C, self.gamma.data, self.beta.data, self.moving_mean_p, self.moving_variance_p, self.eps, act, alpha, out)

    def backward_pool(self, dy_pooled, dbias=None):
        """Backward of ``forward_pool``: dy w.r.t. the pooled output -> dx w.r.t. the pre-norm input (no upsampled gradient tensor)."""
        x = self._x
        act, alpha = self._act
        return ops.bn_act_pool_bwd(x, dy_pooled, self.C, self.gamma.data, self.beta.data, self.moving_mean_p, self.moving_variance_p, self.eps,
                                   act, alpha, torch.empty_like(x), self.gamma.grad, self.beta.grad, dbias)

    def backward(self, dy, dx=None, dbias=None):
        if not self.training_mode:
            return super().backward(dy, dx, dbias)
        x = self._x
        act, alpha = self._act
        dx = dx if dx is not None else torch.empty_like(x)
        cp = self.moving_mean_p.numel()
        tg, tb = torch.zeros(cp, device=x.device), torch.zeros(cp, device=x.device)
        ops.norm_act_bwd(x, dy, self.C, self.gamma.data, self.beta.data, dx, tg, tb, 1, 1, self.eps, act, alpha, self._bmean, self._bvar)
        ops.defer_flush()                      # tg / tb are read right away
        ops.bn_train_bwd_fix(x, dx, self.C, self.gamma.data, self._bmean, self._bvar, self.eps, tg, tb)
        self.gamma.grad.add_(tg[:self.C])      # tiny [C] vectors: host-side glue, not a compute kernel of the path
        self.beta.grad.add_(tb[:self.C])
        return dx                              # (a conv bias feeding a batch-statistics BN has exactly zero gradient: dbias untouched)
