"""Windowed-attention encoder: the module surface of the reference's ``SwinTransformer.py`` (a port of
rishigami/Swin-Transformer-TF; BASELINE configs[4]).

Same class names, constructor arguments and return structures as ``SwinTransformer.py`` - ``Mlp`` (:24-38),
``WindowAttention`` (:60-141), ``SwinTransformerBlock`` (:162-261), ``PatchMerging`` (:264-290), ``BasicLayer`` (:293-339),
``PatchEmbed`` (:341-365), ``SwinTransformerModel`` (:372-455) and the ``CFGS`` table (:8-21) - with explicit ``forward`` /
``backward`` on hand-written gfx950 kernels.  Tokens are kept as NHWC bf16 tensors [B,H,W,C] (the reference's [B, L=H*W, C]
row-major), so:

  * Dense layers are 1x1 convolutions on the LDS-DMA GEMM kernels (bias, exact GELU and the residual adds fused in their
    epilogues);
  * the cyclic shift, window partition and their inverses (:40-58,226-253) are never materialised: ``usseg_window_attn_*`` reads
    each window's q / k / v straight from the token tensor with that index arithmetic, adds the relative-position bias (:94-104)
    and the shifted-window mask (:192-217) on the fly, and writes the result back to the tokens' own positions;
  * PatchEmbed's Conv2D(kernel = stride = patch) (:352) is a space-to-depth of the image (``usseg_patchify``) + a 1x1 GEMM whose
    kernel IS the Keras variable [p,p,Cin,E] read as [p*p*Cin, E];
  * PatchMerging's strided gather + concat (:280-284) is one pass (``usseg_patch_merge``), the 4C-wide LayerNorm runs on the
    wide-row kernel (``usseg_ln_wide_*``, up to 4096 channels).

As the reference is driven, Dropout and DropPath are the identity (no ``training`` argument reaches them, :25,144-155).
Deviations, on purpose: only SQUARE windows are accepted - ``window_reverse`` (:53-58) mixes window_size[0] and [1] and is an
inverse of ``window_partition`` only for square windows, and every entry of CFGS is square (the constructor default [4, 5] of :376
would scramble the tensor in the reference); window sides 2, 4 or 8 (the CFGS use 4 and 8); the factory ``SwinTransformer()`` (:458-486, broken as committed and tied to a download) builds the
named configuration without weights.  ``self.features`` does not accumulate across calls (:434,449 appends forever).
The storage type is bf16 (the build's choice for every path; BASELINE quotes fp16 for this configuration).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .step import TrainStepDriver
from . import ops
from .flat import AdamClip, FlatParams
from .layers import _FUSE_LN_RES, Conv2D
from .ops import ACT_GELU, ACT_NONE, BF16, roundup

_LAZY_BLOCK = os.environ.get("USSEG_SWIN_LAZY", "1") != "0"     # per-block lazy weight gradients (ops.lazy_wgrads)

CFGS = {   # SwinTransformer.py:8-21
    "swin_tiny_224": dict(input_size=(224, 224), window_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24]),
    "swin_small_224": dict(input_size=(224, 224), window_size=4, embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24]),
    "swin_base_224": dict(input_size=(224, 224), window_size=4, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32]),
    "swin_base_384": dict(input_size=(384, 384), window_size=8, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32]),
    "swin_large_224": dict(input_size=(224, 224), window_size=4, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48]),
    "swin_large_384": dict(input_size=(384, 384), window_size=8, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48]),
}
LN_EPS = 1e-5   # every LayerNormalization of the file (:180,187,270,343,419)


def _square(window_size) -> int:
    if isinstance(window_size, int):
        return window_size
    ws = list(window_size)
    if len(ws) == 2 and ws[0] != ws[1]:
        raise ValueError("non-square windows: window_reverse (SwinTransformer.py:53-58) is only an inverse of window_partition for square "
                         "windows; every CFGS entry is square")
    return int(ws[0])


class LayerNorm(nn.Module):
    """tf.keras.layers.LayerNormalization(epsilon=1e-5) over the channel axis of [B,H,W,C] tokens, C up to 4096."""

    def __init__(self, channels: int, epsilon: float = LN_EPS):
        super().__init__()
        assert channels % 8 == 0, "token widths are multiples of 8 in every configuration"
        self.C, self.eps = channels, epsilon
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    # Up to 512 channels the packed per-pixel LayerNorm of the conv path runs (several token rows per wave: at 96 channels a
    # one-row-per-wave kernel keeps 12 of 64 lanes busy); wider rows (the merged 4C tokens, stage 4) use the wide kernel.
    def forward(self, x):
        self._x = x
        if self.C <= 512:
            return ops.norm_act_fwd(x, self.C, self.gamma.data, self.beta.data, torch.empty_like(x), 0, 1, self.eps)
        return ops.ln_wide_fwd(x, self.gamma.data, self.beta.data, self.eps, torch.empty_like(x))

    def backward(self, dy):
        if self.C <= 512:
            return ops.norm_act_bwd(self._x, dy, self.C, self.gamma.data, self.beta.data, torch.empty_like(self._x), self.gamma.grad, self.beta.grad,
                                    0, 1, self.eps)
        return ops.ln_wide_bwd(self._x, dy, self.gamma.data, self.eps, torch.empty_like(self._x), self.gamma.grad, self.beta.grad)

    def backward_residual(self, dy, dres, dbias=None):
        """Backward of x + f(norm(x)) w.r.t. x; one pass up to 512 channels (the bits of backward() + an accumulating copy).
        ``dbias`` (optional) += the column sums of the result: the bias gradient of the Dense layer that produced x's last term."""
        if self.C <= 512 and _FUSE_LN_RES:
            return ops.norm_act_bwd_res(self._x, dy, self.C, self.gamma.data, self.beta.data, dres, torch.empty_like(self._x), self.gamma.grad,
                                        self.beta.grad, self.eps, dbias)
        dx = self.backward(dy)
        ops.copy_channels(dres, dx, accumulate=True)
        if dbias is not None:
            ops.colsum(dx, dbias, self.C)
        return dx


class Dense(Conv2D):
    """tf.keras.layers.Dense on the channel axis = a 1x1 convolution (Glorot-uniform kernel [in, out], zero bias)."""

    def __init__(self, in_features: int, units: int, use_bias: bool = True):
        super().__init__(in_features, units, 1, init="glorot")
        self.use_bias = use_bias
        if not use_bias:
            del self._parameters["bias"]
            self.bias = None

    def forward(self, x, act=ACT_NONE, residual=None):
        return super().forward(x, act=act, alpha=0.0, residual=residual, use_bias=self.use_bias)

    def backward(self, dy, need_dx=True):
        return super().backward(dy, need_dx=need_dx, skip_bias=not self.use_bias)


class Mlp(nn.Module):
    """SwinTransformer.py:24-38: fc1 -> exact GELU -> fc2 (dropouts are the identity as driven)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0, prefix=""):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Dense(in_features, hidden_features)
        self.fc2 = Dense(hidden_features, out_features)

    def forward(self, x, residual=None):
        self._h = self.fc1.forward(x)                                                   # :33 (pre-activation kept for the backward)
        g = ops.act_fwd(self._h, torch.empty_like(self._h), ACT_GELU, 0.0)              # :34
        return self.fc2.forward(g, residual=residual)                                   # :36

    def backward(self, dy, fc2_bias_done=False):
        dg = Conv2D.backward(self.fc2, dy, skip_bias=fc2_bias_done)
        # GELU backward + fc1's bias gradient (column sums of its output) in one pass
        dh = ops.act_bwd_colsum(self._h, dg, torch.empty_like(dg), ACT_GELU, 0.0, self.fc1.bias.grad, self.fc1.cout)
        return Conv2D.backward(self.fc1, dh, skip_bias=True)


class WindowAttention(nn.Module):
    """SwinTransformer.py:60-141.  ``forward(x, shift)`` takes the TOKEN tensor [B,H,W,C] of the whole image (not partitioned
    windows): partition, shift and mask are index arithmetic inside the kernel."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0, prefix=""):
        super().__init__()
        assert qk_scale is None, "qk_scale: the reference's default head_dim ** -0.5 is built into the kernel"
        self.dim, self.window_size, self.num_heads, self.prefix = dim, _square(window_size), num_heads, prefix
        self.qkv = Dense(dim, dim * 3, use_bias=qkv_bias)                               # :72
        self.proj = Dense(dim, dim)                                                     # :75
        T = 2 * self.window_size - 1
        self.relative_position_bias_table = nn.Parameter(torch.zeros(T * T, num_heads))   # :78-82 (Zeros initialiser)

    def forward(self, x, shift=0, residual=None):
        B, H, W, C, _ = ops.geom(x)
        self._qkv = self.qkv.forward(x)                                                 # :106-109 (head split = channel slices)
        self._shift = shift
        ctx = ops.window_attn_fwd(self._qkv, self.relative_position_bias_table.data, self.num_heads, self.window_size, shift,
                                  ops.new_act(B, H, W, C, x.device))                    # :110-136
        return self.proj.forward(ctx, residual=residual)                                # :137 (+ shortcut, :256)

    def backward(self, dy, proj_bias_done=False):
        dctx = Conv2D.backward(self.proj, dy, skip_bias=proj_bias_done)
        dqkv = ops.window_attn_bwd(self._qkv, dctx, self.relative_position_bias_table.data, self.num_heads, self.window_size, self._shift,
                                   torch.empty_like(self._qkv), self.relative_position_bias_table.grad)
        return self.qkv.backward(dqkv)


class SwinTransformerBlock(nn.Module):
    """SwinTransformer.py:162-261 on token tensors [B,H,W,C]."""

    def __init__(self, dim, input_resolution, num_heads, window_size=4, shift_size=0, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path_prob=0.0, norm_layer=None, prefix=""):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads, self.mlp_ratio = dim, tuple(input_resolution), num_heads, mlp_ratio
        ws = _square(window_size)
        if min(self.input_resolution) <= ws:                                            # :172-175: one window, no shift
            shift_size, ws = 0, min(self.input_resolution)
        assert 0 <= shift_size < ws, "shift_size must in 0-window_size"                 # :176
        self.window_size, self.shift_size = ws, shift_size
        self.norm1 = LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size=ws, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, prefix=prefix)
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), prefix=prefix)

    def forward(self, x):
        a = self.attn.forward(self.norm1.forward(x), self.shift_size, residual=x)       # :224-256 shortcut + attention
        return self.mlp.forward(self.norm2.forward(a), residual=a)                      # :257 x + mlp(norm2(x))

    def backward(self, dy, fc2_bias_done=False, prev_fc2_bias=None):
        """``fc2_bias_done``: the kernel that produced dy also summed it into this block's fc2 bias gradient; ``prev_fc2_bias``: the
        bias gradient of the fc2 whose output (+ shortcut) is this block's input - summed by the kernel that produces dx."""
        # + the residual branch of :257; the same pass sums da over the tokens = the attention projection's bias gradient
        da = self.norm2.backward_residual(self.mlp.backward(dy, fc2_bias_done), dy, dbias=self.attn.proj.bias.grad)
        return self.norm1.backward_residual(self.attn.backward(da, proj_bias_done=True), da, dbias=prev_fc2_bias)   # + the residual branch of :256


class PatchMerging(nn.Module):
    """SwinTransformer.py:264-290: 2x2 neighbourhood -> 4C channels (order x0=(0,0), x1=(1,0), x2=(0,1), x3=(1,1)), LN, Dense(2C, no bias)."""

    def __init__(self, input_resolution, dim, norm_layer=None, prefix=""):
        super().__init__()
        self.input_resolution, self.dim = tuple(input_resolution), dim
        self.reduction = Dense(4 * dim, 2 * dim, use_bias=False)
        self.norm = LayerNorm(4 * dim)

    def forward(self, x):
        B, H, W, C, _ = ops.geom(x)
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."             # :276
        self._shape = (B, H, W, C)
        m = ops.new_act(B, H // 2, W // 2, 4 * C, x.device)
        ops.patch_merge(x, m)
        return self.reduction.forward(self.norm.forward(m))

    def backward(self, dy):
        B, H, W, C = self._shape
        dm = self.norm.backward(self.reduction.backward(dy))
        dx = ops.new_act(B, H, W, C, dy.device)
        ops.patch_merge(dx, dm, backward=True)
        return dx


class BasicLayer(nn.Module):
    """SwinTransformer.py:293-339: ``forward`` -> (x after the optional downsample, the stage output before it or None)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path_prob=0.0, norm_layer=None, downsample=None, use_checkpoint=False, prefix=""):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, tuple(input_resolution), depth
        ws = _square(window_size)
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=ws,
                                 shift_size=0 if i % 2 == 0 else ws // 2, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                 prefix=f"{prefix}/blocks{i}") for i in range(depth)])   # :309-321
        self.downsample = downsample(input_resolution, dim=dim, prefix=prefix) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk.forward(x)
        if self.downsample is not None:
            return self.downsample.forward(x), x                                        # :335-337
        return x, None

    def backward(self, dy, dfeature=None):
        if self.downsample is not None:
            dy = self.downsample.backward(dy)
            if dfeature is not None:
                ops.copy_channels(dfeature, dy, accumulate=True)
        done = False
        for i in reversed(range(len(self.blocks))):
            prev = self.blocks[i - 1].mlp.fc2.bias.grad if i > 0 else None
            with ops.lazy_wgrads() if _LAZY_BLOCK else contextlib.nullcontext():    # this block's weight gradients run beside the next block's backward pass
                dy = self.blocks[i].backward(dy, fc2_bias_done=done, prev_fc2_bias=prev)
            done = prev is not None
        return dy


class PatchEmbed(nn.Module):
    """SwinTransformer.py:341-365: Conv2D(embed_dim, kernel = stride = patch) + LayerNorm, as space-to-depth + 1x1 GEMM."""

    def __init__(self, img_size=(224, 224), patch_size=(4, 4), in_chans=3, embed_dim=96, norm_layer=True):
        super().__init__()
        ps = patch_size if isinstance(patch_size, int) else patch_size[0]
        self.img_size, self.patch_size = tuple(img_size), ps
        self.patches_resolution = [img_size[0] // ps, img_size[1] // ps]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = Dense(ps * ps * in_chans, embed_dim)       # kernel [1,1,p*p*Cin,E] == the Keras [p,p,Cin,E] variable, same memory order
        self.norm = LayerNorm(embed_dim) if norm_layer else None

    def forward(self, x):
        B, H, W, C = x.shape
        assert (H, W) == self.img_size, f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."   # :358-359
        t = self.proj.forward(ops.patchify(x.contiguous(), self.patch_size))
        return self.norm.forward(t) if self.norm is not None else t

    def backward(self, dy):
        if self.norm is not None:
            dy = self.norm.backward(dy)
        self.proj.backward(dy, need_dx=False)                  # the image needs no gradient


class SwinTransformerModel(TrainStepDriver, nn.Module):
    """SwinTransformer.py:372-455.  ``forward(x)`` -> (pooled features [B, num_features] fp32, or logits with include_top;
    [token tensor of every stage but the last, as [B, L, C] views]).  ``backward(d_out, d_features=None)`` accumulates every
    parameter gradient.  Owns its parameters in one flat fp32 buffer (flat.FlatParams) with a clip + Adam optimiser, like the
    other model wrappers."""

    def __init__(self, model_name="swin_large_384", include_top=False, img_size=(224, 224), patch_size=(4, 4), in_chans=3, num_classes=1000,
                 embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=4, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=None, ape=False, patch_norm=True, use_checkpoint=False, *,
                 device: Optional[str] = None, seed: Optional[int] = 0, learning_rate: float = 1e-3, **kwargs):
        super().__init__()
        if seed is not None:
            torch.manual_seed(seed)
        self.include_top, self.num_classes, self.num_layers, self.embed_dim = include_top, num_classes, len(depths), embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, norm_layer=patch_norm)
        self.patches_resolution = self.patch_embed.patches_resolution
        res = self.patches_resolution
        # absolute position embedding (:401-408, off by default): [1, L, E], Zeros initialiser; its bf16 copy is rebuilt with the operands
        self.ape = bool(ape)
        if self.ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, res[0] * res[1], embed_dim))
        self.basic_layers = nn.ModuleList([
            BasicLayer(dim=int(embed_dim * 2 ** i), input_resolution=(res[0] // 2 ** i, res[1] // 2 ** i), depth=depths[i], num_heads=num_heads[i],
                       window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                       downsample=PatchMerging if i < self.num_layers - 1 else None, prefix=f"layers{i}") for i in range(self.num_layers)])   # :406-424
        self.norm = LayerNorm(self.num_features)
        self.head = Dense(self.num_features, num_classes) if include_top else None
        dev = device or ("cuda" if torch.cuda.is_available() else None)
        if dev is None:
            raise RuntimeError("SwinTransformerModel needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device(dev)
        self.flat = FlatParams(self, self.device)
        self.optimizer = AdamClip(self.flat, lr=learning_rate, clip_norm=None)
        self.features: List[torch.Tensor] = []

    def repack(self):
        """Every bf16 GEMM operand rebuilt from the fp32 masters in ONE batched launch (device-resident job table)."""
        from .VisionTransformer import repack_all
        repack_all(self)
        if self.ape:
            self._ape_bf16 = self.absolute_pos_embed.data.to(BF16).reshape(1, *self.patches_resolution, self.embed_dim)

    # ---- the shared step driver (step.TrainStepDriver): zero -> forward -> backward -> [per-replica clip -> exchange] -> Adam -> repack.
    # The reference defines no loss for this file (nothing imports it), so the "label" of a step is the upstream gradient of the
    # first output (pooled features, or logits with include_top).
    def _prep_x(self, x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        x = x.to(self.device)
        return (x if x.dtype in (torch.float32, torch.float64) else x.float()).contiguous()

    def _prep_y(self, y):
        if not torch.is_tensor(y):
            y = torch.as_tensor(y)
        return y.to(device=self.device, dtype=torch.float32).contiguous()

    def _zero_grad(self):
        self.flat.zero_grad()

    def _forward_backward(self, x, y):
        out, _ = self.forward(x)
        self._loss = (out * y).sum()          # the linear functional whose gradient w.r.t. the output is ``y``
        self.backward(y)
        return out

    def _repack(self):
        self.repack()

    def _graph_state(self):
        return [self.flat.flat] + self.optimizer.state_tensors()

    def train_step(self, x, d_out):
        """One optimisation step driven by the upstream gradient ``d_out`` of the first output: -> (sum(out * d_out), out); after
        ``capture_graph`` ``out`` is the captured step's static buffer."""
        x, d_out = self._prep_x(x), self._prep_y(d_out)
        out = self._graph_replay(x, d_out) if self._graph is not None else self._train_body(x, d_out)
        return self._loss.clone(), out

    def forward_features(self, x):
        """-> pooled [B, num_features] fp32 (:437-452)."""
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        x = x.to(self.device)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.float()
        t = self.patch_embed.forward(x)
        if self.ape:                                   # x + absolute_pos_embed (:442-443), broadcast over the batch
            for b in range(t.shape[0]):
                ops.copy_channels(self._ape_bf16, t[b:b + 1], accumulate=True)
        self.features = []
        for i, layer in enumerate(self.basic_layers):
            t, f = layer.forward(t)
            if i < self.num_layers - 1:
                B, H, W, C, _ = ops.geom(f)
                self.features.append(f.reshape(B, H * W, C))
        self._last = self.norm.forward(t)
        return ops.token_mean_fwd(self._last)

    def forward(self, x):
        p = self.forward_features(x)
        if self.include_top:
            B, F = p.shape
            self._pooled = p.to(BF16).reshape(B, 1, 1, F)
            p = ops.to_f32(self.head.forward(self._pooled), self.num_classes).reshape(B, self.num_classes)
        return p, self.features

    def backward(self, d_out, d_features=None):
        """d_out: fp32 gradient of the first output ([B, num_features], or [B, num_classes] with include_top); d_features: optional
        list of bf16 gradients of the stage outputs."""
        d_out = d_out.to(device=self.device, dtype=torch.float32).contiguous()
        if self.include_top:
            B = d_out.shape[0]
            dy = torch.zeros((B, 1, 1, roundup(self.num_classes, 8)), dtype=BF16, device=self.device)
            dy[..., :self.num_classes] = d_out.reshape(B, 1, 1, -1).to(BF16)
            d_out = ops.to_f32(self.head.backward(dy)).reshape(B, self.num_features)
        with ops.overlap_region():
            d = self.norm.backward(ops.token_mean_bwd(d_out, self._last))
            for i in reversed(range(self.num_layers)):
                df = None
                if d_features is not None and i < self.num_layers - 1 and d_features[i] is not None:
                    r = self.basic_layers[i].input_resolution
                    df = d_features[i].reshape(d.shape[0], r[0], r[1], -1)
                d = self.basic_layers[i].backward(d, df)
            if self.ape:                               # sum over the batch (an off-by-default option: host-side reduction of the fp32 cast)
                self.absolute_pos_embed.grad += ops.to_f32(d).sum(0).reshape(self.absolute_pos_embed.shape)
            self.patch_embed.backward(d)

    def __call__(self, x, *args, **kwargs):
        return self.forward(x)


def SwinTransformer(model_name="swin_large_384", num_classes=1000, include_top=True, pretrained=False, use_tpu=False, cfgs=CFGS, **kw):
    """SwinTransformer.py:458-486 without the download: builds the named configuration (random initialisation)."""
    if pretrained:
        raise ValueError("pretrained weights are fetched from a URL in the reference (:469-471); there is no network here")
    cfg = cfgs[model_name]
    return SwinTransformerModel(model_name=model_name, include_top=include_top, num_classes=num_classes, img_size=cfg["input_size"],
                                window_size=cfg["window_size"], embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"], **kw)
