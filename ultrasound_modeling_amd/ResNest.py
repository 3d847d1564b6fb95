"""ResNeSt-style split-attention encoder: the module surface of the reference's ``ResNest.py``.

Same class names, constructor arguments, attribute names, NHWC layout and return structure as
``ResNest.py:4-203`` of silverlight6/Ultrasound_Modeling, on hand-written gfx950 kernels.

How a ``residual_S`` stage is executed (ResNest.py:89-104):
  * the ``kpaths`` cardinal blocks read the same input, so their 1x1 convs are ONE GEMM with
    N = kpaths*cv11 output channels and their 3x3 convs are ONE grouped conv (groups = kpaths),
    issued as a block-diagonal implicit GEMM inside the MFMA tile (group widths 3..85 are below the
    MFMA K granule, so a per-group launch could not fill a tile);
  * LayerNormalization runs per pixel and per group;
  * the reference applies the SAME layers ``radix`` times to the same input (ResNest.py:138-145), so the
    radix branches are identical tensors: the branch is computed once and the split attention becomes
    out = radix * y * softmax_c(dense2(...)) - bit-for-bit the same function and the same
    gradients (they flow ``radix`` times into the shared weights, which the factor reproduces);
  * concats_2 (3x3) adds the shortcut branch in its epilogue.
"""
from __future__ import annotations

import contextlib
import os
from typing import List

import torch
import torch.nn as nn

from . import ops
from .layers import (KERAS_LN_EPS, KERAS_LRELU_ALPHA, AveragePooling2D, BatchNormalization, Conv2D, LayerNormalization,
                     LeakyReLU, _Workspace)
from .ops import ACT_LRELU, ACT_NONE, BF16, roundup


# see Decoder._FOLD_BN: convtmp_1bn's scale is folded into convtmp_1's packed forward operand (convtmp_2bn stays unfolded: it is
# already fused with the pool that follows it)
# per-stage lazy weight gradients in the encoder's backward pass: off - the side stream already carries the decoder's (and the ViT
# blocks') weight gradients beside this encoder, and more forks made Arch B 1 % and cfg4 3 % slower
# bit s-1: the weight gradients of encoder stage s run on the side stream beside the next stage's backward-data chain (ops.lazy_wgrads).
# Default 12 = the 16x16 and 32x32 stages (their launches are latency-bound and overlap well: 3.010 -> 2.984 ms); stage 2 or 1 on the side
# stream costs more than it hides (3.03 / 3.06 ms), all four 3.12 ms - the side stream still carries the decoder's weight gradients then.
# (with the ViT bottleneck - TBI_TransUNet.py, 512x512 - the side stream carries the ViT's weight gradients instead and 12 is 2.5 % slower than 0:
# the wrapper sets ``ResNest.stage_lazy`` to 0 there.)  USSEG_ENC_LAZY overrides both.
_STAGE_LAZY_ENV = os.environ.get("USSEG_ENC_LAZY")
_STAGE_LAZY = int(_STAGE_LAZY_ENV) if _STAGE_LAZY_ENV is not None else 12
_FOLD_BN = os.environ.get("USSEG_FOLD_BN", "1") != "0"
# the cardinal group + shortcut of a stage as ONE launch (csrc/cardinal.hip); 0 = the six unfused launches (the cross-check of the tests)
_FUSED_CARDINAL = os.environ.get("USSEG_FUSED_CARDINAL", "1") != "0"
_FUSED_CARDINAL_BWD = os.environ.get("USSEG_FUSED_CARDINAL_BWD", "1") != "0"     # ... and its backward chain + the shortcut norm's backward as one launch
# The fused backward launch holds one 8x8 tile per workgroup with ~200-500 registers per lane (one or two workgroups per CU): it wins where a
# stage has few tiles per CU (16x16 and 32x32 at batch 16: 49 vs 64 and 58 vs 71 us) and loses to the streaming norm kernels on the large stages
# (125 vs 66 us at 64x64, 201 vs 92 us at 128x128: 16 rounds of 10-us tiles).
_CARD_BWD_MAX_PX = int(os.environ.get("USSEG_CARD_BWD_MAX_PX", "32768"))
_LN_PAIR = os.environ.get("USSEG_LN_PAIR", "1") != "0"      # shortcut norm backward + conv2_bn backward of the large stages in one launch
_FUSED_STEM_BWD = os.environ.get("USSEG_FUSED_STEM_BWD", "1") != "0"     # stem backward-data passes fused with the norm / activation backward in front
_STEM_WGRAD_MERGE = os.environ.get("USSEG_STEM_WGRAD_MERGE", "1") != "0"   # the stem's three weight gradients in one multi-job launch
_FUSED_STEM = os.environ.get("USSEG_FUSED_STEM", "1") != "0"             # the stem (three convs, two norms, pool) as one launch (csrc/stem.hip)
_MERGED_DGRAD = os.environ.get("USSEG_MERGED_DGRAD", "1") != "0"     # one backward-data GEMM for a stage's grouped 1x1 and shortcut 1x1


def _span(t: torch.Tensor, n: int) -> torch.Tensor:
    """1-D fp32 view of n floats starting at t's first element (reaches into the following flat-buffer entries)."""
    return torch.as_strided(t, (n,), (1,))


def _make_norm(kind: str, channels: int):
    """``*_bn`` layers: LayerNormalization in ResNest.py / Decoder.py, BatchNormalization in the older copy TBI_TransUNet.py."""
    assert kind in ("ln", "bn")
    return LayerNormalization(channels) if kind == "ln" else BatchNormalization(channels)


class split_attention(nn.Module):
    """ResNest.py:153-199.  Holds dense1 / dense1_bn / dense2; executed by the owning cardinal group."""

    def __init__(self, inchannel, radix, atrous=1, wDecay=None, norm="ln"):
        super().__init__()
        self.inchannel, self.radix, self.atrous, self.wDecay = inchannel, radix, atrous, wDecay
        self.dense1 = Conv2D(inchannel, inchannel // 2, 1)
        self.dense1_bn = _make_norm(norm, inchannel // 2)            # LayerNormalization (:164); BatchNormalization in TBI_TransUNet.py:503
        self.dense1_act = LeakyReLU()
        self.dense2 = Conv2D(inchannel // 2, inchannel, 1)
        for c in (self.dense1, self.dense2):
            c.on_finalize = lambda device: None   # 1x1 on [B,1,1,C]: runs inside the split-attention MLP kernel (fp32)

    def forward(self, inputs: List[torch.Tensor]):
        """Standalone call with a LIST of ``radix`` NHWC tensors (ResNest.py:171).  Uses the general R=len(inputs)
        kernel path; inside ``residual_S`` the identical-branch shortcut is used instead."""
        R = len(inputs)
        B, H, W, Cp, _ = ops.geom(inputs[0])
        Cg = self.inchannel
        dev = inputs[0].device
        ycat = ops.new_act(B, H, W, roundup(R * Cg, 8), dev, zero=True)
        # channel layout (r, c): not 8-aligned in general, so assemble through fp32 (standalone path only)
        stack = torch.cat([ops.to_f32(t, Cg) for t in inputs], dim=3)
        ycat[..., :R * Cg] = stack.to(BF16)
        out = ops.new_act(B, H, W, roundup(Cg, 8), dev)
        bn = isinstance(self.dense1_bn, BatchNormalization)
        d = ops.splitattn_desc(B, H * W, 1, R, Cg, Cg // 2, ycat.shape[3], out.shape[3], ycat.shape[3], out.shape[3], 1.0, 1 if bn else 0,
                               KERAS_LN_EPS, ACT_LRELU, KERAS_LRELU_ALPHA, R == 1)
        w2 = self.dense2.kernel.data.reshape(1, Cg // 2, Cg).repeat(R, 1, 1).contiguous()   # the SAME dense2 for every r (:188)
        b2 = self.dense2.bias.data.repeat(R).contiguous()
        stats = (self.dense1_bn.moving_mean_p, self.dense1_bn.moving_variance_p) if bn else (None, None)
        params = (self.dense1.kernel.data, self.dense1.bias.data, self.dense1_bn.gamma.data, self.dense1_bn.beta.data) + stats + (w2, b2)
        ops.splitattn_fwd(d, ycat, params, out)
        return out

    def __call__(self, inputs, *args, **kwargs):
        return self.forward(inputs)


class cardinal(nn.Module):
    """ResNest.py:110-150."""

    def __init__(self, ksize, outchannel, radix, kpaths, atrous=1, wDecay=None, in_channels=None, norm="ln"):
        super().__init__()
        self.outchannel, self.ksize, self.radix, self.kpaths, self.atrous, self.wDecay = outchannel, ksize, radix, kpaths, atrous, wDecay
        self.in_channels = in_channels if in_channels is not None else outchannel   # stage input = outchannel (= stage_out/2)
        self.cv11 = int(outchannel / radix / kpaths)    # ResNest.py:120
        self.cvkk = int(outchannel / kpaths)            # ResNest.py:121
        self.conv1 = Conv2D(self.in_channels, self.cv11, 1, atrous)
        self.conv1_bn = _make_norm(norm, self.cv11)
        self.conv1_act = LeakyReLU()
        self.conv2 = Conv2D(self.cv11, self.cvkk, ksize, atrous)
        self.conv2_bn = _make_norm(norm, self.cvkk)
        self.conv2_act = LeakyReLU()
        self.split = split_attention(self.cvkk, radix, atrous, wDecay, norm=norm)
        for c in (self.conv1, self.conv2):
            c.on_finalize = lambda device: None   # packed by the owning _CardinalGroup
        self._solo = None

    def forward(self, x):
        if self._solo is None:
            self._solo = _CardinalGroup([self], self.radix)
            self._solo.on_finalize(x.device)
        return self._solo.forward(x)

    def backward(self, dy, dx_residual=None):
        return self._solo.backward(dy, dx_residual)

    def __call__(self, x, *args, **kwargs):
        return self.forward(x)


class _CardinalGroup:
    """Executes P cardinal blocks that share their input as grouped kernels (see module docstring)."""

    def __init__(self, cards: List[cardinal], radix: int):
        self.cards, self.P, self.radix = cards, len(cards), radix
        c0 = cards[0]
        self.cin, self.cv11, self.cvkk, self.k, self.dil = c0.in_channels, c0.cv11, c0.cvkk, c0.ksize, c0.atrous
        self.hid = self.cvkk // 2
        self.U, self.V = self.P * self.cv11, self.P * self.cvkk
        self.Up, self.Vp = roundup(self.U, 8), roundup(self.V, 8)
        self.cin_p = roundup(self.cin, 8)
        # TBI_TransUNet.py variant: the three norms are inference BatchNormalizations -> per-channel affine (norm mode 1, one "group")
        self.bn = isinstance(c0.conv1_bn, BatchNormalization)
        self.nmode, self.ngroups = (1, 1) if self.bn else (0, self.P)
        self.st1 = self.st2 = self.sta = (None, None)

    # variables that must be back to back in the flat buffer (per-path vectors are read as one [P*...] vector)
    def adjacent_params(self):
        cs = self.cards
        groups = [[c.conv1.bias for c in cs], [c.conv1_bn.gamma for c in cs], [c.conv1_bn.beta for c in cs],
                  [c.conv2.bias for c in cs], [c.conv2_bn.gamma for c in cs], [c.conv2_bn.beta for c in cs],
                  [c.split.dense1.kernel for c in cs], [c.split.dense1.bias for c in cs],
                  [c.split.dense1_bn.gamma for c in cs], [c.split.dense1_bn.beta for c in cs],
                  [c.split.dense2.kernel for c in cs], [c.split.dense2.bias for c in cs]]
        return [(g, 0) for g in groups]

    def on_finalize(self, device):
        T = self.k * self.k
        self.w1_f = torch.zeros((roundup(self.Up, 16), self.cin_p), dtype=BF16, device=device)
        self.w1_d = torch.zeros((roundup(self.cin_p, 16), self.Up), dtype=BF16, device=device)
        self.w2_f = torch.zeros((roundup(self.Vp, 16), T * self.Up), dtype=BF16, device=device)
        self.w2_d = torch.zeros((roundup(self.Up, 16), T * self.Vp), dtype=BF16, device=device)
        c0 = self.cards[0]
        sp = lambda p, n: (_span(p.data, n), _span(p.grad, n))
        self.b1, self.db1 = sp(c0.conv1.bias, self.Up)
        self.g1, self.dg1 = sp(c0.conv1_bn.gamma, self.U)
        self.be1, self.dbe1 = sp(c0.conv1_bn.beta, self.U)
        self.b2, self.db2 = sp(c0.conv2.bias, self.Vp)
        self.g2, self.dg2 = sp(c0.conv2_bn.gamma, self.V)
        self.be2, self.dbe2 = sp(c0.conv2_bn.beta, self.V)
        s = c0.split
        names = (s.dense1.kernel, s.dense1.bias, s.dense1_bn.gamma, s.dense1_bn.beta, s.dense2.kernel, s.dense2.bias)
        self.mlp_p = tuple(p.data for p in names)
        self.mlp_g = tuple(p.grad for p in names)
        if self.bn:   # the per-path moving statistics as ONE contiguous vector per norm (the layers keep views of it)
            self.st1 = self._share_stats([c.conv1_bn for c in self.cards], self.Up, device)
            self.st2 = self._share_stats([c.conv2_bn for c in self.cards], self.Vp, device)
            self.sta = self._share_stats([c.split.dense1_bn for c in self.cards], roundup(self.P * self.hid, 8), device)
        # adjacency sanity: path p's variable must start right after path p-1's
        for a, b in zip(self.cards[:-1], self.cards[1:]):
            assert b.conv1.bias.data_ptr() == a.conv1.bias.data_ptr() + 4 * self.cv11, "cardinal params are not adjacent"
            assert b.split.dense2.kernel.data_ptr() == a.split.dense2.kernel.data_ptr() + 4 * self.hid * self.cvkk
        self.repack()

    @staticmethod
    def _share_stats(bns, width, device):
        mean = torch.zeros(width, dtype=torch.float32, device=device)
        var = torch.ones(width, dtype=torch.float32, device=device)
        o = 0
        for bn in bns:
            mean[o:o + bn.C] = bn.moving_mean.to(device)
            var[o:o + bn.C] = bn.moving_variance.to(device)
            bn._buffers["moving_mean_p"], bn._buffers["moving_variance_p"] = mean[o:o + bn.C], var[o:o + bn.C]
            o += bn.C
        return mean, var

    def pack_jobs(self):
        T = self.k * self.k
        jobs = []
        for p, c in enumerate(self.cards):
            k1, k2 = c.conv1.kernel.data, c.conv2.kernel.data
            # conv1 [1,1,cin,cv11]: fwd rows n = p*cv11+j, K = ci ; dgrad rows = ci, K = p*cv11+j
            jobs.append(ops.pack_job(k1, 0, 1, self.cv11, 1, self.cv11, self.cin, self.w1_f, self.cin_p, self.cin_p, p * self.cv11, 0))
            jobs.append(ops.pack_job(k1, 0, self.cv11, 1, 1, self.cin, self.cv11, self.w1_d, self.Up, self.Up, 0, p * self.cv11))
            # conv2 [k,k,cv11,cvkk] on the diagonal block (p,p)
            sT = self.cv11 * self.cvkk
            jobs.append(ops.pack_job(k2, sT, 1, self.cvkk, T, self.cvkk, self.cv11, self.w2_f, T * self.Up, self.Up, p * self.cvkk, p * self.cv11))
            jobs.append(ops.pack_job(k2, sT, self.cvkk, 1, T, self.cv11, self.cvkk, self.w2_d, T * self.Vp, self.Vp, p * self.cv11, p * self.cvkk))
        return jobs

    def repack(self):
        jobs = self.pack_jobs()
        ops.pack_weights_batched(ops.make_pack_table(jobs, self.w1_f.device), len(jobs))

    def _maps(self):
        """Destination maps of the two grouped weight gradients: one block per cardinal path (Keras [k,k,in,out] variables)."""
        key = self.cards[0].conv1.kernel.grad.data_ptr()
        if getattr(self, "_wmaps_key", None) != key:
            sT = self.cv11 * self.cvkk
            m1 = ops.wgrad_dst([(c.conv1.kernel.grad, 0, self.cv11, 1, 0, p * self.cv11, self.cin, self.cv11) for p, c in enumerate(self.cards)])
            m2 = ops.wgrad_dst([(c.conv2.kernel.grad, sT, self.cvkk, 1, p * self.cv11, p * self.cvkk, self.cv11, self.cvkk)
                                for p, c in enumerate(self.cards)])
            self._wmaps, self._wmaps_key = (m1, m2), key
        return self._wmaps

    def _sa_desc(self, B, HW):
        use_sigmoid = self.radix == 1   # ResNest.py:189-190
        return ops.splitattn_desc(B, HW, self.P, 1, self.cvkk, self.hid, self.Vp, self.Vp, self.Vp, self.Vp, float(self.radix), self.nmode,
                                  KERAS_LN_EPS, ACT_LRELU, KERAS_LRELU_ALPHA, use_sigmoid)

    def _mlp_params(self):
        return self.mlp_p[:4] + self.sta + self.mlp_p[4:]

    def fused_ok(self, sc_conv) -> bool:
        """True if csrc/cardinal.hip has a kernel for this stage (LayerNormalization variant, 3x3, no dilation, the channel
        configurations of ResNest.py with radix 3 / kpaths 3)."""
        ok = getattr(self, "_fused_ok", None)
        if ok is None:
            ok = self._fused_ok = (not self.bn and self.k == 3 and self.dil == 1 and sc_conv.k == 1 and sc_conv.cout_p == sc_conv.cout and
                                   ops.cardinal_supported(self.cin_p, self.P, self.cv11, self.cvkk, self.Up, self.Vp, sc_conv.cout))
        return ok and _FUSED_CARDINAL

    def forward_fused(self, x, sc_conv, sc_norm, out=None):
        """ResNest.py:136-147 for all paths + the shortcut :99-101 in one launch, then the split-attention MLP and re-weighting
        (:171-199).  Leaves exactly the state the unfused forward leaves (``_saved``, the shortcut layers' saved inputs), so the
        backward pass is the same code. -> (concats_1, sc)"""
        B, H, W, _, _ = ops.geom(x)
        a = KERAS_LRELU_ALPHA
        u_raw, u, v_raw, y, gap, sc_raw, sc = ops.cardinal_fwd(x, self.w1_f, self.b1, self.g1, self.be1, self.w2_f, self.b2, self.g2, self.be2,
                                                               sc_conv.wp_f, sc_conv.bias.data, sc_norm.gamma.data, sc_norm.beta.data,
                                                               self.P, self.cv11, self.cvkk, self.Up, self.Vp, sc_conv.cout, KERAS_LN_EPS, a)
        out = out if out is not None else ops.new_act(B, H, W, self.Vp, x.device)
        _, g, s, ws = ops.splitattn_fwd(self._sa_desc(B, H * W), y, self._mlp_params(), out, gap=gap)
        self._saved = (x, u_raw, u, v_raw, y, g, s, ws)
        sc_conv._x = x
        sc_norm._x, sc_norm._act = sc_raw, (ACT_LRELU, a)
        return out, sc

    def forward(self, x, out=None):
        B, H, W, C, _ = ops.geom(x)
        dev = x.device
        a = KERAS_LRELU_ALPHA
        u_raw = ops.conv2d_fwd(x, self.w1_f, self.b1, 1, 1, ops.new_act(B, H, W, self.Up, dev))                 # :139
        u = ops.norm_act_fwd(u_raw, self.U, self.g1, self.be1, torch.empty_like(u_raw), self.nmode, self.ngroups, KERAS_LN_EPS, ACT_LRELU, a,
                             *self.st1)                                                                                    # :140-141
        v_raw = ops.conv2d_fwd(u, self.w2_f, self.b2, self.k, self.dil, ops.new_act(B, H, W, self.Vp, dev))    # :142
        # :143-144; the same launch emits the partial rows of the global average pool the split attention starts with (:179)
        y, gap = ops.norm_act_fwd_gap(v_raw, self.V, self.g2, self.be2, torch.empty_like(v_raw), self.nmode, self.ngroups, KERAS_LN_EPS,
                                      ACT_LRELU, a, *self.st2)
        out = out if out is not None else ops.new_act(B, H, W, self.Vp, dev)
        d = self._sa_desc(B, H * W)
        _, g, s, ws = ops.splitattn_fwd(d, y, self._mlp_params(), out, gap=gap)                                   # :171-199
        self._saved = (x, u_raw, u, v_raw, y, g, s, ws)
        return out

    def backward_fused(self, dout, dsc, sc_conv, sc_norm, dcat):
        """The backward pass of ``forward_fused``'s launch as ONE launch (csrc/cardinal.hip, K3 backward): the re-weighting's backward +
        conv2_bn backward -> grouped 3x3 backward-data -> conv1_bn backward into ``dcat[..., :Up]`` and the shortcut norm's backward of ``dsc``
        into ``dcat[..., Up:]``; the nine per-channel gradient vectors accumulate.  The split-attention MLP's backward (a per-image
        reduction over all pixels) stays in front of it, the grouped 3x3's weight gradient reads the dv it leaves."""
        x, u_raw, u, v_raw, y, g, s, ws = self._saved
        B, H, W, _, _ = ops.geom(x)
        sa_s, sa_dg = ops.splitattn_bwd(self._sa_desc(B, H * W), y, dout, self._mlp_params(), self.mlp_g, g, s, ws, None)
        dv = torch.empty_like(v_raw)
        grads = (self.dg2, self.dbe2, self.db2, self.dg1, self.dbe1, self.db1, sc_norm.gamma.grad, sc_norm.beta.grad, sc_conv.bias.grad)
        ops.cardinal_bwd(dout, dsc, v_raw, u_raw, sc_norm._x, self.w2_d, self.g2, self.be2, self.g1, self.be1, sc_norm.gamma.data, sc_norm.beta.data,
                         sa_s, sa_dg, float(self.radix), dv, dcat, grads, self.cin_p, self.P, self.cv11, self.cvkk, self.Up, self.Vp, sc_conv.cout,
                         KERAS_LN_EPS, KERAS_LRELU_ALPHA)
        ops.wgrad_later(lambda: ops.conv2d_wgrad_mapped(u, dv, self.k, self.dil, self._maps()[1]), u, dv)

    def backward(self, dout, dx_residual=None, du_raw_out=None, shortcut=None):
        """``du_raw_out``: write the gradient w.r.t. the grouped 1x1 conv's output there (a channel slice of the stage's [du_raw | dsc_raw]
        buffer) and leave the 1x1 backward-data pass to the caller (one GEMM for the cardinal group AND the shortcut: residual_S.backward).
        ``shortcut`` = (norm layer, dsc, dx slice, conv bias grad): the shortcut norm's backward rides in the same launch as conv2_bn's where the
        pair has an instantiation (``ops.norm_act_bwd_pair``); otherwise it runs here as its own launch."""
        x, u_raw, u, v_raw, y, g, s, ws = self._saved
        B, H, W, _, _ = ops.geom(x)
        dev = x.device
        a = KERAS_LRELU_ALPHA
        T = self.k * self.k
        d = self._sa_desc(B, H * W)
        # the re-weighting's backward (dy = radix*s*dout + dg) is formed inside the norm backward: no dy tensor, no apply pass
        sa_s, sa_dg = ops.splitattn_bwd(d, y, dout, self._mlp_params(), self.mlp_g, g, s, ws, None)
        dv = torch.empty_like(v_raw)
        paired = False
        if shortcut is not None:
            scn, dsc, dx_sc, dbias_sc = shortcut
            if _LN_PAIR and not self.bn and scn._act == (ACT_LRELU, a):
                paired = ops.norm_act_bwd_pair(scn._x, dsc, scn.C, scn.gamma.data, scn.beta.data, dx_sc, scn.gamma.grad, scn.beta.grad, dbias_sc,
                                               v_raw, dout, self.V, self.ngroups, self.g2, self.be2, dv, self.dg2, self.dbe2, self.db2, sa_s, sa_dg,
                                               float(self.radix), KERAS_LN_EPS, a)
            if not paired:
                scn.backward(dsc, dx=dx_sc, dbias=dbias_sc)
        if not paired:
            ops.norm_act_bwd_sa(v_raw, dout, self.V, self.g2, self.be2, dv, self.dg2, self.dbe2, self.nmode, self.ngroups,
                                KERAS_LN_EPS, ACT_LRELU, a, sa_s, sa_dg, float(self.radix), *self.st2, dbias=self.db2)
        # grouped 3x3: dense wgrad into scratch, keep the diagonal blocks
        # grouped 3x3: the dense [T][Up][Vp] gradient is never materialised - only the diagonal blocks are scattered
        ops.wgrad_later(lambda: ops.conv2d_wgrad_mapped(u, dv, self.k, self.dil, self._maps()[1]), u, dv)
        du = ops.conv2d_dgrad(dv, self.w2_d, self.k, self.dil, torch.empty_like(u))
        du_raw = ops.norm_act_bwd(u_raw, du, self.U, self.g1, self.be1, du_raw_out if du_raw_out is not None else torch.empty_like(u_raw),
                                  self.dg1, self.dbe1, self.nmode, self.ngroups, KERAS_LN_EPS, ACT_LRELU, a, *self.st1, dbias=self.db1)
        if du_raw_out is not None:      # the caller runs ONE weight gradient and ONE backward-data pass for [grouped 1x1 | shortcut 1x1]
            return None
        ops.wgrad_later(lambda: ops.conv2d_wgrad_mapped(x, du_raw, 1, 1, self._maps()[0]), x, du_raw)
        return ops.conv2d_dgrad(du_raw, self.w1_d, 1, 1, ops.new_act(B, H, W, self.cin_p, dev), dx_residual)


class residual_S(nn.Module):
    """ResNest.py:61-107."""

    def __init__(self, ksize, outchannel, radix, kpaths, atrous=1, wDecay=None, in_channels=None, norm="ln"):
        super().__init__()
        self.kpaths, self.atrous, self.ksize, self.outchannel, self.radix, self.wDecay = kpaths, atrous, ksize, outchannel, radix, wDecay
        self.in_channels = in_channels if in_channels is not None else outchannel // 2
        self.cardinal_blocks = nn.ModuleList(
            [cardinal(ksize, outchannel // 2, radix, kpaths, atrous, wDecay, in_channels=self.in_channels, norm=norm) for _ in range(kpaths)])
        cvkk = self.cardinal_blocks[0].cvkk
        self.concats_2 = Conv2D(kpaths * cvkk, outchannel, ksize, atrous)
        self.convtmp_sc = Conv2D(self.in_channels, outchannel, 1, atrous)
        self.convtmp_scbn = _make_norm(norm, outchannel)
        self.convtmp_scact = LeakyReLU()
        self._group = _CardinalGroup(list(self.cardinal_blocks), radix)

    def adjacent_params(self):
        return self._group.adjacent_params()

    def on_finalize(self, device):
        g, sc = self._group, self.convtmp_sc
        # backward-data operand of [grouped 1x1 | shortcut 1x1] as ONE GEMM: rows = input channel, K = [Up | Oc] (ResNest.py:99,139 read the same x)
        self.wcat_d = (torch.zeros((roundup(g.cin_p, 16), g.Up + sc.cout_p), dtype=BF16, device=device)
                       if _MERGED_DGRAD and sc.k == 1 and g.P <= 3 else None)        # (a mapped weight gradient scatters into at most four variables)
        g.on_finalize(device)
        self.repack()          # (the group packs its own operands; this adds the merged one - a freshly built model must not step on zeros)

    def repack(self):
        jobs = self.pack_jobs()
        ops.pack_weights_batched(ops.make_pack_table(jobs, self._group.w1_f.device), len(jobs))

    def pack_jobs(self):
        g, sc = self._group, self.convtmp_sc
        jobs = g.pack_jobs()
        if self.wcat_d is not None:
            Kc = g.Up + sc.cout_p
            for p, c in enumerate(g.cards):
                jobs.append(ops.pack_job(c.conv1.kernel.data, 0, g.cv11, 1, 1, g.cin, g.cv11, self.wcat_d, Kc, Kc, 0, p * g.cv11))
            sT, sI, sO = sc._strides_tio()
            jobs.append(ops.pack_job(sc.kernel.data, sT, sI, sO, 1, sc.cin, sc.cout, self.wcat_d, Kc, Kc, 0, g.Up))
        return jobs

    def forward(self, x, out=None):
        if self._group.fused_ok(self.convtmp_sc):
            concats_1, sc = self._group.forward_fused(x, self.convtmp_sc, self.convtmp_scbn)        # :91-96 and :99-101, one launch + the attention
            return self.concats_2.forward(concats_1, out=out, residual=sc)                           # :98,:102
        concats_1 = self._group.forward(x)                                                          # :91-96
        sc_raw = self.convtmp_sc.forward(x)                                                          # :99
        sc = self.convtmp_scbn.forward(sc_raw, ACT_LRELU, KERAS_LRELU_ALPHA)                        # :100-101
        return self.concats_2.forward(concats_1, out=out, residual=sc)                               # :98,:102

    def backward(self, dout, need_dx=True, bias_done=False):
        """``bias_done``: concats_2's bias gradient (the column sums of dout) was already produced by the kernel that made dout."""
        d_c1 = self.concats_2.backward(dout, skip_bias=bias_done)
        if self.wcat_d is not None and need_dx:
            # the gradients w.r.t. both 1x1 outputs land in one buffer and ONE backward-data GEMM with K = Up + Oc gives dx (:99 + :139 read the same x)
            g, sc = self._group, self.convtmp_sc
            x = sc._x
            B, H, W, _, _ = ops.geom(x)
            dcat = ops.new_act(B, H, W, g.Up + sc.cout_p, x.device)
            if _FUSED_CARDINAL_BWD and g.fused_ok(sc) and B * H * W < _CARD_BWD_MAX_PX:
                g.backward_fused(d_c1, dout, sc, self.convtmp_scbn, dcat)
            else:
                g.backward(d_c1, du_raw_out=dcat[..., :g.Up], shortcut=(self.convtmp_scbn, dout, dcat[..., g.Up:], sc.bias.grad))
            # x^T . [du_raw | dsc_raw]: the weight gradients of the paths' 1x1 convs and of the shortcut conv in one launch (four mapped blocks)
            ops.wgrad_later(lambda: ops.conv2d_wgrad_mapped(x, dcat, 1, 1, self._wcat_map()), x, dcat)
            return ops.conv2d_dgrad(dcat, self.wcat_d, 1, 1, ops.new_act(B, H, W, g.cin_p, x.device))
        dsc_raw = self.convtmp_scbn.backward(dout, dbias=self.convtmp_sc.bias.grad)
        dx_a = self._group.backward(d_c1)
        return self.convtmp_sc.backward(dsc_raw, need_dx=need_dx, dx_residual=dx_a, skip_bias=True)

    def _wcat_map(self):
        """Destination map of the merged 1x1 weight gradient [Cin][Up + Oc]: one block per cardinal path + the shortcut conv (Keras [1,1,in,out])."""
        g, sc = self._group, self.convtmp_sc
        key = sc.kernel.grad.data_ptr()
        if getattr(self, "_wcat_key", None) != key:
            blocks = [(c.conv1.kernel.grad, 0, g.cv11, 1, 0, p * g.cv11, g.cin, g.cv11) for p, c in enumerate(g.cards)]
            blocks.append((sc.kernel.grad, 0, sc.cout, 1, 0, g.Up, sc.cin, sc.cout))
            self._wcat_m, self._wcat_key = ops.wgrad_dst(blocks), key
        return self._wcat_m

    def __call__(self, x, *args, **kwargs):
        return self.forward(x)


class ResNest(nn.Module):
    """ResNest.py:4-58.  ``forward(x)`` returns ``(x_4, [x_3, x_2, x_1])`` (NHWC, bf16)."""

    def __init__(self, height, width, channel, ksize, radix=4, kpaths=4, wDecay=None, *, norm="ln", widths=(64, 128, 256, 512)):
        """``norm`` / ``widths``: ("ln", (64,128,256,512)) is ResNest.py; ("bn", (64,128,256,256)) the copy inside TBI_TransUNet.py
        (BatchNormalization at :426,465,472,503 and a 256-channel conv_4 at :368)."""
        super().__init__()
        self.height, self.width, self.channel = height, width, channel
        self.ksize, self.radix, self.kpaths, self.wDecay = ksize, radix, kpaths, wDecay
        self.conv1 = Conv2D(channel, 16, 3)
        self.conv1_act = LeakyReLU()
        self.convtmp_1 = Conv2D(16, 32, 3)
        self.convtmp_1bn = BatchNormalization(32)
        self.convtmp_1act = LeakyReLU()
        self.convtmp_2 = Conv2D(32, 32, 3)
        self.convtmp_2bn = BatchNormalization(32)
        self.convtmp_2act = LeakyReLU()
        self.conv1_pool = AveragePooling2D()
        self.conv2_pool = AveragePooling2D()
        self.conv3_pool = AveragePooling2D()
        self.conv4_pool = AveragePooling2D()
        w = tuple(widths)
        self.widths = w
        self.conv_1 = residual_S(ksize, w[0], radix, kpaths, wDecay=wDecay, in_channels=32, norm=norm)
        self.conv_2 = residual_S(ksize, w[1], radix, kpaths, wDecay=wDecay, in_channels=w[0], norm=norm)
        self.conv_3 = residual_S(ksize, w[2], radix, kpaths, wDecay=wDecay, in_channels=w[1], norm=norm)
        self.conv_4 = residual_S(ksize, w[3], radix, kpaths, wDecay=wDecay, in_channels=w[2], norm=norm)

    def forward(self, x, outs=None):
        """x: NHWC float32/float64 [B,H,W,channel] (cast to bf16 here, ResNest.py:39) or an already-cast bf16 tensor.
        ``outs``: optional destinations [x_3, x_2, x_1] (channel-slice views allowed) the stage outputs are written into."""
        o3, o2, o1 = outs if outs is not None else (None, None, None)
        if x.dtype != BF16:
            x = ops.cast_input(x.contiguous(), roundup(self.channel, 8))
        a = KERAS_LRELU_ALPHA
        bn1, bn2 = self.convtmp_1bn, self.convtmp_2bn
        self._fold = self.convtmp_1.fold_scale() is not None
        if (_FUSED_STEM and self._fold and not bn2.training_mode and self.conv1.cin_p == 8 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
                and (self.conv1.cout, self.convtmp_1.cout, self.convtmp_2.cout) == (16, 32, 32)):
            # :39-47 as ONE launch (csrc/stem.hip); leaves the state the four launches below leave, so the backward pass is the same code
            self._y1, self._t1, c2, t = ops.stem_fwd(x, self.conv1.wp_f, self.conv1.bias.data, self.convtmp_1.wp_f, bn1.fold_shift,
                                                     self.convtmp_2.wp_f, self.convtmp_2.bias.data, bn2.gamma.data, bn2.beta.data,
                                                     bn2.moving_mean_p, bn2.moving_variance_p, bn2.eps, a)
            self.conv1._x, self.convtmp_1._x, self.convtmp_2._x = x, self._y1, self._t1
            bn2._x, bn2._act = c2, (ACT_LRELU, a)
            self._pool_fused = self._stem_fused = True
            return self._stages_forward(t, o1, o2, o3)
        self._stem_fused = False
        self._y1 = self.conv1.forward(x, act=ACT_LRELU, alpha=a)                                  # :39-40
        self._pool_fused = False
        if self._fold:   # :41-43: convtmp_1bn's scale sits in the packed operand, its shift is the bias, LeakyReLU in the epilogue
            self._t1 = t = self.convtmp_1.forward(self._y1, act=ACT_LRELU, alpha=a, bias=bn1.fold_shift)
        else:
            t = self.convtmp_1.forward(self._y1)                                                 # :41
            t = self.convtmp_1bn.forward(t, ACT_LRELU, a)                                        # :42-43
        t = self.convtmp_2.forward(t)                                                            # :44
        self._pool_fused = not bn2.training_mode
        if self._pool_fused:    # :45-47 in one pass: the activated 256x256 tensor feeds the pool only and is never written
            t = self.convtmp_2bn.forward_pool(t, ACT_LRELU, a)
        else:
            t = self.conv1_pool.forward(self.convtmp_2bn.forward(t, ACT_LRELU, a))               # :45-47
        return self._stages_forward(t, o1, o2, o3)

    def _stages_forward(self, t, o1, o2, o3):
        x_1 = self.conv_1.forward(t, out=o1)                                                     # :48
        x_2 = self.conv_2.forward(self.conv2_pool.forward(x_1), out=o2)                          # :49-50
        x_3 = self.conv_3.forward(self.conv3_pool.forward(x_2), out=o3)                          # :51-52
        x_4 = self.conv_4.forward(self.conv4_pool.forward(x_3))                                  # :53-54
        return x_4, [x_3, x_2, x_1]                                                              # :55

    def backward(self, d_x4, d_feats):
        """Gradients w.r.t. (x_4, [x_3, x_2, x_1]) -> accumulates all parameter gradients (input gradient not needed)."""
        d_x3, d_x2, d_x1 = d_feats
        # the pool backward in front of a stage also sums its output over the pixels: that stage's concats_2 bias gradient
        # each stage's weight gradients run on the side stream beside the next stage's backward-data chain (ops.lazy_wgrads)
        mask = getattr(self, "stage_lazy", _STAGE_LAZY)
        lazy = lambda st: ops.lazy_wgrads() if (mask >> (st - 1)) & 1 else contextlib.nullcontext()
        with lazy(4):
            d = self.conv_4.backward(d_x4)
        d = self.conv4_pool.backward(d, add=d_x3, db=self.conv_3.concats_2.bias.grad)
        with lazy(3):
            d = self.conv_3.backward(d, bias_done=True)
        d = self.conv3_pool.backward(d, add=d_x2, db=self.conv_2.concats_2.bias.grad)
        with lazy(2):
            d = self.conv_2.backward(d, bias_done=True)
        d = self.conv2_pool.backward(d, add=d_x1, db=self.conv_1.concats_2.bias.grad)
        with lazy(1):
            d = self.conv_1.backward(d, bias_done=True)
        a = KERAS_LRELU_ALPHA
        if self._pool_fused:
            d = self.convtmp_2bn.backward_pool(d, dbias=self.convtmp_2.bias.grad)
        else:
            d = self.convtmp_2bn.backward(self.conv1_pool.backward(d), dbias=self.convtmp_2.bias.grad)
        bn1, c1, c2 = self.convtmp_1bn, self.convtmp_1, self.convtmp_2
        # the three stem weight gradients (32x32, 16x32, 8x16 channels at full resolution: one 32x32 tile each) share ONE launch at the end of
        # the stem's backward pass instead of three full-chip launches between its backward-data kernels (USSEG_STEM_WGRAD_MERGE=0: as before)
        wjobs = [] if _STEM_WGRAD_MERGE else None
        def wg(conv, dy):
            if wjobs is None:
                conv.backward(dy, need_dx=False, skip_bias=True)
            else:
                wjobs.append(conv.wgrad_job(dy))
        # :44 backwards + :41-43 backwards as ONE launch (csrc/stem.hip dgrad_actbwd_kernel): convtmp_2's backward-data pass, then the folded
        # BatchNorm + LeakyReLU backward from the stored activation; the intermediate gradient never goes to HBM
        dn = torch.empty_like(self._t1) if (_FUSED_STEM_BWD and self._fold) else None
        if dn is not None and ops.conv3_dgrad_actbwd(d, c2.wp_d, self._t1, dn, 2, a, c1.bias.grad, bn1.gamma.data, bn1.beta.data, bn1.moving_variance_p,
                                                     bn1.eps, bn1.gamma.grad, bn1.beta.grad):
            wg(c2, d)                                              # its weight gradient only
            d = dn
        else:
            if wjobs is not None:
                wjobs.append(c2.wgrad_job(d))
            d = c2.backward(d, skip_bias=True, skip_wgrad=wjobs is not None)
            if self._fold:
                d = bn1.backward_folded(self._t1, d, ACT_LRELU, a, dbias=c1.bias.grad)
            else:
                d = bn1.backward(d, dbias=c1.bias.grad)
        # :41 backwards + :40 backwards likewise: convtmp_1's backward-data pass + LeakyReLU' (sign(y) == sign(pre)) + the column sums of the
        # result = conv1's bias gradient
        dn = torch.empty_like(self._y1) if _FUSED_STEM_BWD else None
        if dn is not None and ops.conv3_dgrad_actbwd(d, c1.wp_d, self._y1, dn, 0, KERAS_LRELU_ALPHA, self.conv1.bias.grad):
            wg(c1, d)
            d = dn
        else:
            if wjobs is not None:
                wjobs.append(c1.wgrad_job(d))
            d = c1.backward(d, skip_bias=True, skip_wgrad=wjobs is not None)
            d = ops.act_bwd_colsum(self._y1, d, torch.empty_like(d), ACT_LRELU, KERAS_LRELU_ALPHA, self.conv1.bias.grad, self.conv1.cout)
        wg(self.conv1, d)
        if wjobs:
            keep = [t for job in wjobs for t in job[:2]]
            ops.wgrad_later(lambda: ops.conv2d_wgrad_multi(wjobs), *keep)
        return None

    def bn_fold_jobs(self):
        return [self.convtmp_1bn.fold_job(self.convtmp_1.bias.data), self.convtmp_2bn.fold_job(self.convtmp_2.bias.data)]

    def on_finalize(self, device):
        if getattr(self, "_fold_table", None) is None:
            jobs = self.bn_fold_jobs()
            self._fold_table = (ops.make_bn_fold_table(jobs, device), len(jobs))
        ops.bn_fold_batched(*self._fold_table)
        if _FOLD_BN:    # convtmp_1 packs its forward operand with convtmp_1bn's scale (its own on_finalize / repack_all run after this)
            self.convtmp_1._fold_scale = self.convtmp_1bn.fold_scale
            self.convtmp_1._fold_bns = [self.convtmp_1bn]

    def repack(self):
        for m in self.modules():
            if m is not self and isinstance(m, (Conv2D, residual_S)) and getattr(m, "wp_f", 1) is not None:
                m.repack()
        self.on_finalize(self.conv1.kernel.device)

    def __call__(self, x, *args, **kwargs):
        return self.forward(x)
