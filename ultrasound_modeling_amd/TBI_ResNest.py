"""Arch A: the self-contained ResNeSt/UNet of the reference's ``TBI_ResNest.py`` (model :80-220, loss :234-248, step :35-55).

Surface kept: ``ResNest(height, width, channel, num_class, ksize, radix=4, kpaths=4, learning_rate=1e-3, ckpt_dir)``
with ``.resModel`` and ``.step(x, y, train) -> (loss_map [H,W], accuracy, probs)``.  Variables carry the Keras layer
names of the reference (``Conv1``, ``conv2_1_car_k01_r0``, ``conv2_1_car_k0_att2_r1``, ``upsample_3_t_conv``, ``f_tran`` ...).

Differences from Arch B that matter for the kernels: ELU instead of LeakyReLU, BatchNormalization (inference mode as
driven) instead of LayerNormalization, SEPARATE weights per radix branch (so the split attention
really mixes ``radix`` different tensors), five stages with the shortcut conv only where channel counts differ
(:142), a decoder of 4x4 stride-2 transposed convs + BN + ReLU with ``tf.nn.dropout(0.5)`` that is ALWAYS active on
the first three (:215-216), and the class-balanced loss ``my_loss_cat`` whose value is an [H,W] map.

A stage's kpaths*radix 1x1 convs are one GEMM and its kpaths*radix 3x3 convs one block-diagonal implicit GEMM, as in
Arch B; stages whose branch channels exceed 512 (kpaths*radix*cvkk = 768) are processed in slabs of paths.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .flat import AdamClip, FlatParams
from .layers import (KERAS_BN_EPS, KERAS_ELU_ALPHA, AveragePooling2D, BatchNormalization, Conv2D, Conv2DTranspose, QuadHead, QuadTConv,
                     _Workspace)
from .ops import ACT_ELU, ACT_NONE, ACT_RELU, BF16, roundup
from .step import TrainStepDriver

_STEM_WGRAD_MERGE = os.environ.get("USSEG_STEM_WGRAD_MERGE", "1") != "0"   # the stem's three weight gradients in one multi-job launch (as ResNest.py's)
_STAGE_LAZY = os.environ.get("USSEG_STAGE_LAZY", "1") != "0"    # per-stage lazy weight gradients (ops.lazy_wgrads) in the backward pass: -1.5 %


def _span(t: torch.Tensor, n: int) -> torch.Tensor:
    return torch.as_strided(t, (n,), (1,))


def _managed(conv: Conv2D) -> Conv2D:
    conv.on_finalize = lambda device: None      # packed / executed by its slab
    return conv


class _SlabA:
    """The radix x len(paths) branches of some cardinal paths of one stage, executed as grouped kernels
    (TBI_ResNest.py:153-207)."""

    def __init__(self, model, name, paths, radix, cin, cv11, cvkk, ksize):
        self.paths, self.R, self.P = list(paths), radix, len(paths)
        self.cin, self.cv11, self.cvkk, self.hid, self.k = cin, cv11, cvkk, cvkk // 2, ksize
        self.G = self.P * self.R
        self.U, self.V = self.G * cv11, self.G * cvkk
        self.Up, self.Vp = roundup(self.U, 8), roundup(self.V, 8)
        self.cin_p = roundup(cin, 8)
        self.Co = self.P * cvkk
        g = lambda n: getattr(model, n)
        self.br = []        # per branch (p, r): conv1, bn1, conv2, bn2
        self.att = []       # per path: att1, att_bn, [att2_r]
        for p in self.paths:
            cp = f"{name}_car_k{p}"
            for r in range(radix):
                self.br.append((g(f"{cp}1_r{r}"), g(f"{cp}1_r{r}bn"), g(f"{cp}2_r{r}"), g(f"{cp}2_r{r}bn")))
            self.att.append((g(f"{cp}_att1"), g(f"{cp}_att_bn"), [g(f"{cp}_att2_r{r}") for r in range(radix)]))

    def adjacent_params(self):
        b, a = self.br, self.att
        groups = [[x[0].bias for x in b], [x[1].gamma for x in b], [x[1].beta for x in b],
                  [x[2].bias for x in b], [x[3].gamma for x in b], [x[3].beta for x in b],
                  [x[0].kernel for x in a], [x[0].bias for x in a], [x[1].gamma for x in a], [x[1].beta for x in a],
                  [c.kernel for x in a for c in x[2]], [c.bias for x in a for c in x[2]]]
        return [(g, 0) for g in groups]

    def _share_stats(self, bns, width, device):
        """One contiguous (mean, var) buffer for a list of BatchNormalization layers; the layers keep views of it."""
        mean = torch.zeros(width, dtype=torch.float32, device=device)
        var = torch.ones(width, dtype=torch.float32, device=device)
        o = 0
        for bn in bns:
            mean[o:o + bn.C] = bn.moving_mean.to(device)
            var[o:o + bn.C] = bn.moving_variance.to(device)
            bn._buffers["moving_mean_p"], bn._buffers["moving_variance_p"] = mean[o:o + bn.C], var[o:o + bn.C]
            o += bn.C
        return mean, var

    def on_finalize(self, device):
        T = self.k * self.k
        z = lambda r, c: torch.zeros((r, c), dtype=BF16, device=device)
        self.w1_f, self.w1_d = z(roundup(self.Up, 16), self.cin_p), z(roundup(self.cin_p, 16), self.Up)
        self.w2_f, self.w2_d = z(roundup(self.Vp, 16), T * self.Up), z(roundup(self.Up, 16), T * self.Vp)
        b0, a0 = self.br[0], self.att[0]
        sp = lambda p, n: (_span(p.data, n), _span(p.grad, n))
        self.b1, self.db1 = sp(b0[0].bias, self.Up)
        self.g1, self.dg1 = sp(b0[1].gamma, self.U)
        self.be1, self.dbe1 = sp(b0[1].beta, self.U)
        self.b2, self.db2 = sp(b0[2].bias, self.Vp)
        self.g2, self.dg2 = sp(b0[3].gamma, self.V)
        self.be2, self.dbe2 = sp(b0[3].beta, self.V)
        self.m1, self.v1 = self._share_stats([x[1] for x in self.br], self.Up, device)
        self.m2, self.v2 = self._share_stats([x[3] for x in self.br], self.Vp, device)
        self.ma, self.va = self._share_stats([x[1] for x in self.att], self.P * self.hid, device)
        names = (a0[0].kernel, a0[0].bias, a0[1].gamma, a0[1].beta, a0[2][0].kernel, a0[2][0].bias)
        self.mlp_p = (names[0].data, names[1].data, names[2].data, names[3].data, self.ma, self.va, names[4].data, names[5].data)
        self.mlp_g = tuple(p.grad for p in names)
        for x, y in zip(self.br[:-1], self.br[1:]):
            assert y[0].bias.data_ptr() == x[0].bias.data_ptr() + 4 * self.cv11, "branch params are not adjacent"
        assert self.att[0][2][1].kernel.data_ptr() == self.att[0][2][0].kernel.data_ptr() + 4 * self.hid * self.cvkk or self.R == 1

    def pack_jobs(self):
        T = self.k * self.k
        jobs = []
        for gidx, (c1, _, c2, _) in enumerate(self.br):
            k1, k2 = c1.kernel.data, c2.kernel.data
            jobs.append(ops.pack_job(k1, 0, 1, self.cv11, 1, self.cv11, self.cin, self.w1_f, self.cin_p, self.cin_p, gidx * self.cv11, 0))
            jobs.append(ops.pack_job(k1, 0, self.cv11, 1, 1, self.cin, self.cv11, self.w1_d, self.Up, self.Up, 0, gidx * self.cv11))
            sT = self.cv11 * self.cvkk
            jobs.append(ops.pack_job(k2, sT, 1, self.cvkk, T, self.cvkk, self.cv11, self.w2_f, T * self.Up, self.Up, gidx * self.cvkk,
                                     gidx * self.cv11))
            jobs.append(ops.pack_job(k2, sT, self.cvkk, 1, T, self.cv11, self.cvkk, self.w2_d, T * self.Vp, self.Vp, gidx * self.cv11,
                                     gidx * self.cvkk))
        return jobs

    def _sa_desc(self, B, HW, out):
        return ops.splitattn_desc(B, HW, self.P, self.R, self.cvkk, self.hid, self.Vp, ops.geom(out)[4], self.Vp, self.Co, 1.0, 1,
                                  KERAS_BN_EPS, ACT_ELU, KERAS_ELU_ALPHA, self.R == 1)

    def forward(self, x, out):
        """out: the [.., P*cvkk] channel slice of concats_1 this slab owns."""
        B, H, W, _, _ = ops.geom(x)
        dev = x.device
        e = KERAS_BN_EPS
        u_raw = ops.conv2d_fwd(x, self.w1_f, self.b1, 1, 1, ops.new_act(B, H, W, self.Up, dev))                      # :162
        u = ops.norm_act_fwd(u_raw, self.U, self.g1, self.be1, torch.empty_like(u_raw), 1, 1, e, ACT_ELU, 1.0, self.m1, self.v1)   # :164-165
        v_raw = ops.conv2d_fwd(u, self.w2_f, self.b2, self.k, 1, ops.new_act(B, H, W, self.Vp, dev))               # :167
        y, gap = ops.norm_act_fwd_gap(v_raw, self.V, self.g2, self.be2, torch.empty_like(v_raw), 1, 1, e, ACT_ELU, 1.0, self.m2, self.v2)   # :169-170 (+ pooled partial rows)
        d = self._sa_desc(B, H * W, out)
        _, g, s, ws = ops.splitattn_fwd(d, y, self.mlp_p, out, gap=gap)                                               # :175-207
        self._saved = (x, u_raw, u, v_raw, y, g, s, ws)
        return out

    def _unpack_tables(self, dev):
        """Private scratch gradients of the two grouped convs + the device job tables that scatter their diagonal blocks."""
        key = self.br[0][0].kernel.grad.data_ptr()
        if getattr(self, "_unp_key", None) != key:
            T = self.k * self.k
            n1, n2 = self.cin_p * self.Up, T * self.Up * self.Vp
            pool = getattr(self, "_scratch_pool", None)        # (buffer, offset): slices of the model's ONE scratch (zeroed by one launch)
            if pool is not None:
                s1, s2 = pool[0][pool[1]:pool[1] + n1], pool[0][pool[1] + n1:pool[1] + n1 + n2]
            else:
                s1 = torch.zeros(n1, dtype=torch.float32, device=dev)
                s2 = torch.zeros(n2, dtype=torch.float32, device=dev)
            sT = self.cv11 * self.cvkk
            j1 = [ops.unpack_job(s1, self.cin_p, self.Up, 1, self.cv11, self.cin, gi * self.cv11, 0, c1.kernel.grad, 0, 1, self.cv11)
                  for gi, (c1, _, c2, _) in enumerate(self.br)]
            j2 = [ops.unpack_job(s2, self.Up, self.Vp, T, self.cvkk, self.cv11, gi * self.cvkk, gi * self.cv11, c2.kernel.grad, sT, 1, self.cvkk)
                  for gi, (c1, _, c2, _) in enumerate(self.br)]
            self._unp = (s1, s2, ops.make_unpack_table(j1, dev), ops.make_unpack_table(j2, dev))
            self._unp_jobs = j1 + j2
            self._unp_key = key
        return self._unp

    def backward(self, dout, dx_residual=None):
        x, u_raw, u, v_raw, y, g, s, ws = self._saved
        B, H, W, _, _ = ops.geom(x)
        dev = x.device
        e, T = KERAS_BN_EPS, self.k * self.k
        d = self._sa_desc(B, H * W, dout)
        dy = ops.splitattn_bwd(d, y, dout, self.mlp_p, self.mlp_g, g, s, ws, torch.empty_like(y))
        dv = ops.norm_act_bwd(v_raw, dy, self.V, self.g2, self.be2, torch.empty_like(v_raw), self.dg2, self.dbe2, 1, 1, e, ACT_ELU, 1.0,
                              self.m2, self.v2, dbias=self.db2)
        s1, s2, tab1, tab2 = self._unpack_tables(dev)
        # 12 (path, radix) blocks > the 4 of a mapped destination: the dense gradient goes through this slab's private scratch and
        # its diagonal blocks are scattered (one launch) once the deferred split-K finishes have run at the end of the backward pass
        def params2():
            if getattr(self, "_scratch_pool", None) is None:
                ops.fill_f32(s2, 0.0)
            ops.conv2d_wgrad(u, dv, self.k, 1, s2)
            if not getattr(self, "_unpack_merged", False):   # (the model scatters every slab's blocks in ONE launch after the final flush)
                ops.after_flush(lambda: ops.unpack_wgrad_batched(tab2))
        ops.wgrad_later(params2, u, dv)
        du = ops.conv2d_dgrad(dv, self.w2_d, self.k, 1, torch.empty_like(u))
        du_raw = ops.norm_act_bwd(u_raw, du, self.U, self.g1, self.be1, torch.empty_like(u_raw), self.dg1, self.dbe1, 1, 1, e, ACT_ELU, 1.0,
                                  self.m1, self.v1, dbias=self.db1)
        def params1():
            if getattr(self, "_scratch_pool", None) is None:
                ops.fill_f32(s1, 0.0)
            ops.conv2d_wgrad(x, du_raw, 1, 1, s1)
            if not getattr(self, "_unpack_merged", False):
                ops.after_flush(lambda: ops.unpack_wgrad_batched(tab1))
        ops.wgrad_later(params1, x, du_raw)
        return ops.conv2d_dgrad(du_raw, self.w1_d, 1, 1, ops.new_act(B, H, W, self.cin_p, dev), dx_residual)


class _StageA:
    """residual_S of Arch A (TBI_ResNest.py:130-151)."""

    def __init__(self, model, name, cin, oc, radix, kpaths, ksize):
        self.name, self.cin, self.oc = name, cin, oc
        half = oc // 2
        self.cv11, self.cvkk = int(half / radix / kpaths), int(half / kpaths)       # :157-158
        per_path = radix * self.cvkk
        npaths = max(1, min(kpaths, 512 // per_path))
        while kpaths % npaths:
            npaths -= 1
        self.slabs = [_SlabA(model, name, range(p0, p0 + npaths), radix, cin, self.cv11, self.cvkk, ksize) for p0 in range(0, kpaths, npaths)]
        self.concats_2 = getattr(model, name + "_concats_2")
        self.has_sc = cin != oc                                                    # :142
        if self.has_sc:
            self.cc, self.scbn = getattr(model, name + "_cc"), getattr(model, name + "_scbn")
        self.Vtot = kpaths * self.cvkk

    def forward(self, x):
        B, H, W, _, _ = ops.geom(x)
        c1 = ops.new_act(B, H, W, roundup(self.Vtot, 8), x.device, zero=(self.Vtot % 8 != 0))
        for sl in self.slabs:
            o = sl.paths[0] * self.cvkk
            sl.forward(x, c1[..., o:o + sl.Co])                                     # :132-139
        if self.has_sc:
            sc = self.scbn.forward(self.cc.forward(x), ACT_ELU, KERAS_ELU_ALPHA)    # :143-145
        else:
            sc = x
        return self.concats_2.forward(c1, residual=sc)                              # :140,:148

    def backward(self, dout):
        d_c1 = self.concats_2.backward(dout)
        if self.has_sc:
            dsc = self.scbn.backward(dout, dbias=self.cc.bias.grad)
            dx = self.cc.backward(dsc, skip_bias=True)
        else:
            dx = dout
        for sl in self.slabs:
            o = sl.paths[0] * self.cvkk
            dx = sl.backward(d_c1[..., o:o + sl.Co], dx_residual=dx)
        return dx


class _ResModel(nn.Module):
    """The functional Keras model of TBI_ResNest.py:80-128 as explicit forward / backward."""

    STAGES = (("conv2_1", 64), ("conv2_2", 128), ("conv3_1", 256), ("conv3_2", 512), ("conv4_1", 512))
    UPS = (("upsample_0", 512, True), ("upsample_1", 512, True), ("upsample_2", 512, True), ("upsample_3", 256, False),
           ("upsample_4", 128, False))

    def __init__(self, height, width, channel, num_class, ksize, radix, kpaths):
        super().__init__()
        self.height, self.width, self.channel, self.num_class = height, width, channel, num_class
        self.radix, self.kpaths, self.ksize = radix, kpaths, ksize
        G = dict(init="glorot")                                        # Keras default initialiser everywhere (App. A.5)
        add = self.add_module
        add("Conv1", Conv2D(channel, 16, 3, **G))                      # :83
        add("conv2_1_1", Conv2D(16, 32, 3, **G))                       # :85
        add("conv2_1_2", Conv2D(32, 32, 3, **G))                       # :88
        add("conv2_1_2bn", BatchNormalization(32))                     # :90
        cin = 32
        for name, oc in self.STAGES:
            half = oc // 2
            cv11, cvkk = int(half / radix / kpaths), int(half / kpaths)
            for k in range(kpaths):
                cp = f"{name}_car_k{k}"
                for r in range(radix):
                    add(f"{cp}1_r{r}", _managed(Conv2D(cin, cv11, 1, **G)))          # :162
                    add(f"{cp}1_r{r}bn", BatchNormalization(cv11))                   # :164
                    add(f"{cp}2_r{r}", _managed(Conv2D(cv11, cvkk, ksize, **G)))     # :167
                    add(f"{cp}2_r{r}bn", BatchNormalization(cvkk))                   # :169
                add(f"{cp}_att1", _managed(Conv2D(cvkk, cvkk // 2, 1, **G)))         # :189
                add(f"{cp}_att_bn", BatchNormalization(cvkk // 2))                   # :190
                for r in range(radix):
                    add(f"{cp}_att2_r{r}", _managed(Conv2D(cvkk // 2, cvkk, 1, **G)))   # :195
            add(name + "_concats_2", Conv2D(kpaths * cvkk, oc, ksize, **G))          # :140
            if cin != oc:
                add(name + "_cc", Conv2D(cin, oc, 1, **G))                           # :143
                add(name + "_scbn", BatchNormalization(oc))                          # :144
            cin = oc
        skips = (512, 256, 128, 64, 32)                                              # pool5, pool4, pool3, pool2, pool1 channels
        cin = 512
        for (name, oc, _), sk in zip(self.UPS, skips):
            add(name + "_t_conv", Conv2DTranspose(cin, oc, 4, **G))                  # :210
            add(name + "_bn", BatchNormalization(oc))                                # :213
            cin = oc + sk
        add("f_tran", Conv2DTranspose(cin, num_class, 4, **G))                       # :124
        self._quad = QuadHead(self.f_tran) if (num_class <= 4 and os.environ.get("USSEG_QUAD_HEAD", "1") != "0") else None
        # the five 4x4 stride-2 up-convolutions (:210) as tap-masked convs on space-to-depth tensors
        self._qt = {}
        if os.environ.get("USSEG_QUAD_TCONV", "0") != "0":      # built and parity-tested; with 768-1024 input channels both paths are MFMA-heavy and the quad form measured 5 % slower (11.6 vs 11.0 ms), so off by default
            for name, oc, drop in self.UPS:
                object.__setattr__(self, "_qt_" + name, QuadTConv(getattr(self, name + "_t_conv")))
                self._qt[name] = getattr(self, "_qt_" + name)
        self._stages = None
        self.dropout_seed = 0           # model-level seed of the always-on dropout masks
        self._eval_salt = 0             # 0 inside training steps; the wrapper numbers evaluation calls
        self.injected_masks = None      # tests inject {0,1} keep masks to make the always-on dropout deterministic

    # grouped-kernel plumbing -----------------------------------------------------------------------------------
    def _build(self):
        if self._stages is None:
            cin, st = 32, []
            for name, oc in self.STAGES:
                st.append(_StageA(self, name, cin, oc, self.radix, self.kpaths, self.ksize))
                cin = oc
            object.__setattr__(self, "_stages", st)
        return self._stages

    def adjacent_params(self):
        return [g for st in self._build() for sl in st.slabs for g in sl.adjacent_params()]

    def on_finalize(self, device):
        for st in self._build():
            for sl in st.slabs:
                sl.on_finalize(device)
        if self._quad is not None:
            self._quad.on_finalize(device)
        for q in self._qt.values():
            q.on_finalize(device)

    def pack_jobs(self):
        jobs = [j for st in self._build() for sl in st.slabs for j in sl.pack_jobs()]
        for m in self.modules():
            if isinstance(m, Conv2D) and m.wp_f is not None:
                jobs += m.pack_jobs()
        if self._quad is not None:
            jobs += self._quad.pack_jobs()
        for q in self._qt.values():
            jobs += q.pack_jobs()
        return jobs

    def repack(self):
        table = getattr(self, "_pack_table", None)
        if table is None:
            jobs = self.pack_jobs()
            dev = next(self.parameters()).device
            table = (ops.make_pack_table(jobs, dev), len(jobs), ops.make_pack_tilemap(jobs, dev))
            object.__setattr__(self, "_pack_table", table)
        ops.pack_weights_batched(*table)

    # forward / backward -------------------------------------------------------------------------------------------
    def _mask(self, i, like):
        if self.injected_masks is not None:
            m = self.injected_masks[i]
            return None if m is None else m
        # tf.nn.dropout(out, 0.5), :216.  The mask is a pure function of (model seed, layer, the optimiser's DEVICE step counter,
        # evaluation-call number): the eager step t and the HIP-graph replay of step t draw the same mask, successive steps and
        # successive evaluation calls draw fresh ones
        mask = torch.empty_like(like)
        return ops.dropout_mask(mask, (self.dropout_seed * 1000003 + self._eval_salt) * 7919 + i, 0.5, getattr(self, "step_dev", None))

    def forward(self, x, return_logits=False):
        if x.dtype != BF16:
            x = ops.cast_input(x.contiguous(), roundup(self.channel, 8))
        g = lambda n: getattr(self, n)
        a = KERAS_ELU_ALPHA
        self._r1 = g("Conv1").forward(x)                                               # :83
        t = ops.act_fwd(self._r1, torch.empty_like(self._r1), ACT_ELU, a)              # :84
        self._r2 = g("conv2_1_1").forward(t)                                           # :85
        t = ops.act_fwd(self._r2, torch.empty_like(self._r2), ACT_ELU, a)              # :87
        self._pools = [AveragePooling2D() for _ in range(6)]
        # tf.concat([up_i, pool_(5-i)]) of the decoder (:110-122) costs nothing: pool_1..pool_5 are written straight into the skip slices
        # of the five concat buffers (16-byte aligned channel offsets; every consumer takes a channel stride)
        B0, H0, W0 = x.shape[0], x.shape[1], x.shape[2]
        skips = (512, 256, 128, 64, 32)
        cats = [ops.new_act(B0, H0 >> (5 - i), W0 >> (5 - i), oc + sk, x.device) for i, ((_, oc, _), sk) in enumerate(zip(self.UPS, skips))]
        slot = lambda j: cats[4 - j][..., self.UPS[4 - j][1]:]          # where pooled[j] (pool_(j+1)) lives
        bn = g("conv2_1_2bn")
        self._pool_fused = not bn.training_mode
        if self._pool_fused:   # :88-92 BN + ELU + pool_1 in one pass: the activated 256x256 tensor feeds the pool only
            pooled = [bn.forward_pool(g("conv2_1_2").forward(t), ACT_ELU, a, out=slot(0))]
        else:
            t = bn.forward(g("conv2_1_2").forward(t), ACT_ELU, a)                      # :88-91
            pooled = [self._pools[0].forward(t, out=slot(0))]                          # pool_1 (:92)
        for i, st in enumerate(self._build()):                                         # :93-107
            pooled.append(self._pools[i + 1].forward(st.forward(pooled[-1]), out=slot(i + 1) if i + 1 < 5 else None))
        # pooled = [pool1(32ch), pool2(64), pool3(128), pool4(256), pool5(512), pool6(512)]
        u = pooled[5]
        self._cats, self._masks, self._upraw = [], [], []
        for i, (name, oc, drop) in enumerate(self.UPS):                                # :109-122
            skip = pooled[4 - i]
            B, H, W, _, _ = ops.geom(u)
            raw = self._qt[name].forward(u) if name in self._qt else g(name + "_t_conv").forward(u)   # :210
            cat = cats[i]
            assert cat.shape[1] == 2 * H and cat.shape[3] == oc + skip.shape[3] and skip.data_ptr() == cat[..., oc:].data_ptr()
            mask = self._mask(i, raw) if drop else None
            bn = g(name + "_bn")
            ops.norm_act_fwd(raw, oc, bn.gamma.data, bn.beta.data, cat[..., :oc], 1, 1, KERAS_BN_EPS, ACT_RELU, 0.0, bn.moving_mean_p,
                             bn.moving_variance_p, mask=mask)                          # :213-218
            self._cats.append(cat); self._masks.append(mask); self._upraw.append(raw)
            u = cat
        B, H, W = u.shape[0], 2 * u.shape[1], 2 * u.shape[2]
        self.out_hw = (H, W)
        # :124; logits fp32 [B,H,W,4] or, in quad form, [B,H/2,W/2,16] (ops.softmax_loss indexes it through quad_w)
        logits = self._quad.forward(u) if self._quad is not None else g("f_tran").forward(u, out_f32=True)
        if return_logits:
            return logits
        probs = torch.empty((B, H, W, self.num_class), dtype=torch.float32, device=logits.device)
        ops.softmax_loss(logits, None, probs, None, None, HW=H * W, C_classes=self.num_class, quad_w=self.quad_w)    # :125
        return probs

    @property
    def quad_w(self):
        return self.out_hw[1] if self._quad is not None else 0

    def backward(self, dlogits):
        dpool = [None] * 6                    # gradients w.r.t. pool1..pool6 outputs coming from the decoder skips
        with ops.lazy_wgrads():               # the up-convolutions' weight gradients run on the side stream beside the stages' backward pass
            d = self._decoder_backward(dlogits, dpool)
        return self._stages_backward(d, dpool)

    def _decoder_backward(self, dlogits, dpool):
        g = lambda n: getattr(self, n)
        d = self._quad.backward(dlogits) if self._quad is not None else g("f_tran").backward(dlogits)
        for i in reversed(range(5)):
            name, oc, drop = self.UPS[i]
            dpool[4 - i] = d[..., oc:]        # skip branch of the concat
            bn = g(name + "_bn")
            raw = self._upraw[i]
            draw = ops.norm_act_bwd(raw, d[..., :oc], oc, bn.gamma.data, bn.beta.data, torch.empty_like(raw), bn.gamma.grad, bn.beta.grad,
                                    1, 1, KERAS_BN_EPS, ACT_RELU, 0.0, bn.moving_mean_p, bn.moving_variance_p,
                                    dbias=g(name + "_t_conv").bias.grad, mask=self._masks[i])
            d = self._qt[name].backward(draw, bias_grad=False) if name in self._qt else self._tconv_backward(g(name + "_t_conv"), draw)
        return d                              # w.r.t. pool6 (the input of upsample_0); pool5..pool1 also feed the decoder concats (dpool)

    def _stages_backward(self, d, dpool):
        g = lambda n: getattr(self, n)
        stages = self._build()
        # the grouped convs' dense weight gradients go through private scratch (zero before each backward pass): one buffer, one fill
        if getattr(self, "_wscratch", None) is None or self._wscratch.device != d.device:
            sizes = [(sl, sl.cin_p * sl.Up + sl.k * sl.k * sl.Up * sl.Vp) for st in stages for sl in st.slabs]
            buf = torch.zeros(sum(n for _, n in sizes), dtype=torch.float32, device=d.device)
            off = 0
            for sl, n in sizes:
                sl._scratch_pool, sl._unp_key = (buf, off), None
                off += n
            object.__setattr__(self, "_wscratch", buf)
        ops.fill_f32(self._wscratch, 0.0)
        slabs = [sl for st in stages for sl in st.slabs]
        key = tuple(sl.br[0][0].kernel.grad.data_ptr() for sl in slabs)
        if getattr(self, "_unp_all_key", None) != key:       # one job table for the diagonal-block scatter of every slab
            jobs = []
            for sl in slabs:
                sl._unpack_tables(d.device)
                sl._unpack_merged = True
                jobs += sl._unp_jobs
            object.__setattr__(self, "_unp_all", ops.make_unpack_table(jobs, d.device))
            object.__setattr__(self, "_unp_all_key", key)
        ops.after_flush(lambda: ops.unpack_wgrad_batched(self._unp_all))
        for i in reversed(range(5)):
            with ops.lazy_wgrads() if _STAGE_LAZY else contextlib.nullcontext():   # this stage's weight gradients run beside the next stage
                d = stages[i].backward(self._pools[i + 1].backward(d))   # through pool_{i+2} and stage i -> w.r.t. pool_{i+1}
            d = self._add(d, dpool[i])                               # + the skip branch of the decoder concat
        a = KERAS_ELU_ALPHA
        if self._pool_fused:
            d = g("conv2_1_2bn").backward_pool(d, dbias=g("conv2_1_2").bias.grad)
        else:
            d = g("conv2_1_2bn").backward(self._pools[0].backward(d), dbias=g("conv2_1_2").bias.grad)
        # the three stem weight gradients (32x32, 16x32, 8x16 channels: one 32x32 tile each) share one launch at the end (as in ResNest.py's stem)
        wjobs = [g("conv2_1_2").wgrad_job(d)]
        d = g("conv2_1_2").backward(d, skip_bias=True, skip_wgrad=True)
        # ELU' on the stored pre-activations; the same pass sums its output over the pixels = the conv's bias gradient
        d = ops.act_bwd_colsum(self._r2, d, torch.empty_like(d), ACT_ELU, a, g("conv2_1_1").bias.grad, g("conv2_1_1").cout)
        wjobs.append(g("conv2_1_1").wgrad_job(d))
        d = g("conv2_1_1").backward(d, skip_bias=True, skip_wgrad=True)
        d = ops.act_bwd_colsum(self._r1, d, torch.empty_like(d), ACT_ELU, a, g("Conv1").bias.grad, g("Conv1").cout)
        wjobs.append(g("Conv1").wgrad_job(d))
        keep = [t for job in wjobs for t in job[:2]]
        if _STEM_WGRAD_MERGE:
            ops.wgrad_later(lambda: ops.conv2d_wgrad_multi(wjobs), *keep)
        else:
            ops.wgrad_later(lambda: [ops.conv2d_wgrad_multi([j]) for j in wjobs], *keep)

    @staticmethod
    def _add(a, b):
        out = a.clone() if not a.is_contiguous() else a
        ops.copy_channels(b, out, accumulate=True)
        return out

    @staticmethod
    def _tconv_backward(layer, dy):
        # Conv2DTranspose.backward computes the bias gradient itself; here it came from the BN backward (dbias)
        x = layer._x
        # straight into the Keras [k,k,Cout,Cin] variable, slabs instead of atomics
        ops.wgrad_later(lambda: ops.tconv2d_wgrad_mapped(x, dy, layer.k, layer._wgrad_map()), x, dy)
        B, H, W, _, _ = ops.geom(x)
        return ops.tconv2d_dgrad(dy, layer.wp_d, layer.k, ops.new_act(B, H, W, layer.cin_p, dy.device))


class ResNest(TrainStepDriver):
    """TBI_ResNest.py:15-55.  ``step(x, y, train)`` -> (loss map [H,W], accuracy, probabilities)."""

    def __init__(self, height, width, channel, num_class, ksize, radix=4, kpaths=4, learning_rate=1e-3, ckpt_dir="./Checkpoint", *,
                 device: Optional[str] = None, seed: Optional[int] = 0):
        if seed is not None:
            torch.manual_seed(seed)
        assert height % 64 == 0 and width % 64 == 0, "six 2x2 poolings: H and W must be multiples of 64"
        self.height, self.width, self.channel, self.num_class = height, width, channel, num_class
        self.ksize, self.learning_rate, self.radix, self.kpaths, self.ckpt_dir = ksize, learning_rate, radix, kpaths, ckpt_dir
        dev = device or ("cuda" if torch.cuda.is_available() else None)
        if dev is None:
            raise RuntimeError("ResNest needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device(dev)
        self.resModel = _ResModel(height, width, channel, num_class, ksize, radix, kpaths)       # :26
        self.flat = FlatParams(self.resModel, self.device)
        self.resModel.repack()
        self.optimizer = AdamClip(self.flat, lr=learning_rate, clip_norm=None)                  # :28 (no clipping, :46)
        self.class_factor = [0.06329, 0.027567, 0.90914]                                        # :30 (unused by my_loss_cat)
        self._loss_map = torch.zeros(height * width, dtype=torch.float32, device=self.device)
        self._scale = torch.zeros(height * width * num_class, dtype=torch.float32, device=self.device)
        self.grad_sync = None
        self._graph = None
        self.resModel.step_dev = self.optimizer.step_dev      # dropout masks follow the device step counter (graph replays)
        object.__setattr__(self.resModel, "save", self.save_params)     # the driver calls neuralnet.resModel.save(path) (TBI_ResNest.py:472)

    # parameters in / out under the Keras layer names ------------------------------------------------------------------
    def load_params(self, params: dict):
        own = dict(self.resModel.named_parameters())
        bufs = {}
        for name, mod in self.resModel.named_modules():
            if isinstance(mod, BatchNormalization):
                bufs[name + ".moving_mean"], bufs[name + ".moving_variance"] = mod.moving_mean, mod.moving_variance
        missing = [k for k in own if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]} ...")
        for k, v in params.items():
            if k in own:
                assert tuple(own[k].shape) == tuple(v.shape), f"{k}: {tuple(own[k].shape)} vs {tuple(v.shape)}"
                own[k].data.copy_(v.to(torch.float32))
            elif k in bufs:
                bufs[k].copy_(v.to(torch.float32))
            else:
                raise KeyError(f"unexpected parameter {k}")
        self.resModel.repack()

    def export_params(self) -> dict:
        out = {k: v.data.detach().clone() for k, v in self.resModel.named_parameters()}
        for name, mod in self.resModel.named_modules():
            if isinstance(mod, BatchNormalization):
                out[name + ".moving_mean"], out[name + ".moving_variance"] = mod.moving_mean.clone(), mod.moving_variance.clone()
        return out

    def export_grads(self) -> dict:
        return {k: v.grad.detach().clone() for k, v in self.resModel.named_parameters()}

    def save_params(self, path=None):
        """TBI_ResNest.py:57-66 (a tf.train.CheckpointManager there, broken as committed): variables, BN statistics and the
        optimiser state in one file."""
        path = path or os.path.join(self.ckpt_dir, "usseg_arch_a.pt")
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        d = self.export_params()
        d.update(self.optimizer.state_dict())
        torch.save(d, path)
        return path

    def load_params_file(self, path=None):
        """TBI_ResNest.py:68-78 ``load_params``: restore what ``save_params`` wrote."""
        d = torch.load(path or os.path.join(self.ckpt_dir, "usseg_arch_a.pt"), map_location="cpu", weights_only=True)
        self.optimizer.load_state_dict({k: d.pop(k) for k in list(d) if k.startswith("__adam_")})
        self.load_params(d)

    def my_loss_cat(self, y_true, y_pred):
        """TBI_ResNest.py:234-248 on PROBABILITIES: per class c, scale[h,w] = 1/(sum_b y[b,h,w,c] + 1)/(H*W);
        -sum_{b,c} y*log(p+1e-7)*scale -> the [H,W] loss map.  (step() uses the kernel fused with the head softmax.)"""
        y_true, y_pred = self._prep_y(y_true), self._prep_y(y_pred)
        B, H, W, Cc = y_pred.shape
        scale = torch.empty(H * W * Cc, dtype=torch.float32, device=self.device)
        loss = torch.empty(H * W, dtype=torch.float32, device=self.device)
        ops.loss_cat_scale(y_true, scale)
        ops.loss_from_probs(y_pred, y_true, loss, HW=H * W, C_classes=Cc, loss_kind=1, scale=scale)
        return loss.reshape(H, W)

    # ---- hooks of the shared step order (step.TrainStepDriver); the step is also capturable as HIP graph(s)
    def _prep_x(self, x):
        return torch.as_tensor(x).to(self.device).contiguous()

    def _prep_y(self, y):
        return torch.as_tensor(y).to(device=self.device, dtype=torch.float32).contiguous()

    def _zero_grad(self):
        self.flat.zero_grad()

    def _forward_loss(self, x, y, with_grad: bool):
        net = self.resModel
        if with_grad:
            net._eval_salt = 0
        else:
            self._eval_calls = getattr(self, "_eval_calls", 0) + 1
            net._eval_salt = self._eval_calls
        logits = net.forward(x, return_logits=True)                                            # :40
        B, (H, W), qw = logits.shape[0], net.out_hw, net.quad_w
        probs = torch.empty((B, H, W, self.num_class), dtype=torch.float32, device=self.device)
        dlogits = None
        if with_grad:
            dlogits = ops.new_act(B, H // 2, W // 2, 16, self.device) if qw else ops.new_act(B, H, W, 8, self.device)
        ops.loss_cat_scale(y, self._scale)                                                     # :240-241 (the loss map is overwritten)
        ops.softmax_loss(logits, y, probs, self._loss_map, dlogits, HW=H * W, C_classes=self.num_class, loss_kind=1,
                         scale=self._scale, quad_w=qw)                                         # :125, :234-248
        return probs, dlogits

    def _forward_backward(self, x, y):
        probs, dlogits = self._forward_loss(x, y, with_grad=True)
        with ops.overlap_region():              # deferred / batched finishing reductions of the norm backward and bias sums
            self.resModel.backward(dlogits)                                                    # :43 (gradient of the SUM of the map)
        return probs

    def _repack(self):
        self.resModel.repack()

    def _graph_state(self):
        bn = [b for m in self.resModel.modules() if isinstance(m, BatchNormalization) for b in (m.moving_mean_p, m.moving_variance_p)]
        return [self.flat.flat] + self.optimizer.state_tensors() + bn

    def step(self, x, y, train=False):
        """TBI_ResNest.py:35-55 -> (loss map [H,W], accuracy, probabilities)."""
        x, y = self._prep_x(x), self._prep_y(y)
        if not train:
            probs, _ = self._forward_loss(x, y, with_grad=False)
        elif self._graph is not None:
            probs = self._graph_replay(x, y)
        else:
            probs = self._train_body(x, y)                                                     # :43-46 (no clipping: clip_norm None)
        if getattr(self, "_acc", None) is None or self._acc.device != probs.device:
            self._acc = torch.zeros(ops.ACC_FLOATS, dtype=torch.float32, device=probs.device)
        accuracy = ops.accuracy(probs, y, self._acc)                                           # :48-51 (metric only): one pass, one count
        if train and self._graph is not None:   # replayed step: loss map / accuracy / probabilities are static buffers, valid until the next call
            return self._loss_map.reshape(self.height, self.width), accuracy, probs     # (two clones = two more dispatches behind every replay)
        return self._loss_map.clone().reshape(self.height, self.width), accuracy.clone(), probs

    def modules(self):
        return self.resModel.modules()

    def repack(self):
        self.resModel.repack()

    def eval_step(self, x, y):
        """MirroredTrainer-compatible evaluation: -> (sum of the loss map, probs)."""
        loss_map, _, probs = self.step(x, y, train=False)
        return self._loss_total(loss_map), probs

    def train_step(self, x, y):
        """MirroredTrainer-compatible alias: -> (sum of the loss map, probs)."""
        loss_map, _, probs = self.step(x, y, train=True)
        return self._loss_total(loss_map), probs

    def _loss_total(self, loss_map):
        """The scalar the mirrored step reduces (MainParallel.py:131-134): the sum of the [H,W] loss map, by the library's ordered sum."""
        if getattr(self, "_lsum", None) is None or self._lsum.device != loss_map.device:
            self._lsum = torch.zeros(ops.ACC_FLOATS, dtype=torch.float32, device=loss_map.device)
        return ops.sum_f32(loss_map.contiguous().reshape(-1), self._lsum)
