"""UNet-style decoder: the module surface of the reference's ``Decoder.py`` (DecoderBlock, DecoderCup).

Same class names, constructor arguments, attribute names and call signatures as ``Decoder.py:7-146`` of
silverlight6/Ultrasound_Modeling, on hand-written gfx950 kernels.  Concatenations never copy more than they
must: every producer writes straight into its channel slice of the concat buffer (``tf.concat(axis=3)`` at
Decoder.py:66,75,87,141); only tensors that already exist elsewhere (skip features, the re-injected hidden
state) are copied in.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .layers import (KERAS_BN_EPS, KERAS_LRELU_ALPHA, AveragePooling2D, BatchNormalization, Conv2D, Conv2DTranspose, LayerNormalization,
                     LeakyReLU, QuadHead, QuadTConv)
from .ops import ACT_LRELU, ACT_NONE, BF16, roundup


_FUSED_DGRAD = os.environ.get("USSEG_FUSED_DGRAD", "1") != "0"
_BRANCH4 = os.environ.get("USSEG_BRANCH4", "1") != "0"     # the 1x1 branch of a DecoderBlock stage as a fourth job of its multi-job conv launch
_QUAD_UP = os.environ.get("USSEG_QUAD_UP", "0") != "0"
# Inference-mode BatchNorm folded into the producing conv: its scale goes into the PACKED forward operand (W' = W*gamma*rstd per output
# channel), its shift replaces the bias, LeakyReLU rides in the plain epilogue - no norm launch and no pre-norm tensor in the forward
# pass; the backward pass recovers what it needs from the activated output (norm backward mode 2).
_FOLD_BN = os.environ.get("USSEG_FOLD_BN", "1") != "0"


class DecoderBlock(nn.Module):
    """Decoder.py:7-94: tconv3x3s2 -> concat skip -> 4 parallel convs (1x1, 3x3 d2/d4/d8)+BN -> LeakyReLU, twice."""

    def __init__(self, out_channels, wDecay=None, in_channels=None, skip_channels=None):
        super().__init__()
        self.wDecay = wDecay
        oc = out_channels
        self.out_channels = oc
        self.in_channels = in_channels if in_channels is not None else oc
        self.skip_channels = oc if skip_channels is None else skip_channels
        c1 = oc + self.skip_channels
        dil = (1, 2, 4, 8)
        for j in range(4):                                                            # :11-25
            setattr(self, f"conv1_{j}", Conv2D(c1, oc // 4, 1 if j == 0 else 3, dil[j]))
        self.pool = AveragePooling2D()                                                # dead in the reference (:26)
        self.LeakyReLU1 = LeakyReLU()
        for j in range(4):                                                            # :32-35
            setattr(self, f"bn1_{j}", BatchNormalization(oc // 4))
        for j in range(4):                                                            # :36-50
            setattr(self, f"conv2_{j}", Conv2D(oc, oc // 4, 1 if j == 0 else 3, dil[j]))
        for j in range(4):                                                            # :51-54
            setattr(self, f"bn2_{j}", BatchNormalization(oc // 4))
        self.up = Conv2DTranspose(self.in_channels, oc, 3)                            # :57
        self._qup = QuadTConv(self.up) if _QUAD_UP else None

    # ---- the four parallel branches of a stage share one BatchNormalization launch: their gamma / beta / conv-bias
    # variables are laid out back to back in the flat buffer and their moving statistics share one device buffer
    def adjacent_params(self):
        groups = []
        for st in ("1", "2"):
            groups.append(([getattr(self, f"bn{st}_{j}").gamma for j in range(4)], 0))
            groups.append(([getattr(self, f"bn{st}_{j}").beta for j in range(4)], 0))
            groups.append(([getattr(self, f"conv{st}_{j}").bias for j in range(4)], 0))
        return groups

    def on_finalize(self, device):
        oc, q = self.out_channels, self.out_channels // 4
        assert q % 8 == 0
        self._bn = {}
        for st in ("1", "2"):
            bns = [getattr(self, f"bn{st}_{j}") for j in range(4)]
            mean = torch.cat([b.moving_mean_p.to(device) for b in bns])
            var = torch.cat([b.moving_variance_p.to(device) for b in bns])
            for j, b in enumerate(bns):
                b._buffers["moving_mean_p"] = mean[j * q:(j + 1) * q]
                b._buffers["moving_variance_p"] = var[j * q:(j + 1) * q]
            span = lambda t: torch.as_strided(t, (oc,), (1,))
            c0 = getattr(self, f"conv{st}_0")
            self._bn[st] = dict(mean=mean, var=var, gamma=span(bns[0].gamma.data), beta=span(bns[0].beta.data),
                                dgamma=span(bns[0].gamma.grad), dbeta=span(bns[0].beta.grad), dbias=span(c0.bias.grad),
                                bias=span(c0.bias.data), fscale=torch.ones(oc, device=device), fshift=torch.zeros(oc, device=device))
            for a_, b_ in zip(bns[:-1], bns[1:]):
                assert b_.gamma.data_ptr() == a_.gamma.data_ptr() + 4 * q and b_.beta.data_ptr() == a_.beta.data_ptr() + 4 * q
        # backward-data operand of a stage's four branches, concatenated along K (one implicit GEMM, dx written once)
        self._wd_cat = {}
        for st in ("1", "2"):
            cin_p = getattr(self, f"conv{st}_0").cin_p
            self._wd_cat[st] = torch.zeros((roundup(cin_p, 16), 28 * q), dtype=BF16, device=device)
        if self._qup is not None:
            self._qup.on_finalize(device)
        jobs = self.bn_fold_jobs()
        ops.bn_fold_batched(ops.make_bn_fold_table(jobs, device), len(jobs))
        if _FOLD_BN:     # the branch convs pack their forward operands with the folded scale (their own on_finalize / repack_all run after this)
            for st in ("1", "2"):
                for j in range(4):
                    cv = getattr(self, f"conv{st}_{j}")
                    cv._fold_scale = self._bn[st]["fscale"][j * q:(j + 1) * q]
                    cv._fold_bns = [getattr(self, f"bn{s2}_{jj}") for s2 in ("1", "2") for jj in range(4)]    # the block folds as a whole
        ops.pack_weights_batched(ops.make_pack_table(self.pack_jobs(), device), len(self.pack_jobs()))

    def bn_fold_jobs(self):
        """One folded inference BatchNorm per stage (its four branch BNs are adjacent vectors): scale/shift for the conv epilogues."""
        return [ops.bn_fold_job(b["gamma"], b["beta"], b["mean"], b["var"], b["bias"], b["fscale"], b["fshift"], self.out_channels, KERAS_BN_EPS)
                for b in (self._bn["1"], self._bn["2"])]

    def pack_jobs(self):
        """Pack jobs of the concatenated backward-data operands (the per-conv operands are packed by the Conv2D layers)."""
        q = self.out_channels // 4
        jobs = []
        for st in ("1", "2"):
            base = 0
            for j in range(4):
                c = getattr(self, f"conv{st}_{j}")
                T = c.k * c.k
                sT, sI, sO = c._strides_tio()
                jobs.append(ops.pack_job(c.kernel.data, sT, sI, sO, T, c.cin, c.cout, self._wd_cat[st], 28 * q, q, 0, base))
                base += T * q
        if self._qup is not None:
            jobs += self._qup.pack_jobs()
        return jobs

    def _bn_fwd(self, st, raw, out):
        d = self._bn[st]
        return ops.norm_act_fwd(raw, self.out_channels, d["gamma"], d["beta"], out, 1, 1, KERAS_BN_EPS, ACT_LRELU, KERAS_LRELU_ALPHA,
                                d["mean"], d["var"])

    def _bn_bwd(self, st, raw, dy, dx):
        """``raw``: the pre-norm conv outputs, or (folded) the ACTIVATED outputs - mode 2 recovers what it needs from them."""
        d = self._bn[st]
        return ops.norm_act_bwd(raw, dy, self.out_channels, d["gamma"], d["beta"], dx, d["dgamma"], d["dbeta"], 2 if self._fold else 1, 1,
                                KERAS_BN_EPS, ACT_LRELU, KERAS_LRELU_ALPHA, d["mean"], d["var"], dbias=d["dbias"])

    def _branches_fwd(self, st, x, raw):
        """The four parallel convs of a stage: the 1x1, then the three dilated 3x3 convs as ONE multi-job launch.
        Folded: inference-mode BN + LeakyReLU ride in the conv epilogues and ``raw`` receives the activated output."""
        q = self.out_channels // 4
        convs = [getattr(self, f"conv{st}_{j}") for j in range(4)]
        b = self._bn[st]
        sl = lambda t, j: t[j * q:(j + 1) * q]
        # the 1x1 branch rides in the dilated branches' launch as a fourth, centre-tap-only job (it reads the same x tile): one dispatch
        # less per stage on the forward chain.  USSEG_BRANCH4=0: its own launch, as before (the comparison of its test)
        four = _BRANCH4 and convs[0].k == 1 and convs[0].dil == 1
        convs[0]._x = x
        if four:
            first = (0,)
        elif self._fold:      # the scale is inside wp_f (pack time); the shift is the bias; LeakyReLU in the plain epilogue
            convs[0].forward(x, out=raw[..., :q], act=ACT_LRELU, alpha=KERAS_LRELU_ALPHA, bias=sl(b["fshift"], 0))
            first = ()
        else:
            convs[0].forward(x, out=raw[..., :q])
            first = ()
        jobs = []
        for j in first + (1, 2, 3):
            c = convs[j]
            c._x = x
            if self._fold:
                jobs.append((x, c.wp_f, sl(b["fshift"], j), c.k, c.dil, raw[..., j * q:(j + 1) * q], ACT_LRELU, KERAS_LRELU_ALPHA))
            else:
                jobs.append((x, c.wp_f, c.bias.data, c.k, c.dil, raw[..., j * q:(j + 1) * q], ACT_NONE, 0.0))
        ops.conv2d_fwd_multi(jobs)

    def _branches_bwd(self, st, draw, dx):
        """Backward of the four parallel convs: the three dilated weight gradients as ONE multi-job launch on the side
        stream (a third of the split-K slabs each), the backward-data passes accumulating into ``dx`` on the main stream."""
        q = self.out_channels // 4
        convs = [getattr(self, f"conv{st}_{j}") for j in range(4)]
        dys = [draw[..., j * q:(j + 1) * q] for j in range(4)]
        ops.wgrad_later(lambda: ops.conv2d_wgrad_multi([convs[j].wgrad_job(dys[j]) for j in (1, 2, 3)]), convs[1]._x, draw)
        convs[0].backward(dys[0], need_dx=False, skip_bias=True)
        if _FUSED_DGRAD:
            ops.conv2d_dgrad_branches(draw, self._wd_cat[st], [c.k for c in convs], [c.dil for c in convs], [j * q for j in range(4)], q, dx)
        else:
            for j in range(4):
                ops.conv2d_dgrad(dys[j], convs[j].wp_d, convs[j].k, convs[j].dil, dx, None, j > 0)

    def skip_slot(self, B, H, W, device):
        """Allocate this block's concat buffer [B,2H,2W,oc+skip] ahead of time and return its skip slice: a producer that
        writes the skip tensor THERE makes ``tf.concat([x, skip])`` (Decoder.py:66) free."""
        self._cat = ops.new_act(B, 2 * H, 2 * W, self.out_channels + self.skip_channels, device)
        return self._cat[..., self.out_channels:]

    def forward(self, x, skip=None, out=None):
        """x [B,h,w,in]; skip [B,2h,2w,skip] or None; ``out``: optional [B,2h,2w,oc] slice to write the result into."""
        B, H, W, _, _ = ops.geom(x)
        dev = x.device
        oc, q = self.out_channels, self.out_channels // 4
        has_skip = skip is not None
        c1 = oc + (self.skip_channels if has_skip else 0)
        assert has_skip or self.skip_channels == 0, "block was built for a skip connection"
        pre = getattr(self, "_cat", None)
        in_place = (has_skip and pre is not None and tuple(pre.shape) == (B, 2 * H, 2 * W, c1)
                    and skip.data_ptr() == pre[..., oc:].data_ptr() and ops.geom(skip)[4] == c1)
        cat = pre if in_place else ops.new_act(B, 2 * H, 2 * W, c1, dev)
        self._cat = None
        if self._qup is not None:
            self._qup.forward(x, out=cat[..., :oc])
        else:
            self.up.forward(x, out=cat[..., :oc])                                     # :63
        if has_skip and not in_place:
            ops.copy_channels(skip, cat[..., oc:])                                    # :66
        self._fold = self.conv1_0.fold_scale() is not None and self.conv2_0.fold_scale() is not None     # both stages in inference mode
        out = out if out is not None else ops.new_act(B, 2 * H, 2 * W, oc, dev)
        if self._fold:   # :67-76, :79-88 with BN + LeakyReLU in the conv epilogues: no pre-norm tensors, no norm launches
            act1 = ops.new_act(B, 2 * H, 2 * W, oc, dev)
            self._branches_fwd("1", cat, act1)
            self._branches_fwd("2", act1, out)
            self._has_skip, self._raw = has_skip, (act1, out)
            return out
        raw1 = ops.new_act(B, 2 * H, 2 * W, oc, dev)
        self._branches_fwd("1", cat, raw1)                                            # :67-74 convs write their channel slice
        act1 = self._bn_fwd("1", raw1, ops.new_act(B, 2 * H, 2 * W, oc, dev))        # :68-76 four BNs + LeakyReLU, one launch
        raw2 = ops.new_act(B, 2 * H, 2 * W, oc, dev)
        self._branches_fwd("2", act1, raw2)                                           # :79-86
        self._bn_fwd("2", raw2, out)                                                  # :80-88
        self._has_skip, self._raw = has_skip, (raw1, raw2)
        return out

    def backward(self, dout):
        """dout [B,2h,2w,oc] (may be a channel slice) -> (dx, dskip)."""
        oc, q = self.out_channels, self.out_channels // 4
        B, H2, W2, _, _ = ops.geom(dout)
        dev = dout.device
        raw1, raw2 = self._raw
        draw2 = self._bn_bwd("2", raw2, dout, ops.new_act(B, H2, W2, oc, dev))
        dact1 = ops.new_act(B, H2, W2, oc, dev)
        self._branches_bwd("2", draw2, dact1)
        draw1 = self._bn_bwd("1", raw1, dact1, ops.new_act(B, H2, W2, oc, dev))
        c1 = oc + (self.skip_channels if self._has_skip else 0)
        dcat = ops.new_act(B, H2, W2, c1, dev)
        self._branches_bwd("1", draw1, dcat)
        dx = self._qup.backward(dcat[..., :oc]) if self._qup is not None else self.up.backward(dcat[..., :oc])
        dskip = dcat[..., oc:] if self._has_skip else None
        return dx, dskip

    def __call__(self, x, skip=None, *args, **kwargs):
        return self.forward(x, skip)


class DecoderCup(nn.Module):
    """Decoder.py:98-146.  ``forward(hidden_states [B,N,hidden], features)`` -> class probabilities [B,H,W,classes] (fp32).

    The literal 16x5 grid of Decoder.py:128,140 is generalised to ``grid=(H/16, W/16)``; with
    H=256, W=80 it reproduces the reference's reshapes exactly.
    """

    def __init__(self, num_classes, wDecay=None, hidden_size=512, grid=(16, 5), norm="ln"):
        super().__init__()
        head_channels = 256
        self.wDecay, self.num_classes, self.hidden_size, self.grid = wDecay, num_classes, hidden_size, tuple(grid)
        self.conv_more = Conv2D(hidden_size, head_channels, 3)                        # :103
        self.LeakyReLU1 = LeakyReLU()
        # :112 LayerNormalization; the copy in TBI_TransUNet.py:304 uses BatchNormalization
        self.bn1 = LayerNormalization(head_channels) if norm == "ln" else BatchNormalization(head_channels)
        skip_channels = [256, 128, 64]
        blocks, cin = [], head_channels
        for i, sk in enumerate(skip_channels):                                        # :114-117
            blocks.append(DecoderBlock(sk, wDecay, in_channels=cin))
            cin = sk + hidden_size // (4 ** (i + 1))
        self.blocks = nn.ModuleList(blocks)
        self.head = Conv2DTranspose(cin, num_classes, 3)                              # :120 (softmax fused downstream)
        # <= 4 classes: the head runs in "quad" form - a 2x2-tap conv at the input resolution whose 16 channels are the four
        # output parities (one HBM-bound launch each way instead of a 4-class gather GEMM and a 9-tap per-tap weight gradient)
        self.quad_head = num_classes <= 4 and os.environ.get("USSEG_QUAD_HEAD", "1") != "0"
        self._quad = QuadHead(self.head) if self.quad_head else None

    def on_finalize(self, device):
        if self._quad is not None:
            self._quad.on_finalize(device)

    def pack_jobs(self):
        return self._quad.pack_jobs() if self._quad is not None else []

    def _head_forward(self, x):
        return self._quad.forward(x)

    def _head_backward(self, dl4):
        return self._quad.backward(dl4)

    def prepare(self, B, device):
        """Allocate the blocks' concat buffers ahead of the encoder and return the destinations [x_3, x_2, x_1] for its stage
        outputs (``ResNest.forward(x, outs=...)``): the skip connections then need no copy (Decoder.py:66)."""
        gh, gw = self.grid
        return [blk.skip_slot(B, gh * 2 ** i, gw * 2 ** i, device) for i, blk in enumerate(self.blocks)]

    def forward(self, hidden_states, features: Optional[List[torch.Tensor]] = None, return_logits=False, return_head_input=False):
        assert features is not None, "this implementation is built for the skip-connected configuration the drivers use"
        B = hidden_states.shape[0]
        gh, gw = self.grid
        hs = self.hidden_size
        dev = hidden_states.device
        y = hidden_states.reshape(B, gh, gw, hs)                                      # :128 (a view: same memory)
        self._hidden_shape = hidden_states.shape
        # the hidden state re-injected at every scale by a raw row-major reshape (:140-141): all three scales in one launch
        cats, slots = [], []
        for i, blk in enumerate(self.blocks):
            s, c0 = 2 ** (i + 1), hs // (4 ** (i + 1))
            cats.append(ops.new_act(B, gh * s, gw * s, blk.out_channels + c0, dev))
            slots.append(cats[i][..., blk.out_channels:])
        ops.reinject_hidden(y if y.is_contiguous() else y.contiguous(), slots, backward=False)
        x = self.conv_more.forward(y)                                                 # :129
        x = self.bn1.forward(x, ACT_LRELU, KERAS_LRELU_ALPHA)                         # :130-131
        for i, blk in enumerate(self.blocks):                                         # :132
            blk.forward(x, features[i], out=cats[i][..., :blk.out_channels])          # :137 (x0 already sits behind it, :140-141)
            x = cats[i]
        Ho, Wo = 2 * x.shape[1], 2 * x.shape[2]
        self.out_hw = (Ho, Wo)
        if return_head_input:       # the caller runs head + softmax + loss as one launch (QuadHead.forward_loss)
            return x
        # :142; logits fp32: [B,Ho,Wo,4] or, in quad form, [B,Ho/2,Wo/2,16] (softmax_loss indexes it through quad_w)
        logits = self._head_forward(x) if self.quad_head else self.head.forward(x, out_f32=True)
        self._logits = logits
        if return_logits:
            return logits
        probs = torch.empty((B, Ho, Wo, self.num_classes), dtype=torch.float32, device=dev)
        ops.softmax_loss(logits, None, probs, None, None, HW=Ho * Wo, C_classes=self.num_classes, quad_w=self.quad_w)          # :121
        return probs

    @property
    def quad_w(self):
        return self.out_hw[1] if self.quad_head else 0

    def backward(self, dlogits):
        """dlogits: bf16 [B,H,W,8] gradient w.r.t. the head's pre-softmax output -> (d_hidden [B,N,hidden], [d_x3,d_x2,d_x1])."""
        B = dlogits.shape[0]
        gh, gw = self.grid
        hs = self.hidden_size
        dev = dlogits.device
        d = self._head_backward(dlogits) if self.quad_head else self.head.backward(dlogits)
        d_hidden = ops.new_act(B, gh, gw, hs, dev)
        dfeats = [None, None, None]
        reinj = []                                   # gradient slices of the three re-injections (views keep their buffers alive)
        for i in reversed(range(3)):
            blk = self.blocks[i]
            reinj.append(d[..., blk.out_channels:])
            d, dfeats[i] = blk.backward(d[..., :blk.out_channels])
        ops.reinject_hidden(d_hidden, reinj, backward=True)                           # d_hidden = sum of the three, one launch
        d = self.bn1.backward(d, dbias=self.conv_more.bias.grad)
        self.conv_more.backward(d, dx=d_hidden, accumulate_dx=True, skip_bias=True)
        return d_hidden.reshape(self._hidden_shape), dfeats

    def __call__(self, hidden_states, features=None, *args, **kwargs):
        return self.forward(hidden_states, features)


# ------------------------------------------------------------------------------------------------ KSAC (Decoder.py:150-346)
def ksac_effective_dilations(dilation_rates_list, as_written=True):
    """kernel_sharing_conv2d re-assigns ``value`` inside its loop over the rates (Decoder.py:280-285), so the per-tap shifts
    ACCUMULATE: as written, branch j is a 'same' dilated convolution with dilation d_0 + ... + d_j ((1,2,4,8,16) -> (1,3,7,15,31)).
    ``as_written=False``: the listed rates (the KSAC paper's intent).  The equivalence is proven against a line-by-line
    restatement of the reference in tests/test_oracle_kat.py."""
    if not as_written:
        return tuple(dilation_rates_list)
    out, acc = [], 0
    for d in dilation_rates_list:
        acc += d
        out.append(acc)
    return tuple(out)


class KernelSharingConv(nn.Module):
    """Decoder.py:294-346: ONE 3x3 kernel [kh,kw,Cin,filters] (no bias) applied at every dilation rate, each branch followed by its
    own BatchNormalization (inference mode as driven) and exact GELU; ``forward`` returns the LIST of branch tensors.

    MI355X form: the branches are independent jobs that share one packed weight operand - forward and backward-data run as
    multi-job conv launches (rates that divide the image take the LDS-DMA halo kernels, the others the gather GEMM), the five
    weight gradients accumulate into the one shared variable.  Keras builds the kernel lazily from the input shape; here the
    input width is the keyword ``in_channels``."""

    def __init__(self, filters, kernel_size, dilation_rates_list=(1, 2, 4, 8, 16), trainable=True, kernel_initializer="glorot_uniform",
                 kernel_regularizer=None, use_bn=True, name=None, *, in_channels, as_written=True):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        assert ks == 3, "the reference instantiates 3x3 kernels (Decoder.py:154)"
        self.filters, self.dilation_rates_list, self.use_bn = int(filters), tuple(dilation_rates_list), use_bn
        self.dilations = ksac_effective_dilations(self.dilation_rates_list, as_written)
        init = "he" if str(kernel_initializer).lower().startswith("he") else "glorot"
        self.conv = Conv2D(in_channels, self.filters, 3, 1, init=init)
        del self.conv._parameters["bias"]                      # add_weight('kernel') only (:322-327): no bias variable
        self.conv.bias = None
        if use_bn:
            for r in self.dilation_rates_list:                 # :333 name=f"bn_r_{r}"
                setattr(self, f"bn_r_{r}", BatchNormalization(self.filters))

    @property
    def kernel(self):
        return self.conv.kernel

    def forward(self, inputs):
        B, H, W, _, _ = ops.geom(inputs)
        dev = inputs.device
        c = self.conv
        c._x = inputs
        raws = [ops.new_act(B, H, W, c.cout_p, dev) for _ in self.dilations]
        jobs = [(inputs, c.wp_f, None, 3, d, raws[j], ACT_NONE, 0.0) for j, d in enumerate(self.dilations)]
        for i in range(0, len(jobs), 4):
            ops.conv2d_fwd_multi(jobs[i:i + 4])                # :266-288, all rates of a group in one launch where the geometry allows
        self._raws = raws
        if not self.use_bn:
            self._outs = [ops.act_fwd(r, torch.empty_like(r), ops.ACT_GELU, 0.0) for r in raws]
            return self._outs
        return [getattr(self, f"bn_r_{r}").forward(raw, ops.ACT_GELU, 0.0) for raw, r in zip(raws, self.dilation_rates_list)]   # :337-345

    def backward(self, dys, need_dx=True):
        """dys: list of gradients w.r.t. the branch outputs -> gradient w.r.t. the input; the kernel gradient accumulates."""
        c, x = self.conv, self.conv._x
        draws = []
        for dy, raw, r in zip(dys, self._raws, self.dilation_rates_list):
            draws.append(getattr(self, f"bn_r_{r}").backward(dy) if self.use_bn else ops.act_bwd(raw, dy, torch.empty_like(dy), ops.ACT_GELU, 0.0))
        plain = c.cin_p == c.cin and c.cout_p == c.cout
        wj = [(x, dr, 3, d, c.kernel.grad if plain else None, None if plain else c._wgrad_map()) for dr, d in zip(draws, self.dilations)]
        for i in range(0, len(wj), 4):
            ops.conv2d_wgrad_multi(wj[i:i + 4])                # five gradients into ONE variable (deferred finishes are serialised per destination)
        if not need_dx:
            return None
        B, H, W, _, _ = ops.geom(x)
        dx = ops.new_act(B, H, W, c.cin_p, x.device)
        for j, (dr, d) in enumerate(zip(draws, self.dilations)):
            ops.conv2d_dgrad(dr, c.wp_d, 3, d, dx, None, j > 0)
        return dx

    def __call__(self, inputs, *args, **kwargs):
        return self.forward(inputs)


class KSACBlock(nn.Module):
    """Decoder.py:150-176: Conv2DTranspose(3x3, s2) -> concat skip -> KernelSharingConv -> KernelSharingConv.

    As written the block cannot run: ``conv1`` returns a LIST of five tensors, ``tf.convert_to_tensor`` stacks it to
    [5,N,H,W,C] (:168) and ``conv2`` then reshapes that with N,H,W read from the wrong axes (:241-249), which fails for W > 1; no
    driver instantiates it.  ``fuse="sum"`` (default) is the executable reading this class offers: the branches of each
    KernelSharingConv are SUMMED (the fusion of the KSAC paper); ``fuse=None`` reproduces the reference's failure."""

    def __init__(self, out_channels, wDecay=None, *, in_channels=None, skip_channels=None, fuse="sum", as_written=True):
        super().__init__()
        oc = out_channels
        self.wDecay, self.kernel_size, self.out_channels, self.fuse = wDecay, [3, 3], oc, fuse
        self.in_channels = in_channels if in_channels is not None else oc
        self.skip_channels = oc if skip_channels is None else skip_channels
        self.conv1 = KernelSharingConv(oc, self.kernel_size, kernel_regularizer=wDecay, name="KSAC_1", kernel_initializer="HeNormal",
                                       in_channels=oc + self.skip_channels, as_written=as_written)
        self.conv2 = KernelSharingConv(oc, self.kernel_size, kernel_regularizer=wDecay, name="KSAC_1", kernel_initializer="HeNormal",
                                       in_channels=oc, as_written=as_written)
        self.up = Conv2DTranspose(self.in_channels, oc, 3)

    @staticmethod
    def _sum(ts):
        out = ts[0].clone()
        for t in ts[1:]:
            ops.copy_channels(t, out, accumulate=True)
        return out

    def forward(self, x, skip=None):
        if self.fuse is None:
            raise ValueError("KSACBlock as written (Decoder.py:168-171) feeds the stacked [5,N,H,W,C] list into a layer that reshapes it as "
                             "[N,H*W,c]: it cannot run for W > 1; build it with fuse='sum'")
        B, H, W, _, _ = ops.geom(x)
        oc = self.out_channels
        cat = ops.new_act(B, 2 * H, 2 * W, oc + (self.skip_channels if skip is not None else 0), x.device)
        self.up.forward(x, out=cat[..., :oc])                                         # :163
        if skip is not None:
            ops.copy_channels(skip, cat[..., oc:])                                    # :166
        self._has_skip = skip is not None
        y = self._sum(self.conv1.forward(cat))                                        # :167 (+ branch fusion)
        return self._sum(self.conv2.forward(y))                                       # :169

    def backward(self, dout):
        n = len(self.conv2.dilations)
        dy = self.conv2.backward([dout] * n)                                          # d(sum)/d(branch) = identity
        dcat = self.conv1.backward([dy] * n)
        oc = self.out_channels
        dx = self.up.backward(dcat[..., :oc])
        return dx, (dcat[..., oc:] if self._has_skip else None)

    def __call__(self, x, skip=None, *args, **kwargs):
        return self.forward(x, skip)
