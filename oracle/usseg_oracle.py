"""CPU oracle for the ResNeSt/UNet hot path of silverlight6/Ultrasound_Modeling.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ultrasound_modeling_amd/`` may import
this file; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.

PARITY UNPINNED: the reference ships no tests, golden vectors or weights
 and needs TensorFlow, which is not installed here, so it can
neither be imported nor run.  This file is a restatement written from the
reference source text plus the Keras-default sheet of SURVEY.md Appendix A; it is
pinned only by analytic known-answer tests (tests/test_oracle_kat.py), by
finite-difference gradient checks and by an independent NumPy loop
implementation of every primitive (tests/test_oracle_numpy_xcheck.py).

Conventions
-----------
* activations are NHWC ``torch`` tensors (float64 by default) exactly as at the
  reference surface; weights are in Keras layout: Conv2D ``[kh,kw,Cin,Cout]``,
  Conv2DTranspose ``[kh,kw,Cout,Cin]``.
* every function cites the reference ``file:line`` it follows (paths are
  relative to the reference repo root).
* parameters are plain ``dict[str, Tensor]`` whose keys are the attribute paths
  of the reference modules (``conv_1.cardinal_blocks.0.conv1.kernel`` ...), the
  same names the product's ``load_params`` accepts.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# ----------------------------------------------------------------------------
# Keras / TF defaults the reference relies on.  One place.
# ----------------------------------------------------------------------------
KERAS = dict(
    leaky_relu_alpha=0.3,      # tf.keras.layers.LeakyReLU() default (A.3)
    elu_alpha=1.0,             # tf.keras.layers.ELU() default (A.3)
    bn_eps=1e-3,               # BatchNormalization epsilon (A.4)
    bn_momentum=0.99,          # BatchNormalization momentum (A.4)
    ln_eps=1e-3,               # LayerNormalization epsilon (A.5)
    vit_ln_eps=1e-6,           # VisionTransformer.py:131-132,158
    cce_label_smoothing=0.1,   # VisionTransformer.py:205
    cce_clip=1e-7,             # Keras backend epsilon (A.6)
    adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-7,  # tf.optimizers.Adam (A.6)
    clip_norm=1.0,             # VisionTransformer.py:244
    bn_training=False,         # as driven: BN runs in inference mode (A.4)
    # Judgement call a TensorFlow-equipped check can flip (parity is unpinned): Keras >= 2.4
    # backend.categorical_crossentropy, given a tensor that a Keras softmax produced (Decoder.py:121 -> VisionTransformer.py:205), finds its
    # cached `_keras_logits` and calls softmax_cross_entropy_with_logits on THEM: no renormalisation, no 1e-7 clip.  False = the documented
    # probability path (renormalise, clip), which the product implements; the two differ only on saturated pixels (p < 1e-7).
    cce_cached_logits=False,
)


# ----------------------------------------------------------------------------
# optional emulation of the product's bf16 storage (diagnostic for tests only; default off = exact fp64)
# ----------------------------------------------------------------------------
STORAGE_DTYPE = None   # set to torch.bfloat16 to round every stored activation (and its gradient) and conv weights


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, round_grad):
        ctx.dt, ctx.round_grad = dt, round_grad
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dt).to(g.dtype) if ctx.round_grad else g), None, None


def _q(x: Tensor) -> Tensor:
    """Round a stored activation (forward) and its gradient (backward) to STORAGE_DTYPE; identity when unset."""
    return x if STORAGE_DTYPE is None else _RoundSTE.apply(x, STORAGE_DTYPE, True)


def _qw(w: Tensor) -> Tensor:
    """Round a conv weight to STORAGE_DTYPE for the matmul only (fp32 master weights keep the full gradient)."""
    return w if STORAGE_DTYPE is None else _RoundSTE.apply(w, STORAGE_DTYPE, False)


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def _nchw(x: Tensor) -> Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: Tensor) -> Tensor:
    return x.permute(0, 2, 3, 1)


def conv2d_same(x: Tensor, w: Tensor, b: Optional[Tensor] = None, dilation: int = 1) -> Tensor:
    """Keras Conv2D, stride 1, padding 'SAME', NHWC, kernel [kh,kw,Cin,Cout].

    ResNest.py:14,17,21,77,82,122,128,160,166; Decoder.py:11-25,36-50,103;
    TBI_ResNest.py:83-88,140,143,162,167,189,195.  Appendix A.1: cross-correlation,
    odd k, stride 1 => symmetric zero pad d*(k-1)/2.
    """
    kh, kw = w.shape[0], w.shape[1]
    assert kh % 2 == 1 and kw % 2 == 1
    wt = _qw(w).permute(3, 2, 0, 1)  # -> [Cout,Cin,kh,kw]
    y = F.conv2d(_nchw(x), wt, b, stride=1,
                 padding=(dilation * (kh - 1) // 2, dilation * (kw - 1) // 2), dilation=dilation)
    return _q(_nhwc(y))


def conv2d_transpose_s2_same(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """Keras Conv2DTranspose(strides=2, padding='same'), kernel [kh,kw,Cout,Cin].

    Decoder.py:57,120 (k=3) and TBI_ResNest.py:124,210 (k=4).  Appendix A.2:
    k=3: out[2i+k] += x[i]*w[k], keep rows/cols 0..2H-1 (pad_before 0, crop the end);
    k=4: out[2i+k-1] += x[i]*w[k] (pad 1 / 1).
    """
    k = w.shape[0]
    assert w.shape[1] == k and k in (3, 4)
    B, H, W, _ = x.shape
    wt = _qw(w).permute(3, 2, 0, 1)  # [Cin,Cout,kh,kw] (torch conv_transpose2d layout)
    if k == 3:
        y = F.conv_transpose2d(_nchw(x), wt, b, stride=2, padding=0)[..., : 2 * H, : 2 * W]
    else:
        y = F.conv_transpose2d(_nchw(x), wt, b, stride=2, padding=1)
    return _q(_nhwc(y))


def leaky_relu(x: Tensor) -> Tensor:
    """tf.keras.layers.LeakyReLU() (alpha 0.3) - ResNest.py:16,20,24,87,126,133,165; Decoder.py:31,111."""
    return _q(torch.where(x >= 0, x, KERAS["leaky_relu_alpha"] * x))


def elu(x: Tensor) -> Tensor:
    """tf.keras.layers.ELU() - TBI_ResNest.py:84,87,91,145,165,170,191."""
    return _q(torch.where(x > 0, x, KERAS["elu_alpha"] * torch.expm1(torch.clamp(x, max=0.0))))


def avg_pool2(x: Tensor) -> Tensor:
    """AveragePooling2D(pool_size=2, strides=2), VALID - ResNest.py:25-28; TBI_ResNest.py:92-107."""
    B, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    return _q(x.reshape(B, H // 2, 2, W // 2, 2, C).mean(dim=(2, 4)))


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: Optional[float] = None) -> Tensor:
    """LayerNormalization(axis=-1): per pixel over channels, biased variance (A.5).

    ResNest.py:86,125,132,164; Decoder.py:112 (eps 1e-3); VisionTransformer.py:131-132,158 (eps 1e-6).
    """
    eps = KERAS["ln_eps"] if eps is None else eps
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def batch_norm(x: Tensor, gamma: Tensor, beta: Tensor, moving_mean: Tensor, moving_var: Tensor,
               training: Optional[bool] = None):
    """BatchNormalization(axis=-1), eps 1e-3, momentum 0.99 (A.4).

    ResNest.py:19,23; Decoder.py:32-35,51-54; TBI_ResNest.py:90,144,164,169,190,213.
    As the reference is driven the layer runs in inference mode (A.4): moving statistics
    are used and never change.  ``training=True`` gives the Keras training-mode behaviour
    (batch mean / biased variance, moving stats updated) and returns the new stats too.
    """
    training = KERAS["bn_training"] if training is None else training
    eps = KERAS["bn_eps"]
    if not training:
        return (x - moving_mean) / torch.sqrt(moving_var + eps) * gamma + beta
    red = tuple(range(x.dim() - 1))
    mu = x.mean(dim=red)
    var = ((x - mu) ** 2).mean(dim=red)
    y = (x - mu) / torch.sqrt(var + eps) * gamma + beta
    m = KERAS["bn_momentum"]
    return y, (moving_mean * m + mu.detach() * (1 - m), moving_var * m + var.detach() * (1 - m))


def softmax_lastaxis(z: Tensor) -> Tensor:
    """tf.keras.activations.softmax(z) = softmax over axis -1 (A.3)."""
    return torch.softmax(z, dim=-1)


# ----------------------------------------------------------------------------
# channel arithmetic (ResNest.py:73,120-121,160; TBI_ResNest.py:134,157-158,189)
# ----------------------------------------------------------------------------
def cardinal_channels(stage_out: int, radix: int, kpaths: int) -> Tuple[int, int, int]:
    """-> (cv11, cvkk, attention_hidden) for a residual_S stage with ``stage_out`` output channels."""
    half = stage_out // 2                      # ResNest.py:73
    cv11 = int(half / radix / kpaths)          # ResNest.py:120
    cvkk = int(half / kpaths)                  # ResNest.py:121
    return cv11, cvkk, cvkk // 2               # ResNest.py:160


# ----------------------------------------------------------------------------
# Arch B: ResNest.py
# ----------------------------------------------------------------------------
def _norm(x: Tensor, P: Params, name: str, norm: str) -> Tensor:
    """The layer called ``*_bn`` / ``bn1``: LayerNormalization in ResNest.py / Decoder.py (ResNest.py:86,125,132,164; Decoder.py:112),
    BatchNormalization (inference mode as driven, App. A.4) in the older copy TBI_TransUNet.py (:304,426,465,472,503)."""
    if norm == "ln":
        return layer_norm(x, P[name + ".gamma"], P[name + ".beta"])
    assert norm == "bn"
    return _bn(x, P, name)


def split_attention(inputs: Sequence[Tensor], P: Params, prefix: str, radix: int, norm: str = "ln") -> Tensor:
    """split_attention.forward - ResNest.py:171-199.

    sum of the radix inputs, global average pool, dense1 (1x1)+LN+LeakyReLU, then ``radix``
    times the SAME dense2 followed by a softmax over the CHANNEL axis (sigmoid if radix==1),
    output = sum_r inputs[r] * z.
    """
    global STORAGE_DTYPE
    holder = inputs[0]
    for t in inputs[1:]:
        holder = holder + t                                            # :173-177
    y = holder.mean(dim=(1, 2))[:, None, None, :]                      # :179-180
    saved_storage, STORAGE_DTYPE = STORAGE_DTYPE, None                 # the product runs this tiny MLP in fp32
    y = conv2d_same(y, P[prefix + "dense1.kernel"], P[prefix + "dense1.bias"])       # :182
    y = _norm(y, P, prefix + "dense1_bn", norm)                        # :183 (TBI_TransUNet.py:503: BatchNormalization)
    y = leaky_relu(y)                                                  # :184
    out = None
    for r in range(radix):                                             # :187
        z = conv2d_same(y, P[prefix + "dense2.kernel"], P[prefix + "dense2.bias"])   # :188
        z = torch.sigmoid(z) if radix == 1 else softmax_lastaxis(z)    # :189-192
        out = inputs[r] * z if out is None else out + inputs[r] * z    # :194-197
    STORAGE_DTYPE = saved_storage
    return _q(out)


def cardinal(x: Tensor, P: Params, prefix: str, radix: int, as_executed: bool = True, norm: str = "ln") -> Tensor:
    """cardinal.forward - ResNest.py:136-147.

    The loop applies the SAME conv1/conv1_bn/conv2/conv2_bn ``radix`` times to the same input,
    so the radix branches are identical tensors.  ``as_executed=False`` computes the branch once
    and re-uses it (must give identical results; tests check that).
    """
    def branch():
        y = conv2d_same(x, P[prefix + "conv1.kernel"], P[prefix + "conv1.bias"])        # :139
        y = leaky_relu(_norm(y, P, prefix + "conv1_bn", norm))                          # :140-141 (TBI_TransUNet.py:465)
        y = conv2d_same(y, P[prefix + "conv2.kernel"], P[prefix + "conv2.bias"])        # :142
        return leaky_relu(_norm(y, P, prefix + "conv2_bn", norm))                       # :143-144 (TBI_TransUNet.py:472)
    if as_executed:
        inputs = [branch() for _ in range(radix)]
    else:
        y = branch()
        inputs = [y] * radix
    return split_attention(inputs, P, prefix + "split.", radix, norm)  # :147


def residual_S(x: Tensor, P: Params, prefix: str, radix: int, kpaths: int,
               as_executed: bool = True, norm: str = "ln") -> Tensor:
    """residual_S.forward - ResNest.py:89-104."""
    cards = [cardinal(x, P, f"{prefix}cardinal_blocks.{k}.", radix, as_executed, norm) for k in range(kpaths)]
    concats_1 = torch.cat(cards, dim=3)                                 # :91-96
    concats_2 = conv2d_same(concats_1, P[prefix + "concats_2.kernel"], P[prefix + "concats_2.bias"])  # :98
    sc = conv2d_same(x, P[prefix + "convtmp_sc.kernel"], P[prefix + "convtmp_sc.bias"])  # :99
    sc = leaky_relu(_norm(sc, P, prefix + "convtmp_scbn", norm))                        # :100-101 (TBI_TransUNet.py:426)
    return _q(sc + concats_2)                                           # :102


def _bn(x, P, name, training=None):
    out = batch_norm(x, P[name + ".gamma"], P[name + ".beta"], P[name + ".moving_mean"],
                     P[name + ".moving_variance"], training)
    return out[0] if isinstance(out, tuple) else out


def resnest_forward(x: Tensor, P: Params, radix: int, kpaths: int, prefix: str = "",
                    as_executed: bool = True, taps: Optional[dict] = None, norm: str = "ln"):
    """ResNest.forward - ResNest.py:38-55.  Returns (x_4, [x_3, x_2, x_1])."""
    p = prefix
    x = leaky_relu(conv2d_same(x, P[p + "conv1.kernel"], P[p + "conv1.bias"]))             # :39-40
    x = conv2d_same(x, P[p + "convtmp_1.kernel"], P[p + "convtmp_1.bias"])                  # :41
    x = leaky_relu(_bn(x, P, p + "convtmp_1bn"))                                           # :42-43
    x = conv2d_same(x, P[p + "convtmp_2.kernel"], P[p + "convtmp_2.bias"])                  # :44
    x = leaky_relu(_bn(x, P, p + "convtmp_2bn"))                                           # :45-46
    if taps is not None:
        taps["stem"] = x
    x = avg_pool2(x)                                                                        # :47
    x_1 = residual_S(x, P, p + "conv_1.", radix, kpaths, as_executed, norm)                       # :48
    x_2 = residual_S(avg_pool2(x_1), P, p + "conv_2.", radix, kpaths, as_executed, norm)          # :49-50
    x_3 = residual_S(avg_pool2(x_2), P, p + "conv_3.", radix, kpaths, as_executed, norm)          # :51-52
    x_4 = residual_S(avg_pool2(x_3), P, p + "conv_4.", radix, kpaths, as_executed, norm)          # :53-54
    return x_4, [x_3, x_2, x_1]                                                             # :55


# ----------------------------------------------------------------------------
# Arch B: Decoder.py
# ----------------------------------------------------------------------------
def decoder_block(x: Tensor, skip: Optional[Tensor], P: Params, prefix: str) -> Tensor:
    """DecoderBlock.forward - Decoder.py:61-91 (self.pool / self.conv1_4 are dead, :26-29)."""
    x = conv2d_transpose_s2_same(x, P[prefix + "up.kernel"], P[prefix + "up.bias"])        # :63
    if skip is not None:
        x = torch.cat([x, skip], dim=3)                                                     # :66
    for stage in ("1", "2"):                                                                # :67-76, :79-88
        outs = []
        for j, d in enumerate((1, 2, 4, 8)):   # conv?_0 is 1x1; conv?_1..3 are 3x3 dilated 2,4,8 (:11-25,:36-50)
            name = f"{prefix}conv{stage}_{j}"
            y = conv2d_same(x, P[name + ".kernel"], P[name + ".bias"], dilation=d)
            outs.append(_bn(y, P, f"{prefix}bn{stage}_{j}"))
        x = leaky_relu(torch.cat(outs, dim=3))
    return x


def decoder_cup(hidden: Tensor, features: Optional[List[Tensor]], P: Params, grid: Tuple[int, int],
                prefix: str = "", norm: str = "ln") -> Tensor:
    """DecoderCup.forward - Decoder.py:124-143.

    ``grid`` generalises the literal (16, 5) of :128,:140 to (H/16, W/16).
    The hidden state is re-injected at every scale by a RAW row-major reshape (:140).
    Returns class probabilities (softmax is the head's activation, :121).
    """
    B = hidden.shape[0]
    gh, gw = grid
    y = hidden
    x = hidden.reshape(B, gh, gw, -1)                                                        # :128
    x = conv2d_same(x, P[prefix + "conv_more.kernel"], P[prefix + "conv_more.bias"])        # :129
    x = leaky_relu(_norm(x, P, prefix + "bn1", norm))                                       # :130-131 (TBI_TransUNet.py:304,549-550)
    for i in range(3):                                                                      # :132
        skip = features[i] if features is not None else None                               # :133-136
        x = decoder_block(x, skip, P, f"{prefix}blocks.{i}.")                               # :137
        x0 = y.reshape(B, gh * 2 ** (i + 1), gw * 2 ** (i + 1), -1)                          # :140
        x = torch.cat([x, x0], dim=3)                                                       # :141
    logits = conv2d_transpose_s2_same(x, P[prefix + "head.kernel"], P[prefix + "head.bias"])  # :142
    return softmax_lastaxis(logits)                                                          # :121


# ----------------------------------------------------------------------------
# Decoder.py:150-346 - kernel-sharing atrous convolution (KSAC); defined in the reference, used by no driver
# ----------------------------------------------------------------------------
KSAC_DILATIONS = (1, 2, 4, 8, 16)   # Decoder.py:294


def ksac_effective_dilations(dilations=KSAC_DILATIONS, as_written: bool = True):
    """kernel_sharing_conv2d (Decoder.py:266-288) computes one 1x1 product per tap and, for every dilation rate in turn, slices
    and zero-pads it to apply the tap's shift - but it re-assigns ``value`` inside the loop over the rates (:280-285), so rate j
    shifts the ALREADY shifted tensor: the shifts accumulate.  Shifting twice in the same direction with zero fill equals one
    shift by the sum, hence as written branch j is an ordinary 'same' dilated convolution with dilation d_0 + ... + d_j:
    (1, 2, 4, 8, 16) -> (1, 3, 7, 15, 31).  ``as_written=False`` gives the listed rates (what the KSAC paper intends)."""
    if not as_written:
        return tuple(dilations)
    out, acc = [], 0
    for d in dilations:
        acc += d
        out.append(acc)
    return tuple(out)


def kernel_sharing_conv2d(x: Tensor, w: Tensor, dilations=KSAC_DILATIONS, as_written: bool = True) -> List[Tensor]:
    """Decoder.py:227-291: the SAME 3x3 kernel [kh,kw,Cin,Cout] (no bias) at every dilation rate -> list of tensors."""
    return [conv2d_same(x, w, None, dilation=d) for d in ksac_effective_dilations(dilations, as_written)]


def kernel_sharing_conv2d_literal(x: Tensor, w: Tensor, dilations=KSAC_DILATIONS) -> List[Tensor]:
    """Line-by-line restatement of Decoder.py:227-291 (k*k batched matmuls, slice, pad, add) INCLUDING the cumulative re-slicing of
    ``value`` - the known-answer check for ksac_effective_dilations (pure slicing / padding, small inputs only)."""
    N, H, W, c = x.shape
    kh, kw, _, C = w.shape
    xs = x.reshape(N, H * W, c)
    ys = [torch.zeros(N, H, W, C, dtype=x.dtype) for _ in dilations]
    for i in range(kh * kw):
        r, q = i // kw, i % kw
        value = (xs @ w[r, q]).reshape(N, H, W, C)                                   # :268-269
        for j, d in enumerate(dilations):
            v_shift, h_shift = (kh // 2 - r) * d, (kw // 2 - q) * d                    # :183-184
            v0, h0 = (-v_shift if v_shift < 0 else 0), (-h_shift if h_shift < 0 else 0)
            v1, h1 = (-v_shift if v_shift > 0 else H), (-h_shift if h_shift > 0 else W)   # :277-278 (0 -> full extent)
            value = value[:, v0:v1, h0:h1, :]                                          # :280 (re-assigned: cumulative!)
            pad_v = (v_shift, 0) if v_shift > 0 else (0, -v_shift)                     # :196-197
            pad_h = (h_shift, 0) if h_shift > 0 else (0, -h_shift)
            value = F.pad(value, (0, 0, pad_h[0], pad_h[1], pad_v[0], pad_v[1]))       # :284
            ys[j] = ys[j] + value                                                      # :288
    return ys


def ksac_layer(x: Tensor, P: Params, prefix: str, dilations=KSAC_DILATIONS, as_written: bool = True) -> List[Tensor]:
    """KernelSharingConv.call - Decoder.py:335-346: shared-kernel convs, per-rate BatchNormalization (inference mode), exact GELU."""
    ys = kernel_sharing_conv2d(x, P[prefix + "kernel"], dilations, as_written)
    return [_q(gelu_exact(_bn(y, P, f"{prefix}bn_r_{r}"))) for y, r in zip(ys, dilations)]


def init_ksac_params(cin: int, filters: int, seed: int = 0, dtype=torch.float64, prefix: str = "", perturb: bool = False,
                     dilations=KSAC_DILATIONS, he: bool = True) -> Params:
    bld = _Builder(seed, dtype)
    w = _he_normal(bld.gen, (3, 3, cin, filters), 9 * cin, dtype) if he else _glorot_uniform(bld.gen, (3, 3, cin, filters), 9 * cin, 9 * filters, dtype)
    bld.P[prefix + "kernel"] = w
    for r in dilations:
        bld.norm(f"{prefix}bn_r_{r}", filters, bn=True, perturb=perturb)
    return bld.P


# ----------------------------------------------------------------------------
# Arch B wrapper: VisionTransformer.py
# ----------------------------------------------------------------------------
def gelu_exact(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def vit_block(x: Tensor, P: Params, prefix: str, num_heads: int = 4):
    """Block.forward - VisionTransformer.py:136-146 with Attention :32-53 and Mlp :68-74."""
    B, N, Hd = x.shape
    h = x
    xn = layer_norm(x, P[prefix + "attention_norm.gamma"], P[prefix + "attention_norm.beta"], KERAS["vit_ln_eps"])
    def dense(t, name):
        return t @ P[prefix + name + ".kernel"] + P[prefix + name + ".bias"]
    def heads(t):
        return t.reshape(B, N, num_heads, Hd // num_heads).permute(0, 2, 1, 3)            # :29-30
    q, k, v = heads(dense(xn, "attn.query")), heads(dense(xn, "attn.key")), heads(dense(xn, "attn.value"))
    scores = q @ k.transpose(-1, -2) / math.sqrt(float(num_heads))                           # :41-42 (sqrt(num_heads)!)
    probs = torch.softmax(scores, dim=3)                                                     # :23,:43
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, N, Hd)                                 # :47-49
    x = dense(ctx, "attn.out") + h                                                          # :50,:140
    h = x
    xn = layer_norm(x, P[prefix + "ffn_norm.gamma"], P[prefix + "ffn_norm.beta"], KERAS["vit_ln_eps"])
    m = gelu_exact(xn @ P[prefix + "ffn.fc1.kernel"] + P[prefix + "ffn.fc1.bias"])          # :69-71
    m = m @ P[prefix + "ffn.fc2.kernel"] + P[prefix + "ffn.fc2.bias"]                        # :72
    return m + h, probs                                                                     # :145-146


def vision_transformer_forward(x: Tensor, P: Params, radix: int = 3, kpaths: int = 3,
                               use_vit: bool = False, num_vit_layers: int = 8,
                               as_executed: bool = True, norm: str = "ln") -> Tensor:
    """VisionTransformer.forward - VisionTransformer.py:220-223 (Embeddings :112-120, Transformer :183-186).

    ``use_vit=False`` is BASELINE config 2 ("Arch B, no ViT"): the hidden state fed to the decoder is
    the patch embedding itself.  ``use_vit=True`` inserts Encoder.forward (:164-170).
    The position "embedding" is a constant zero tensor (:108) and dropout rates are 0 (:10,:61,:85).
    """
    B, H, W, _ = x.shape
    x4, feats = resnest_forward(x, P, radix, kpaths, "transformer.embeddings.hybrid_model.", as_executed, norm=norm)
    e = conv2d_same(x4, P["transformer.embeddings.patch_embeddings.kernel"],
                    P["transformer.embeddings.patch_embeddings.bias"])                      # :114
    gh, gw = H // 16, W // 16
    hidden = e.reshape(B, gh * gw, e.shape[-1])                                             # :116 (+0, :118)
    if use_vit:
        for l in range(num_vit_layers):
            hidden, _ = vit_block(hidden, P, f"transformer.encoder.Transformer_layers.{l}.")
        hidden = layer_norm(hidden, P["transformer.encoder.encoder_norm.gamma"],
                            P["transformer.encoder.encoder_norm.beta"], KERAS["vit_ln_eps"])  # :169
    return decoder_cup(hidden, feats, P, (gh, gw), "decoder.", norm)                        # :222


def cce_label_smoothing(y_true: Tensor, probs: Tensor, logits: Optional[Tensor] = None) -> Tensor:
    """CategoricalCrossentropy(label_smoothing=0.1, reduction=NONE) on probabilities -> [B,H,W].

    VisionTransformer.py:205; Appendix A.6: y <- y*(1-ls) + ls/C; p <- p/sum p; clip; -sum y log p.
    """
    ls = KERAS["cce_label_smoothing"]
    C = y_true.shape[-1]
    y = y_true * (1.0 - ls) + ls / C
    if KERAS["cce_cached_logits"]:      # the `_keras_logits` reading: -sum y * log_softmax(logits), no clip
        lp = torch.log_softmax(logits, dim=-1) if logits is not None else torch.log(probs)
        return -(y * lp).sum(dim=-1)
    p = probs / probs.sum(dim=-1, keepdim=True)
    eps = KERAS["cce_clip"]
    p = torch.clamp(p, eps, 1.0 - eps)
    return -(y * torch.log(p)).sum(dim=-1)


def compute_loss(y_true: Tensor, probs: Tensor, global_batch_size: int) -> Tensor:
    """VisionTransformer.compute_loss - VisionTransformer.py:225-227: sum(per-pixel loss)/global_batch."""
    return cce_label_smoothing(y_true, probs).sum() / global_batch_size


def clip_by_global_norm(grads: Sequence[Tensor], clip_norm: Optional[float] = None):
    """tf.clip_by_global_norm - VisionTransformer.py:244 (A.6): g * clip / max(norm, clip)."""
    clip_norm = KERAS["clip_norm"] if clip_norm is None else clip_norm
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    scale = clip_norm / torch.clamp(gn, min=clip_norm)
    return [g * scale for g in grads], gn


def adam_step(params: Sequence[Tensor], grads: Sequence[Tensor], m: List[Tensor], v: List[Tensor],
              step: int, lr: float):
    """tf.optimizers.Adam.apply_gradients - VisionTransformer.py:204,245; TBI_ResNest.py:28,46 (A.6).

    Keras form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps), eps 1e-7.
    ``step`` is the 1-based iteration count.  In-place on params/m/v.
    """
    b1, b2, eps = KERAS["adam_beta1"], KERAS["adam_beta2"], KERAS["adam_eps"]
    lr_t = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    for p, g, mi, vi in zip(params, grads, m, v):
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.sub_(lr_t * mi / (vi.sqrt() + eps))


def trainable_names(P: Params) -> List[str]:
    """Names of trainable variables (everything but BN moving statistics), in dict order."""
    return [k for k in P if not (k.endswith(".moving_mean") or k.endswith(".moving_variance"))]


def train_step(x: Tensor, y: Tensor, P: Params, opt_state: dict, global_batch_size: int, lr: float = 1e-3,
               radix: int = 3, kpaths: int = 3, use_vit: bool = False, as_executed: bool = False,
               grad_allreduce=None, transunet: bool = False):
    """VisionTransformer.train_step - VisionTransformer.py:235-246.

    forward -> loss (sum / GLOBAL batch) -> gradients -> clip_by_global_norm(1.0) -> Adam.
    ``grad_allreduce`` (optional callable on the list of clipped grads) models the
    MirroredStrategy SUM all-reduce that happens inside apply_gradients, i.e. AFTER the
    per-replica clip (MainParallel.py:130).
    Returns (loss, probs, grads_before_clip dict).  P is updated in place.
    """
    names = trainable_names(P)
    leaves = [P[n].detach().clone().requires_grad_(True) for n in names]
    Pl = dict(P)
    Pl.update(dict(zip(names, leaves)))
    probs = vision_transformer_forward(x, Pl, radix, kpaths, use_vit, as_executed=as_executed, norm="bn" if transunet else "ln")
    if transunet:   # TBI_TransUNet.py:546,583: CategoricalCrossentropy(label_smoothing=0.1) with the DEFAULT reduction = mean over B*H*W
        loss = cce_label_smoothing(y, probs).mean()
    else:
        loss = compute_loss(y, probs, global_batch_size)
    grads = torch.autograd.grad(loss, leaves)
    clipped, gnorm = clip_by_global_norm(grads)
    if grad_allreduce is not None:
        clipped = grad_allreduce(clipped)
    if "m" not in opt_state:
        opt_state["m"] = [torch.zeros_like(l) for l in leaves]
        opt_state["v"] = [torch.zeros_like(l) for l in leaves]
        opt_state["step"] = 0
    opt_state["step"] += 1
    with torch.no_grad():
        new = [l.detach().clone() for l in leaves]
        adam_step(new, clipped, opt_state["m"], opt_state["v"], opt_state["step"], lr)
        for n, t in zip(names, new):
            P[n] = t
    return loss.detach(), probs.detach(), dict(zip(names, [g.detach() for g in grads])), gnorm.detach()


# ----------------------------------------------------------------------------
# Arch A: TBI_ResNest.py (functional Keras model; separate weights per radix branch)
# ----------------------------------------------------------------------------
def archA_split_attention(inputs: Sequence[Tensor], P: Params, prefix: str) -> Tensor:
    """TBI_ResNest.py:175-207: dense1+BN+ELU, one dense2 PER radix branch, softmax over channels."""
    global STORAGE_DTYPE
    radix = len(inputs)
    holder = inputs[0]
    for t in inputs[1:]:
        holder = holder + t                                                                  # :179-184
    g = holder.mean(dim=(1, 2))[:, None, None, :]                                            # :186-187
    saved_storage, STORAGE_DTYPE = STORAGE_DTYPE, None                                       # the product runs this tiny MLP in fp32
    a = conv2d_same(g, P[prefix + "1.kernel"], P[prefix + "1.bias"])                         # :189
    a = elu(_bn(a, P, prefix + "_bn"))                                                       # :190-191
    out = None
    for r in range(radix):
        z = conv2d_same(a, P[f"{prefix}2_r{r}.kernel"], P[f"{prefix}2_r{r}.bias"])           # :195
        z = torch.sigmoid(z) if radix == 1 else softmax_lastaxis(z)                          # :197-200
        out = inputs[r] * z if out is None else out + inputs[r] * z                          # :202-205
    STORAGE_DTYPE = saved_storage
    return _q(out)


def archA_cardinal(x: Tensor, P: Params, prefix: str, radix: int) -> Tensor:
    """TBI_ResNest.py:153-173: per radix branch NEW conv1x1+BN+ELU, conv3x3+BN+ELU."""
    inputs = []
    for r in range(radix):
        y = conv2d_same(x, P[f"{prefix}1_r{r}.kernel"], P[f"{prefix}1_r{r}.bias"])           # :162
        y = elu(_bn(y, P, f"{prefix}1_r{r}bn"))                                             # :164-165
        y = conv2d_same(y, P[f"{prefix}2_r{r}.kernel"], P[f"{prefix}2_r{r}.bias"])           # :167
        inputs.append(elu(_bn(y, P, f"{prefix}2_r{r}bn")))                                   # :169-170
    return archA_split_attention(inputs, P, prefix + "_att")                                 # :173


def archA_residual_S(x: Tensor, P: Params, name: str, radix: int, kpaths: int) -> Tensor:
    """TBI_ResNest.py:130-151: shortcut conv only when channel counts differ (:142)."""
    cards = [archA_cardinal(x, P, f"{name}_car_k{k}", radix) for k in range(kpaths)]
    c2 = conv2d_same(torch.cat(cards, dim=3), P[name + "_concats_2.kernel"], P[name + "_concats_2.bias"])  # :140
    if x.shape[-1] != c2.shape[-1]:                                                          # :142
        sc = conv2d_same(x, P[name + "_cc.kernel"], P[name + "_cc.bias"])                    # :143
        x = elu(_bn(sc, P, name + "_scbn"))                                                  # :144-145
    return _q(x + c2)                                                                        # :148


def archA_upsample(x: Tensor, P: Params, name: str, dropout_mask: Optional[Tensor]) -> Tensor:
    """TBI_ResNest.py:209-220: tconv4x4s2 + BN + (tf.nn.dropout(0.5), ALWAYS on) + ReLU.

    tf.nn.dropout(out, 0.5) keeps with prob 0.5 and scales kept values by 2; the mask is
    injected (``dropout_mask`` in {0,1}) so the oracle is deterministic.
    """
    out = conv2d_transpose_s2_same(x, P[name + "_t_conv.kernel"], P[name + "_t_conv.bias"])  # :210
    out = _bn(out, P, name + "_bn")                                                         # :213
    if dropout_mask is not None:
        out = out * dropout_mask * 2.0                                                       # :216
    return _q(torch.relu(out))                                                               # :218


def archA_forward(x: Tensor, P: Params, radix: int = 3, kpaths: int = 4,
                  dropout_masks: Optional[Sequence[Optional[Tensor]]] = None) -> Tensor:
    """ResNest.model - TBI_ResNest.py:80-128.  Returns class probabilities [B,H,W,num_class]."""
    dm = list(dropout_masks) if dropout_masks is not None else [None, None, None]
    c = elu(conv2d_same(x, P["Conv1.kernel"], P["Conv1.bias"]))                              # :83-84
    c = elu(conv2d_same(c, P["conv2_1_1.kernel"], P["conv2_1_1.bias"]))                      # :85,:87
    c = conv2d_same(c, P["conv2_1_2.kernel"], P["conv2_1_2.bias"])                           # :88
    c = elu(_bn(c, P, "conv2_1_2bn"))                                                        # :90-91
    pool1 = avg_pool2(c)                                                                     # :92
    pool2 = avg_pool2(archA_residual_S(pool1, P, "conv2_1", radix, kpaths))                  # :93-95
    pool3 = avg_pool2(archA_residual_S(pool2, P, "conv2_2", radix, kpaths))                  # :96-98
    pool4 = avg_pool2(archA_residual_S(pool3, P, "conv3_1", radix, kpaths))                  # :99-101
    pool5 = avg_pool2(archA_residual_S(pool4, P, "conv3_2", radix, kpaths))                  # :102-104
    pool6 = avg_pool2(archA_residual_S(pool5, P, "conv4_1", radix, kpaths))                  # :105-107
    u = torch.cat([archA_upsample(pool6, P, "upsample_0", dm[0]), pool5], dim=3)             # :109-110
    u = torch.cat([archA_upsample(u, P, "upsample_1", dm[1]), pool4], dim=3)                 # :112-113
    u = torch.cat([archA_upsample(u, P, "upsample_2", dm[2]), pool3], dim=3)                 # :115-116
    u = torch.cat([archA_upsample(u, P, "upsample_3", None), pool2], dim=3)                  # :118-119
    u = torch.cat([archA_upsample(u, P, "upsample_4", None), pool1], dim=3)                  # :121-122
    logits = conv2d_transpose_s2_same(u, P["f_tran.kernel"], P["f_tran.bias"])               # :124
    return softmax_lastaxis(logits)                                                          # :125


def my_loss_cat(y_true: Tensor, y_pred: Tensor, height: int, width: int) -> Tensor:
    """ResNest.my_loss_cat - TBI_ResNest.py:234-248.  Returns an [H,W] map (NOT a scalar).

    per class c: scale[h,w] = 1/(sum_b y[b,h,w,c] + 1)/(H*W); CE += sum_b y*log(p+1e-7) * scale.
    tape.gradient of this non-scalar = gradient of its sum (A.6).
    """
    CE = 0
    for c in range(3):                                                                       # :239
        scale = 1.0 / (y_true[..., c].sum(dim=0) + 1.0) / (height * width)                   # :240-241
        CE = CE + (y_true[..., c] * torch.log(y_pred[..., c] + 1e-7)).sum(dim=0) * scale     # :244-245
    return -CE                                                                               # :246


# ----------------------------------------------------------------------------
# parameter construction (Keras initialisers, A.5) - for end-to-end runs and fixtures
# ----------------------------------------------------------------------------
def _he_normal(gen: torch.Generator, shape, fan_in: int, dtype) -> Tensor:
    """HeNormal = truncated normal (+-2 sigma) with variance 2/fan_in (Keras VarianceScaling)."""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    t = torch.empty(shape, dtype=torch.float64)
    torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
    return (t * std).to(dtype)


def _glorot_uniform(gen: torch.Generator, shape, fan_in: int, fan_out: int, dtype) -> Tensor:
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


class _Builder:
    def __init__(self, seed: int, dtype):
        self.P: Params = OrderedDict()
        self.gen = torch.Generator().manual_seed(seed)
        self.dtype = dtype

    def conv(self, name, k, cin, cout, he=True, transpose=False, rand_bias=False):
        shape = (k, k, cout, cin) if transpose else (k, k, cin, cout)
        if transpose:   # Keras fans for Conv2DTranspose kernels [kh,kw,out,in]: fan_in = k*k*out (shape[-2])
            fan_in, fan_out = k * k * cout, k * k * cin
        else:
            fan_in, fan_out = k * k * cin, k * k * cout
        w = _he_normal(self.gen, shape, fan_in, self.dtype) if he else \
            _glorot_uniform(self.gen, shape, fan_in, fan_out, self.dtype)
        self.P[name + ".kernel"] = w
        b = torch.zeros(cout, dtype=self.dtype)
        if rand_bias:
            b = (torch.randn(cout, generator=self.gen, dtype=torch.float64) * 0.1).to(self.dtype)
        self.P[name + ".bias"] = b

    def dense(self, name, cin, cout, rand_bias=False):
        self.P[name + ".kernel"] = _glorot_uniform(self.gen, (cin, cout), cin, cout, self.dtype)
        b = torch.zeros(cout, dtype=self.dtype)
        if rand_bias:
            b = (torch.randn(cout, generator=self.gen, dtype=torch.float64) * 0.1).to(self.dtype)
        self.P[name + ".bias"] = b

    def norm(self, name, c, bn=False, perturb=False):
        g = torch.ones(c, dtype=self.dtype)
        b = torch.zeros(c, dtype=self.dtype)
        if perturb:
            g = (1.0 + 0.2 * torch.randn(c, generator=self.gen, dtype=torch.float64)).to(self.dtype)
            b = (0.1 * torch.randn(c, generator=self.gen, dtype=torch.float64)).to(self.dtype)
        self.P[name + ".gamma"], self.P[name + ".beta"] = g, b
        if bn:
            mm = torch.zeros(c, dtype=self.dtype)
            mv = torch.ones(c, dtype=self.dtype)
            if perturb:
                mm = (0.1 * torch.randn(c, generator=self.gen, dtype=torch.float64)).to(self.dtype)
                mv = (1.0 + 0.3 * torch.rand(c, generator=self.gen, dtype=torch.float64)).to(self.dtype)
            self.P[name + ".moving_mean"], self.P[name + ".moving_variance"] = mm, mv


def init_resnest_params(channel: int, radix: int, kpaths: int, ksize: int = 3, seed: int = 0,
                        dtype=torch.float64, prefix: str = "", perturb: bool = False,
                        builder: Optional[_Builder] = None, widths=(64, 128, 256, 512), bn: bool = False) -> Params:
    """Parameters of ResNest(height,width,channel,ksize,radix,kpaths) - ResNest.py:7-36 (HeNormal everywhere).

    ``perturb=True`` randomises biases / norm affine parameters / BN moving statistics so that
    parity tests exercise every term (Keras initialises them to 0/1, which would hide bugs).
    """
    bld = builder or _Builder(seed, dtype)
    p = prefix
    bld.conv(p + "conv1", 3, channel, 16, rand_bias=perturb)
    bld.conv(p + "convtmp_1", 3, 16, 32, rand_bias=perturb)
    bld.norm(p + "convtmp_1bn", 32, bn=True, perturb=perturb)
    bld.conv(p + "convtmp_2", 3, 32, 32, rand_bias=perturb)
    bld.norm(p + "convtmp_2bn", 32, bn=True, perturb=perturb)
    cin = 32
    for s, oc in enumerate(widths, start=1):   # TBI_TransUNet.py:368: conv_4 has 256 output channels
        sp = f"{p}conv_{s}."
        cv11, cvkk, hid = cardinal_channels(oc, radix, kpaths)
        for k in range(kpaths):
            cp = f"{sp}cardinal_blocks.{k}."
            bld.conv(cp + "conv1", 1, cin, cv11, rand_bias=perturb)
            bld.norm(cp + "conv1_bn", cv11, bn=bn, perturb=perturb)
            bld.conv(cp + "conv2", ksize, cv11, cvkk, rand_bias=perturb)
            bld.norm(cp + "conv2_bn", cvkk, bn=bn, perturb=perturb)
            bld.conv(cp + "split.dense1", 1, cvkk, hid, rand_bias=perturb)
            bld.norm(cp + "split.dense1_bn", hid, bn=bn, perturb=perturb)
            bld.conv(cp + "split.dense2", 1, hid, cvkk, rand_bias=perturb)
        bld.conv(sp + "concats_2", ksize, kpaths * cvkk, oc, rand_bias=perturb)
        bld.conv(sp + "convtmp_sc", 1, cin, oc, rand_bias=perturb)
        bld.norm(sp + "convtmp_scbn", oc, bn=bn, perturb=perturb)
        cin = oc
    return bld.P


def init_decoder_params(num_classes: int = 3, hidden: int = 512, seed: int = 0, dtype=torch.float64,
                        prefix: str = "", perturb: bool = False, builder: Optional[_Builder] = None, bn: bool = False) -> Params:
    """Parameters of DecoderCup(num_classes) - Decoder.py:99-122 with DecoderBlock :8-59."""
    bld = builder or _Builder(seed, dtype)
    p = prefix
    bld.conv(p + "conv_more", 3, hidden, 256, rand_bias=perturb)
    bld.norm(p + "bn1", 256, bn=bn, perturb=perturb)
    cin = 256
    for i, oc in enumerate((256, 128, 64)):
        bp = f"{p}blocks.{i}."
        bld.conv(bp + "up", 3, cin, oc, transpose=True, rand_bias=perturb)
        c1 = oc + oc  # concat with the skip feature (x_3/x_2/x_1 have oc channels)
        for stage, ci in (("1", c1), ("2", oc)):
            for j in range(4):
                bld.conv(f"{bp}conv{stage}_{j}", 1 if j == 0 else 3, ci, oc // 4, rand_bias=perturb)
                bld.norm(f"{bp}bn{stage}_{j}", oc // 4, bn=True, perturb=perturb)
        cin = oc + hidden // (4 ** (i + 1))
    bld.conv(p + "head", 3, cin, num_classes, transpose=True, rand_bias=perturb)
    return bld.P


def init_vit_params(bld: _Builder, prefix: str, hidden: int = 512, mlp: int = 2048, layers: int = 8,
                    perturb: bool = False):
    for l in range(layers):
        lp = f"{prefix}Transformer_layers.{l}."
        bld.norm(lp + "attention_norm", hidden, perturb=perturb)
        for n in ("query", "key", "value", "out"):
            bld.dense(lp + "attn." + n, hidden, hidden, rand_bias=perturb)
        bld.norm(lp + "ffn_norm", hidden, perturb=perturb)
        bld.dense(lp + "ffn.fc1", hidden, mlp, rand_bias=perturb)
        bld.dense(lp + "ffn.fc2", mlp, hidden, rand_bias=perturb)
    bld.norm(prefix + "encoder_norm", hidden, perturb=perturb)


def init_vision_transformer_params(channel: int = 10, num_classes: int = 3, radix: int = 3, kpaths: int = 3,
                                   use_vit: bool = False, seed: int = 0, dtype=torch.float64,
                                   perturb: bool = False, transunet: bool = False) -> Params:
    """Parameters of VisionTransformer(...) - VisionTransformer.py:193-210 (ResNest radix=3,kpaths=3,ksize=3, :100).
    ``transunet=True``: the older self-contained copy TBI_TransUNet.py - BatchNormalization where ResNest.py / Decoder.py use
    LayerNormalization (:304,426,465,472,503) and a 256-channel stage 4 (:368), hence a 256 -> 512 patch embedding (:110)."""
    bld = _Builder(seed, dtype)
    widths = (64, 128, 256, 256) if transunet else (64, 128, 256, 512)
    init_resnest_params(channel, radix, kpaths, 3, prefix="transformer.embeddings.hybrid_model.",
                        perturb=perturb, builder=bld, widths=widths, bn=transunet)
    bld.conv("transformer.embeddings.patch_embeddings", 1, widths[3], 512, he=False, rand_bias=perturb)  # :106 (Glorot)
    if use_vit:
        init_vit_params(bld, "transformer.encoder.", perturb=perturb)
    init_decoder_params(num_classes, 512, prefix="decoder.", perturb=perturb, builder=bld, bn=transunet)
    return bld.P


def init_archA_params(channel: int = 1, num_class: int = 3, radix: int = 3, kpaths: int = 4, ksize: int = 3,
                      seed: int = 0, dtype=torch.float64, perturb: bool = False) -> Params:
    """Parameters of TBI_ResNest.ResNest.model() - TBI_ResNest.py:80-220 (all Glorot-uniform, A.5)."""
    bld = _Builder(seed, dtype)
    g = dict(he=False, rand_bias=perturb)
    bld.conv("Conv1", 3, channel, 16, **g)
    bld.conv("conv2_1_1", 3, 16, 32, **g)
    bld.conv("conv2_1_2", 3, 32, 32, **g)
    bld.norm("conv2_1_2bn", 32, bn=True, perturb=perturb)
    cin = 32
    for name, oc in (("conv2_1", 64), ("conv2_2", 128), ("conv3_1", 256), ("conv3_2", 512), ("conv4_1", 512)):
        cv11, cvkk, hid = cardinal_channels(oc, radix, kpaths)
        for k in range(kpaths):
            cp = f"{name}_car_k{k}"
            for r in range(radix):
                bld.conv(f"{cp}1_r{r}", 1, cin, cv11, **g)
                bld.norm(f"{cp}1_r{r}bn", cv11, bn=True, perturb=perturb)
                bld.conv(f"{cp}2_r{r}", ksize, cv11, cvkk, **g)
                bld.norm(f"{cp}2_r{r}bn", cvkk, bn=True, perturb=perturb)
            bld.conv(f"{cp}_att1", 1, cvkk, hid, **g)
            bld.norm(f"{cp}_att_bn", hid, bn=True, perturb=perturb)
            for r in range(radix):
                bld.conv(f"{cp}_att2_r{r}", 1, hid, cvkk, **g)
        bld.conv(name + "_concats_2", ksize, kpaths * cvkk, oc, **g)
        if cin != oc:
            bld.conv(name + "_cc", 1, cin, oc, **g)
            bld.norm(name + "_scbn", oc, bn=True, perturb=perturb)
        cin = oc
    ups = (("upsample_0", 512, 512, 512), ("upsample_1", 1024, 512, 256), ("upsample_2", 768, 512, 128),
           ("upsample_3", 640, 256, 64), ("upsample_4", 320, 128, 32))
    for name, ci, co, _skip in ups:
        bld.conv(name + "_t_conv", 4, ci, co, transpose=True, **g)
        bld.norm(name + "_bn", co, bn=True, perturb=perturb)
    bld.conv("f_tran", 4, 160, num_class, transpose=True, **g)
    return bld.P


# ----------------------------------------------------------------------------
# synthetic data
# ----------------------------------------------------------------------------
def synthetic_batch(batch: int, height: int, width: int, channel: int, num_classes: int = 3, seed: int = 0,
                    dtype=torch.float64):
    """x ~ N(0,1) clipped to [-1,1]; y = soft 3-class maps by the label2vec rule (Dataset_2.py:6-20)
    from a blocky label image with values 0/1/2 drawn 70/25/5 % in 16x16 blocks, softened at
    block level with a seeded uniform in [0, 0.9) added to the label so c1/c2 are not one-hot."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, height, width, channel, generator=gen, dtype=torch.float64).clamp_(-1, 1)
    bh, bw = max(height // 16, 1), max(width // 16, 1)
    u = torch.rand(batch, bh, bw, generator=gen, dtype=torch.float64)
    lab = torch.zeros_like(u)
    lab[u > 0.70] = 1.0
    lab[u > 0.95] = 2.0
    lab = lab + 0.9 * torch.rand(batch, bh, bw, generator=gen, dtype=torch.float64) * (lab >= 1.0)
    lab = lab.repeat_interleave(height // bh, 1).repeat_interleave(width // bw, 2)
    y = label2vec(lab, num_classes)
    return x.to(dtype), y.to(dtype)


def label2vec(label: Tensor, num_classes: int = 3) -> Tensor:
    """Dataset_2.py:6-20: c2 = clip(l-1,0,1) where l>=1.05; c1 = 1-c2 where l>0.95; c0 = 1 where l<=0.95."""
    out = torch.zeros(*label.shape, num_classes, dtype=label.dtype)
    c2 = torch.where(label >= 1.05, (label - 1.0).clamp(0, 1), torch.zeros_like(label))
    c1 = torch.where(label > 0.95, 1.0 - c2, torch.zeros_like(label))
    c0 = torch.where(label <= 0.95, torch.ones_like(label), torch.zeros_like(label))
    out[..., 0], out[..., 1], out[..., 2] = c0, c1, c2
    return out
