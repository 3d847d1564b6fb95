"""CPU oracle (fp64 PyTorch) of the reference's ``SwinTransformer.py`` (a port of rishigami/Swin-Transformer-TF): windowed
attention encoder - BASELINE configs[4], SURVEY.md section 8f rank 4.  TEST INFRASTRUCTURE ONLY: tests/ and bench.py's CPU leg may
import it, the product never does.  **Parity unpinned**: the reference holds no fixtures for this file, nothing imports it
(only a commented line, VisionTransformer.py:101-102), its factory needs a download (:469-471) and TensorFlow is absent; the
restatement follows the class definitions line by line and is pinned by the known-answer tests in tests/test_swin_oracle.py.

Semantics as the reference would execute them (every file:line is SwinTransformer.py):
  * Keras layers called without ``training`` -> Dropout / DropPath are the identity (:25,144-155,186,258-259);
  * LayerNormalization epsilon 1e-5 (:180,187,270,343,419), exact GELU (:36), Dense = x @ kernel + bias;
  * window attention (:60-141): q scaled by head_dim**-0.5, learned relative-position bias table indexed by
    relative_position_index (:84-99), -100 mask between different regions of a cyclically shifted image (:192-217);
  * ``window_reverse`` (:53-58) mixes window_size[0] / [1] and is only a true inverse for SQUARE windows - all CFGS (:8-21) are
    square; non-square windows (the constructor default [4, 5], :376) are rejected here instead of reproducing a scrambled
    tensor;
  * PatchMerging (:264-290): x0=[0::2,0::2], x1=[1::2,0::2], x2=[0::2,1::2], x3=[1::2,1::2] concatenated, LN(4C), Dense(2C, no bias);
  * the model returns (pooled features [B, num_features] (or logits with include_top), [stage outputs before each PatchMerging])
    (:437-455).  ``self.features`` accumulates across calls in the reference (:434,449); one call is modelled.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-5


def layer_norm(x: Tensor, g: Tensor, b: Tensor) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * g + b


def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def window_partition(x: Tensor, ws: int) -> Tensor:
    B, H, W, C = x.shape                                                    # :40-48
    x = x.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C)


def window_reverse(windows: Tensor, ws: int, H: int, W: int) -> Tensor:
    C = windows.shape[-1]                                                   # :51-58 (square windows)
    x = windows.reshape(-1, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, H, W, C)


def relative_position_index(ws: int) -> Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).reshape(2, -1)   # :84-87
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).clone()                                   # :88-89
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)                                                                                         # :93


def shift_mask(H: int, W: int, ws: int, shift: int) -> Tensor:
    """attn_mask [nW, N, N] of a cyclically shifted image (:192-217): 0 inside a region, -100 across regions."""
    img = torch.zeros(1, H, W, 1, dtype=torch.float64)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).reshape(-1, ws * ws)
    m = mw[:, None, :] - mw[:, :, None]
    return torch.where(m != 0, torch.full_like(m, -100.0), torch.zeros_like(m))


def window_attention(x: Tensor, P: Dict[str, Tensor], pre: str, ws: int, heads: int, mask) -> Tensor:
    B_, N, C = x.shape                                                      # :101-141
    d = C // heads
    qkv = (x @ P[pre + "attn/qkv/kernel"] + P[pre + "attn/qkv/bias"]).reshape(B_, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-1, -2)
    idx = relative_position_index(ws).reshape(-1)
    bias = P[pre + "attn/relative_position_bias_table"][idx].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(-1, nW, heads, N, N) + mask[None, :, None]).reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return out @ P[pre + "attn/proj/kernel"] + P[pre + "attn/proj/bias"]


def swin_block(x: Tensor, P, pre: str, res: Tuple[int, int], ws: int, shift: int, heads: int) -> Tensor:
    H, W = res                                                              # :219-261
    B, L, C = x.shape
    shortcut = x
    x = layer_norm(x, P[pre + "norm1/gamma"], P[pre + "norm1/beta"]).reshape(B, H, W, C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).reshape(-1, ws * ws, C)
    aw = window_attention(xw, P, pre, ws, heads, shift_mask(H, W, ws, shift) if shift > 0 else None)
    x = window_reverse(aw.reshape(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + x.reshape(B, H * W, C)
    h = layer_norm(x, P[pre + "norm2/gamma"], P[pre + "norm2/beta"])
    h = gelu(h @ P[pre + "mlp/fc1/kernel"] + P[pre + "mlp/fc1/bias"]) @ P[pre + "mlp/fc2/kernel"] + P[pre + "mlp/fc2/bias"]
    return x + h


def patch_merging(x: Tensor, P, pre: str, res: Tuple[int, int]) -> Tensor:
    H, W = res                                                              # :272-290
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1).reshape(B, (H // 2) * (W // 2), 4 * C)
    return layer_norm(x, P[pre + "downsample/norm/gamma"], P[pre + "downsample/norm/beta"]) @ P[pre + "downsample/reduction/kernel"]


def block_window(res: Tuple[int, int], ws: int, i: int) -> Tuple[int, int]:
    """(window, shift) of block i of a stage: shift = 0 / ws//2 alternating (:312-313); a stage whose resolution does not exceed the
    window uses ONE window of min(resolution) tokens per side and no shift (:172-175)."""
    shift = 0 if i % 2 == 0 else ws // 2
    if min(res) <= ws:
        return min(res), 0
    return ws, shift


def swin_forward(x: Tensor, P, cfg: dict):
    """SwinTransformerModel.call (:437-455): -> (pooled [B, F] or logits, [feature of every stage but the last])."""
    ps, E, depths, heads, ws = cfg["patch_size"], cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window_size"]
    B, H, W, Cin = x.shape
    w = P["patch_embed/proj/kernel"]                                        # [ps, ps, Cin, E], stride ps (:352-353)
    t = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), P["patch_embed/proj/bias"], stride=ps).permute(0, 2, 3, 1)
    res = (H // ps, W // ps)
    t = t.reshape(B, res[0] * res[1], E)
    t = layer_norm(t, P["patch_embed/norm/gamma"], P["patch_embed/norm/beta"])   # patch_norm=True (:380,398)
    if cfg.get("ape"):
        t = t + P["absolute_pos_embed"]                                     # [1, L, E] (:401-408,442-443)
    feats = []
    for il, depth in enumerate(depths):
        pre = f"layers{il}/"
        for i in range(depth):
            wsz, shift = block_window(res, ws, i)
            t = swin_block(t, P, f"{pre}blocks{i}/", res, wsz, shift, heads[il])
        if il < len(depths) - 1:
            feats.append(t)                                                 # :336-338,448-449
            t = patch_merging(t, P, pre, res)
            res = (res[0] // 2, res[1] // 2)
    t = layer_norm(t, P["norm/gamma"], P["norm/beta"]).mean(dim=1)          # :450-451
    if cfg.get("include_top"):
        t = t @ P["head/kernel"] + P["head/bias"]
    return t, feats


def init_swin_params(cfg: dict, in_chans: int = 1, seed: int = 0, dtype=torch.float64, perturb: bool = True) -> "OrderedDict[str, Tensor]":
    """Keras initialisers (Glorot-uniform kernels, zero biases, zero bias table, unit LN); ``perturb`` randomises the zero / unit
    ones so that every term is exercised."""
    g = torch.Generator().manual_seed(seed)
    P: "OrderedDict[str, Tensor]" = OrderedDict()
    ps, E, depths, heads, ws = cfg["patch_size"], cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window_size"]

    def glorot(name, *shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        P[name] = ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype)

    def vec(name, n, base, sc):
        P[name] = (base + (sc * torch.randn(n, generator=g, dtype=torch.float64) if perturb else 0.0) * torch.ones(n, dtype=torch.float64)).to(dtype)

    def ln(name, n):
        vec(name + "/gamma", n, 1.0, 0.2)
        vec(name + "/beta", n, 0.0, 0.1)

    def dense(name, i, o, bias=True):
        glorot(name + "/kernel", i, o, fan_in=i, fan_out=o)
        if bias:
            vec(name + "/bias", o, 0.0, 0.1)
    glorot("patch_embed/proj/kernel", ps, ps, in_chans, E, fan_in=ps * ps * in_chans, fan_out=ps * ps * E)
    vec("patch_embed/proj/bias", E, 0.0, 0.1)
    ln("patch_embed/norm", E)
    if cfg.get("ape"):       # Zeros initialiser (:403-407); perturbed so that the term is exercised
        res = cfg["img_size"][0] // ps, cfg["img_size"][1] // ps
        P["absolute_pos_embed"] = ((0.3 * torch.randn(1, res[0] * res[1], E, generator=g, dtype=torch.float64)) if perturb
                                   else torch.zeros(1, res[0] * res[1], E, dtype=torch.float64)).to(dtype)
    for il, depth in enumerate(depths):
        C = E * 2 ** il
        for i in range(depth):
            pre = f"layers{il}/blocks{i}/"
            ln(pre + "norm1", C)
            dense(pre + "attn/qkv", C, 3 * C)
            nb = (2 * ws - 1) ** 2
            P[pre + "attn/relative_position_bias_table"] = ((0.5 * torch.randn(nb, heads[il], generator=g, dtype=torch.float64)) if perturb
                                                            else torch.zeros(nb, heads[il], dtype=torch.float64)).to(dtype)
            dense(pre + "attn/proj", C, C)
            ln(pre + "norm2", C)
            dense(pre + "mlp/fc1", C, int(C * cfg.get("mlp_ratio", 4.0)))
            dense(pre + "mlp/fc2", int(C * cfg.get("mlp_ratio", 4.0)), C)
        if il < len(depths) - 1:
            ln(f"layers{il}/downsample/norm", 4 * C)
            dense(f"layers{il}/downsample/reduction", 4 * C, 2 * C, bias=False)
    F_ = E * 2 ** (len(depths) - 1)
    ln("norm", F_)
    if cfg.get("include_top"):
        dense("head", F_, cfg["num_classes"])
    return P
