"""Model wrapper and training step: the surface of the reference's ``VisionTransformer.py`` (:192-254).

``VisionTransformer(batch_size, img_size, num_classes, learning_rate, weight_decay)`` exposes ``forward``, ``step``,
``train_step`` and ``compute_loss`` exactly as the drivers call them (MainNumpy.py:45,95; MainParallel.py:130,158):
``train_step(x, y) -> (loss, probs)`` with x NHWC float32/float64 ``[B,H,W,C]`` and y float32 soft one-hot
``[B,H,W,classes]``.

Differences that are parameters here and literals in the reference: the input channel count (10 at
VisionTransformer.py:100,198) and the 16x5 token grid (:90) which becomes (H/16, W/16).
``use_vit=False`` (default) is BASELINE config 2 ("ResNeSt encoder + Decoder.py, no ViT"): the patch embedding feeds the
decoder directly.  ``use_vit=True`` inserts the 8-layer ViT bottleneck (:9-189), i.e. the model the reference's drivers
actually build; its variables carry the reference's attribute paths (``transformer.encoder.Transformer_layers.3.attn.query``).
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from . import Decoder as _DEC
from . import ResNest as _ENC
from .Decoder import DecoderBlock, DecoderCup
from .flat import AdamClip, FlatParams
from .layers import BatchNormalization, Conv2D, LayerNormalization, _Workspace
from .ResNest import ResNest, cardinal, residual_S

_FUSED_HEAD = os.environ.get("USSEG_FUSED_HEAD", "1") != "0"      # quad-form head conv + softmax + loss as one launch (csrc/attn_loss_optim.hip)
from .ops import BF16, roundup
from .step import TrainStepDriver

input_size = (256, 80)   # VisionTransformer.py:7
_VIT_LAZY = os.environ.get("USSEG_VIT_LAZY", "enc")              # ops.lazy_wgrads over the whole encoder ("enc"), per block, or "0"
_FUSED_ATTN = os.environ.get("USSEG_FUSED_ATTN", "1") != "0"   # train-step attention on csrc/flash_attn.hip (head size 128)


def repack_all(root: nn.Module):
    """Rebuild every bf16 packed operand from the fp32 master weights (after an optimiser step or a weight load):
    ONE batched launch driven by a device-resident job table that is built once (all pointers are static)."""
    table = getattr(root, "_pack_table", None)
    if table is None:
        jobs = []
        for m in root.modules():
            if isinstance(m, residual_S):
                jobs += m.pack_jobs()
            elif isinstance(m, cardinal):
                if m._solo is not None:
                    jobs += m._solo.pack_jobs()
            elif isinstance(m, (Attention, DecoderBlock, DecoderCup)):
                jobs += m.pack_jobs()
            elif isinstance(m, Conv2D) and m.wp_f is not None:
                jobs += m.pack_jobs()
        dev = next(root.parameters()).device
        folds = []
        for m in root.modules():
            if isinstance(m, (ResNest, DecoderBlock)):
                folds += m.bn_fold_jobs()
        table = (ops.make_pack_table(jobs, dev), len(jobs), ops.make_bn_fold_table(folds, dev) if folds else None, len(folds),
                 ops.make_pack_tilemap(jobs, dev))
        object.__setattr__(root, "_pack_table", table)
    if table[3] and (_DEC._FOLD_BN or _ENC._FOLD_BN):
        ops.bn_fold_batched(table[2], table[3])     # folded inference BatchNorm constants FIRST: the packs below multiply them in
    ops.pack_weights_batched(table[0], table[1], table[4])


class Embeddings(nn.Module):
    """VisionTransformer.py:81-124: ResNest hybrid model + 1x1 patch embedding (+ constant zero position term)."""

    def __init__(self, img_size, hidden_size=512, dropout_rate=0.0, wDecay=None, in_channels=10, transunet=False):
        super().__init__()
        self.img_size, self.hidden_size, self.wDecay = img_size, hidden_size, wDecay
        self.grid_size = (img_size[0] // 16, img_size[1] // 16)      # (16, 5) at 256x80 (:90)
        self.seq_len = self.grid_size[0] * self.grid_size[1]
        # :100; TBI_TransUNet.py:107 builds its own copy: BatchNormalization for LayerNormalization and a 256-channel conv_4 (:368)
        norm, widths = ("bn", (64, 128, 256, 256)) if transunet else ("ln", (64, 128, 256, 512))
        self.hybrid_model = ResNest(img_size[0], img_size[1], in_channels, radix=3, ksize=3, kpaths=3, norm=norm, widths=widths)
        self.hybrid_model.stage_lazy = _ENC._STAGE_LAZY           # (Transformer.__init__ clears it when a ViT follows the embedding)
        self.patch_embeddings = Conv2D(widths[3], hidden_size, 1, init="glorot")                          # :106

    def forward(self, x, feature_slots=None):
        """``feature_slots``: optional [x_3, x_2, x_1] destinations (the skip slices of the decoder's concat buffers): the
        encoder stages write their outputs there, so ``tf.concat([x, skip])`` (Decoder.py:66) costs no copy."""
        x4, features = self.hybrid_model.forward(x, outs=feature_slots)               # :113
        e = self.patch_embeddings.forward(x4)                                         # :114
        B = e.shape[0]
        return e.reshape(B, self.seq_len, self.hidden_size), features                 # :116 (+ zeros, :118; dropout 0)

    def backward(self, d_hidden, d_feats):
        B = d_hidden.shape[0]
        gh, gw = self.grid_size
        d_x4 = self.patch_embeddings.backward(d_hidden.reshape(B, gh, gw, self.hidden_size))
        self.hybrid_model.backward(d_x4, d_feats)


class Attention(nn.Module):
    """VisionTransformer.py:9-57: 4 heads of 128, scores divided by sqrt(num_heads) = 2 (NOT sqrt(head_dim), :42), softmax
    over keys, returns the attention weights.  Q/K/V projections run as ONE GEMM (N = 1536); the per-(image, head)
    products are batched GEMMs on the conv kernels with the head split expressed as a channel slice."""

    def __init__(self, num_heads=4, attention_head_size=512, attention_dropout_rate=0.0, wDecay=None):
        super().__init__()
        self.num_heads, self.hidden_size, self.wDecay = num_heads, attention_head_size, wDecay
        self.qkv_size = attention_head_size // num_heads
        hs = self.hidden_size
        self.query, self.key, self.value = (Conv2D(hs, hs, 1, init="glorot") for _ in range(3))     # Dense = 1x1 conv on [B,N,1,C]
        for c in (self.query, self.key, self.value):
            c.on_finalize = lambda device: None                                                       # packed together below
        self.out = Conv2D(hs, hs, 1, init="glorot")

    def adjacent_params(self):
        return [([self.query.bias, self.key.bias, self.value.bias], 0)]

    def on_finalize(self, device):
        hs = self.hidden_size
        self.w_f = torch.zeros((3 * hs, hs), dtype=BF16, device=device)
        self.w_d = torch.zeros((hs, 3 * hs), dtype=BF16, device=device)
        self.b_qkv = torch.as_strided(self.query.bias.data, (3 * hs,), (1,))
        self.db_qkv = torch.as_strided(self.query.bias.grad, (3 * hs,), (1,))
        assert self.key.bias.data_ptr() == self.query.bias.data_ptr() + 4 * hs
        jobs = self.pack_jobs()
        ops.pack_weights_batched(ops.make_pack_table(jobs, device), len(jobs))

    def pack_jobs(self):
        hs, jobs = self.hidden_size, []
        for i, c in enumerate((self.query, self.key, self.value)):       # Dense kernel [in, out]
            jobs.append(ops.pack_job(c.kernel.data, 0, 1, hs, 1, hs, hs, self.w_f, hs, hs, i * hs, 0))
            jobs.append(ops.pack_job(c.kernel.data, 0, hs, 1, 1, hs, hs, self.w_d, 3 * hs, 3 * hs, 0, i * hs))
        return jobs

    def _qkv_map(self):
        key = self.query.kernel.grad.data_ptr()
        if getattr(self, "_qkv_map_key", None) != key:
            hs = self.hidden_size
            self._qkv_wmap = ops.wgrad_dst([(c.kernel.grad, 0, hs, 1, 0, i * hs, hs, hs) for i, c in enumerate((self.query, self.key, self.value))])
            self._qkv_map_key = key
        return self._qkv_wmap

    def forward(self, xn, residual, need_weights=True):
        """xn, residual: [B,N,1,hidden] bf16 -> (attention output + residual, weights fp32 [B,heads,N,N]).
        ``need_weights=False`` (the train step, which drops them: :220-223,243) runs the fused kernels of csrc/flash_attn.hip - no
        [B,heads,N,N] tensor exists - and returns None for the weights."""
        B, N, _, hs = xn.shape
        nh, dh, dev = self.num_heads, self.qkv_size, xn.device
        qkv = ops.conv2d_fwd(xn, self.w_f, self.b_qkv, 1, 1, ops.new_act(B, N, 1, 3 * hs, dev))          # :34-36
        if not need_weights and dh == 128 and _FUSED_ATTN:
            ctx = ops.new_act(B, N, 1, hs, dev)
            lse = torch.empty((B * nh, N), dtype=torch.float32, device=dev)
            ctx32 = torch.empty((B, N, hs), dtype=torch.float32, device=dev)
            ops.flash_attn_fwd(qkv, nh, 1.0 / (float(nh) ** 0.5), ctx, lse, ctx32)                         # :41-49
            out = self.out.forward(ctx, residual=residual)
            self._saved = (xn, qkv, ctx, lse, ctx32)
            return out, None
        q, k, v = qkv[..., :hs], qkv[..., hs:2 * hs], qkv[..., 2 * hs:]
        S = torch.empty((B, nh, N, N), dtype=torch.float32, device=dev)
        ops.gemm_nt_batched(q, k, S, N, N, dh, 3 * hs, 3 * hs, N, B, nh, (N * 3 * hs, dh), (N * 3 * hs, dh), (nh * N * N, N * N), out_f32=True)  # :41
        P32, Pb = torch.empty_like(S), torch.empty((B, nh, N, N), dtype=BF16, device=dev)
        ops.softmax_rows_fwd(S, N, 1.0 / (float(nh) ** 0.5), P32, Pb)                                     # :42-43
        vt = torch.empty((B * nh, dh, N), dtype=BF16, device=dev)
        kt = torch.empty((B * nh, dh, N), dtype=BF16, device=dev)
        ops.transpose_batched(v, N, dh, 3 * hs, B, nh, (N * 3 * hs, dh), vt)
        ops.transpose_batched(k, N, dh, 3 * hs, B, nh, (N * 3 * hs, dh), kt)
        ctx = ops.new_act(B, N, 1, hs, dev)
        ops.gemm_nt_batched(Pb, vt, ctx, N, dh, N, N, N, hs, B, nh, (nh * N * N, N * N), (nh * dh * N, dh * N), (N * hs, dh))                 # :47-49
        out = self.out.forward(ctx, residual=residual)                                                    # :50 (+ h, :140)
        self._saved = (xn, qkv, P32, Pb, kt)
        return out, P32

    def _backward_tail(self, xn, dqkv):
        # fused projection backward: the [hidden, 3*hidden] gradient is scattered straight into the three Dense kernels
        # (one destination block each) - no shared scratch that a deferred split-K finish would still be filling
        B, N, _, hs = xn.shape
        def params():
            ops.conv2d_wgrad_mapped(xn, dqkv, 1, 1, self._qkv_map())
            ops.colsum(dqkv, self.db_qkv, 3 * hs)
        ops.wgrad_later(params, xn, dqkv)
        return ops.conv2d_dgrad(dqkv, self.w_d, 1, 1, ops.new_act(B, N, 1, hs, xn.device))

    def backward(self, d_out, out_bias_done=False):
        if len(self._saved) == 5 and self._saved[3].dim() == 2:      # fused forward: probabilities recomputed from the log-sum-exp
            xn, qkv, ctx, lse, ctx32 = self._saved
            B, N, _, hs = xn.shape
            dctx = self.out.backward(d_out, skip_bias=out_bias_done)
            dqkv = ops.new_act(B, N, 1, 3 * hs, xn.device)
            ops.flash_attn_bwd(qkv, self.num_heads, 1.0 / (float(self.num_heads) ** 0.5), ctx, dctx, lse, torch.empty_like(lse), dqkv, ctx32)
            return self._backward_tail(xn, dqkv)
        xn, qkv, P32, Pb, kt = self._saved
        B, N, _, hs = xn.shape
        nh, dh, dev = self.num_heads, self.qkv_size, xn.device
        q, k, v = qkv[..., :hs], qkv[..., hs:2 * hs], qkv[..., 2 * hs:]
        dctx = self.out.backward(d_out, skip_bias=out_bias_done)
        dqkv = ops.new_act(B, N, 1, 3 * hs, dev)
        s_pp, s_qkv, s_ctx, s_hd = (nh * N * N, N * N), (N * 3 * hs, dh), (N * hs, dh), (nh * N * dh, N * dh)
        dV = torch.zeros((B, nh, N, dh), dtype=torch.float32, device=dev)
        ops.gemm_tn_batched(Pb, dctx, dV, N, dh, N, N, hs, B, nh, s_pp, s_ctx, s_hd)                       # dV = P^T dO
        dP = torch.empty((B, nh, N, N), dtype=torch.float32, device=dev)
        ops.gemm_nt_batched(dctx, v, dP, N, N, dh, hs, 3 * hs, N, B, nh, s_ctx, s_qkv, s_pp, out_f32=True)  # dP = dO V^T
        dS = torch.empty((B, nh, N, N), dtype=BF16, device=dev)
        ops.softmax_rows_bwd(P32, dP, N, 1.0 / (float(nh) ** 0.5), dS)
        ops.gemm_nt_batched(dS, kt, dqkv, N, dh, N, N, N, 3 * hs, B, nh, s_pp, (nh * dh * N, dh * N), s_qkv)  # dQ = dS K
        dK = torch.zeros((B, nh, N, dh), dtype=torch.float32, device=dev)
        ops.gemm_tn_batched(dS, q, dK, N, dh, N, N, 3 * hs, B, nh, s_pp, s_qkv, s_hd)                       # dK = dS^T Q
        ops.cast_f32_to_bf16_batched(dK, N, dh, B, nh, dqkv[..., hs:2 * hs], 3 * hs, s_qkv)
        ops.cast_f32_to_bf16_batched(dV, N, dh, B, nh, dqkv[..., 2 * hs:], 3 * hs, s_qkv)
        return self._backward_tail(xn, dqkv)


class Mlp(nn.Module):
    """VisionTransformer.py:60-78: fc1 -> (dropout 0) -> exact GELU -> fc2."""

    def __init__(self, hidden_size=512, mlp_dim=2048, dropout_rate=0.0):
        super().__init__()
        self.fc1 = Conv2D(hidden_size, mlp_dim, 1, init="glorot")
        self.fc2 = Conv2D(mlp_dim, hidden_size, 1, init="glorot")

    def forward(self, x, residual):
        self._raw = self.fc1.forward(x)                                                                   # :69
        g = ops.act_fwd(self._raw, torch.empty_like(self._raw), ops.ACT_GELU, 0.0)                       # :71
        return self.fc2.forward(g, residual=residual)                                                     # :72 (+ h, :145)

    def backward(self, d, fc2_bias_done=False):
        dg = self.fc2.backward(d, skip_bias=fc2_bias_done)
        # GELU backward + fc1's bias gradient (column sums of its output) in one pass
        dr = ops.act_bwd_colsum(self._raw, dg, torch.empty_like(dg), ops.ACT_GELU, 0.0, self.fc1.bias.grad, self.fc1.cout)
        return self.fc1.backward(dr, skip_bias=True)


class Block(nn.Module):
    """VisionTransformer.py:127-150: pre-LN (eps 1e-6) attention and MLP with residuals."""

    def __init__(self, hidden_size=512, wDecay=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.attention_norm = LayerNormalization(hidden_size, epsilon=1e-6)
        self.ffn_norm = LayerNormalization(hidden_size, epsilon=1e-6)
        self.ffn = Mlp(hidden_size)
        self.attn = Attention(attention_head_size=hidden_size, wDecay=wDecay)

    def forward(self, x, need_weights=True):
        a, weights = self.attn.forward(self.attention_norm.forward(x), residual=x, need_weights=need_weights)   # :137-140
        return self.ffn.forward(self.ffn_norm.forward(a), residual=a), weights                            # :142-146

    def backward(self, d, fc2_bias_done=False, prev_fc2_bias=None):
        """``fc2_bias_done``: the kernel that produced d also summed it into this block's fc2 bias gradient; ``prev_fc2_bias``: the bias
        gradient of the fc2 whose output (+ shortcut) is this block's input - summed by the kernel that produces dx."""
        # + the residual branch of x = x + h (:145); the same pass sums da over the tokens = the output projection's bias gradient
        da = self.ffn_norm.backward_residual(self.ffn.backward(d, fc2_bias_done), d, dbias=self.attn.out.bias.grad)
        return self.attention_norm.backward_residual(self.attn.backward(da, out_bias_done=True), da, dbias=prev_fc2_bias)   # + the residual branch (:140)


class Encoder(nn.Module):
    """VisionTransformer.py:153-174."""

    def __init__(self, xdim, ydim, num_layers=8, wDecay=None):
        super().__init__()
        self.xDim, self.yDim = xdim, ydim
        self.encoder_norm = LayerNormalization(512, epsilon=1e-6)
        self.Transformer_layers = nn.ModuleList([Block(wDecay=wDecay) for _ in range(num_layers)])

    def forward(self, hidden_states, need_weights=True):
        attn_weights = []
        for blk in self.Transformer_layers:                                                               # :166-168
            hidden_states, w = blk.forward(hidden_states, need_weights)
            attn_weights.append(w)
        return self.encoder_norm.forward(hidden_states), attn_weights                                     # :169

    def backward(self, d):
        d = self.encoder_norm.backward(d)
        blocks, done = self.Transformer_layers, False
        for i in reversed(range(len(blocks))):
            prev = blocks[i - 1].ffn.fc2.bias.grad if i > 0 else None
            with ops.lazy_wgrads() if _VIT_LAZY == "block" else contextlib.nullcontext():
                d = blocks[i].backward(d, fc2_bias_done=done, prev_fc2_bias=prev)
            done = prev is not None
        return d


class Transformer(nn.Module):
    """VisionTransformer.py:177-189."""

    def __init__(self, img_size, wDecay=None, in_channels=10, use_vit=False, transunet=False):
        super().__init__()
        self.embeddings = Embeddings(img_size=img_size, in_channels=in_channels, transunet=transunet)
        self.encoder = Encoder(img_size[0], img_size[1], wDecay=wDecay) if use_vit else None
        if self.encoder is not None and _ENC._STAGE_LAZY_ENV is None:
            self.embeddings.hybrid_model.stage_lazy = 0       # the side stream carries the ViT's weight gradients (ResNest.py of this repo, _STAGE_LAZY)

    def forward(self, input_ids, feature_slots=None, need_weights=True):
        embedding_output, features = self.embeddings.forward(input_ids, feature_slots)
        if self.encoder is None:
            return embedding_output, [], features
        B, N, hs = embedding_output.shape
        encoded, attn_weights = self.encoder.forward(embedding_output.reshape(B, N, 1, hs), need_weights)   # :185
        return encoded.reshape(B, N, hs), attn_weights, features

    def backward(self, d_hidden, d_feats):
        if self.encoder is not None:
            B, N, hs = d_hidden.shape
            # the blocks' weight gradients run beside the hybrid embedding's (ResNeSt) backward pass
            with ops.lazy_wgrads() if _VIT_LAZY == "enc" else contextlib.nullcontext():
                d_hidden = self.encoder.backward(d_hidden.reshape(B, N, 1, hs).contiguous()).reshape(B, N, hs)
        self.embeddings.backward(d_hidden, d_feats)


class VisionTransformer(TrainStepDriver, nn.Module):
    def __init__(self, batch_size, img_size=(256, 80), num_classes=3, learning_rate=1e-3, weight_decay=1e-4, *,
                 in_channels: int = 10, use_vit: bool = False, device: Optional[str] = None, seed: Optional[int] = 0,
                 transunet: bool = False):
        """``transunet=True`` builds the older self-contained copy of this model in TBI_TransUNet.py (BASELINE configs[3]):
        BatchNormalization where ResNest.py / Decoder.py use LayerNormalization (:304,426,465,472,503), conv_4 = 256 channels
        (:368), and the loss is CategoricalCrossentropy(label_smoothing=0.1) with the DEFAULT reduction - the mean over all
        B*H*W pixels (:546,583) instead of sum / global batch.  See ultrasound_modeling_amd/TBI_TransUNet.py for its surface."""
        super().__init__()
        if seed is not None:
            torch.manual_seed(seed)
        assert img_size[0] % 16 == 0 and img_size[1] % 16 == 0, "H and W must be multiples of 16 (four 2x2 poolings)"
        self.num_classes = num_classes
        self.img_size = tuple(img_size)
        self.transunet = transunet
        self.transformer = Transformer(img_size, in_channels=in_channels, use_vit=use_vit, transunet=transunet)
        self.decoder = DecoderCup(num_classes, grid=(img_size[0] // 16, img_size[1] // 16), norm="bn" if transunet else "ln")
        self.input_shape = [img_size[0], img_size[1], in_channels]
        self.batch_size = batch_size            # GLOBAL batch: the loss is divided by it (:227)
        self.weight_decay = weight_decay        # accepted and unused, as in the reference (:242 is commented out)
        self.learning_rate = learning_rate
        dev = device or ("cuda" if torch.cuda.is_available() else None)
        if dev is None:
            raise RuntimeError("VisionTransformer needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device(dev)
        self.flat = FlatParams(self, self.device)
        self.optimizer = AdamClip(self.flat, lr=learning_rate, clip_norm=1.0)          # :204,:244
        self._loss = torch.zeros(ops.ACC_FLOATS, dtype=torch.float32, device=self.device)   # [0] = the scalar loss
        self.grad_sync = None                   # set by the data-parallel wrapper: callable(flat_grad)
        self._graph = None

    # ------------------------------------------------------------------ parameters in / out (Keras names)
    @property
    def visionModel(self):
        """The drivers only use ``.layers`` and ``.save`` on it (MainNumpy.py:172,177)."""
        return self

    def save(self, path):
        """model.save of a compiled Keras model (MainNumpy.py:177): variables, BN statistics AND the optimiser state."""
        d = self.export_params()
        d.update(self.optimizer.state_dict())
        torch.save(d, path)

    def load(self, path):
        d = torch.load(path, map_location="cpu", weights_only=True)
        self.optimizer.load_state_dict({k: d.pop(k) for k in list(d) if k.startswith("__adam_")})
        self.load_params(d)

    @property
    def layers(self):
        return [m for m in self.modules() if isinstance(m, (Conv2D, BatchNormalization))]

    def load_params(self, params: dict):
        """Load a {reference-attribute-path: tensor} dict (the oracle's naming).  BN moving statistics included."""
        own = dict(self.named_parameters())
        bufs = {}
        for name, mod in self.named_modules():
            if isinstance(mod, BatchNormalization):
                bufs[name + ".moving_mean"] = mod.moving_mean
                bufs[name + ".moving_variance"] = mod.moving_variance
        missing = [k for k in own if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]} ...")
        for k, v in params.items():
            t = own.get(k)
            if t is not None:
                assert t.numel() == v.numel() and tuple(t.shape)[-2:] == tuple(v.shape)[-2:] or tuple(t.shape) == tuple(v.shape), \
                    f"{k}: {tuple(t.shape)} vs {tuple(v.shape)}"
                t.data.copy_(v.to(torch.float32).reshape(t.shape))        # Dense [in,out] == 1x1 conv [1,1,in,out]
            elif k in bufs:
                bufs[k].copy_(v.to(torch.float32))
            else:
                raise KeyError(f"unexpected parameter {k}")
        repack_all(self)

    def repack(self):
        repack_all(self)

    def export_params(self) -> dict:
        out = {k: v.data.detach().clone() for k, v in self.named_parameters()}
        for name, mod in self.named_modules():
            if isinstance(mod, BatchNormalization):
                out[name + ".moving_mean"] = mod.moving_mean.clone()
                out[name + ".moving_variance"] = mod.moving_variance.clone()
        return out

    def export_grads(self) -> dict:
        return {k: v.grad.detach().clone() for k, v in self.named_parameters()}

    # ------------------------------------------------------------------ forward / loss (:220-227)
    def _prep_x(self, x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        return x.to(self.device).contiguous()

    def _prep_y(self, y):
        if not torch.is_tensor(y):
            y = torch.as_tensor(y)
        return y.to(device=self.device, dtype=torch.float32).contiguous()

    def forward(self, x):
        """-> (probs fp32 [B,H,W,classes], attn_weights) (:220-223)."""
        x = self._prep_x(x)
        hidden, attn_weights, features = self.transformer.forward(x, feature_slots=self.decoder.prepare(x.shape[0], self.device))
        return self.decoder.forward(hidden, features), attn_weights

    def _forward_loss(self, x, y, with_grad: bool):
        hidden, _, features = self.transformer.forward(x, feature_slots=self.decoder.prepare(x.shape[0], self.device), need_weights=False)
        fused = _FUSED_HEAD and self.decoder.quad_head
        xh = self.decoder.forward(hidden, features, return_head_input=True) if fused else None
        logits = None if fused else self.decoder.forward(hidden, features, return_logits=True)
        B = (xh if fused else logits).shape[0]
        H, W = self.decoder.out_hw
        qw = self.decoder.quad_w                 # head in quad form: logits / dlogits are [B,H/2,W/2,16]
        probs = torch.empty((B, H, W, self.num_classes), dtype=torch.float32, device=self.device)
        dlogits = None
        if with_grad:   # quad layout: the four pixels of a 2x2 block fill all 16 channels of their quad pixel, so no zero-fill
            dlogits = ops.new_act(B, H // 2, W // 2, 16, self.device) if qw else ops.new_act(B, H, W, 8, self.device)
        # :205,:227 sum / GLOBAL batch; TBI_TransUNet.py:546 (default reduction): mean over the B*H*W pixels of the batch
        scale = 1.0 / float(B * H * W) if self.transunet else 1.0 / float(self.batch_size)
        if fused:       # head conv + softmax + loss in one launch (:142, :205, :225-229); falls back to the three launches if it has no kernel
            if self.decoder._quad.forward_loss(xh, y, probs, self._loss, dlogits, label_smoothing=0.1, clip_eps=1e-7, inv_global_batch=scale):
                return probs, dlogits
            logits = self.decoder._head_forward(xh)
            self.decoder._logits = logits
        ops.softmax_loss(logits, y, probs, self._loss, dlogits, HW=H * W, C_classes=self.num_classes, loss_kind=0,
                         label_smoothing=0.1, clip_eps=1e-7, inv_global_batch=scale, quad_w=qw)
        return probs, dlogits

    def compute_loss(self, y_true, y_pred):
        """VisionTransformer.py:225-227: CategoricalCrossentropy(label_smoothing=0.1, reduction NONE) on PROBABILITIES, summed
        and divided by the global batch size -> scalar.  (train_step/step use the fused softmax+loss kernel instead.)"""
        y_true, y_pred = self._prep_y(y_true), self._prep_y(y_pred)
        H, W = y_pred.shape[1], y_pred.shape[2]
        loss = torch.empty(ops.ACC_FLOATS, dtype=torch.float32, device=self.device)
        ops.fill_f32(loss, 0.0)
        ops.loss_from_probs(y_pred, y_true, loss, HW=H * W, C_classes=y_pred.shape[-1], loss_kind=0, label_smoothing=0.1, clip_eps=1e-7,
                            inv_global_batch=1.0 / float(self.batch_size))
        return loss[0].clone()

    def step(self, x, y):
        """Evaluation step (:248-254): -> (loss, probs)."""
        probs, _ = self._forward_loss(self._prep_x(x), self._prep_y(y), with_grad=False)
        return self._loss[0].clone(), probs

    # ------------------------------------------------------------------ hooks of the shared step order (step.TrainStepDriver)
    def _zero_grad(self):
        self.flat.zero_grad()

    def _forward_backward(self, x, y):
        probs, dlogits = self._forward_loss(x, y, with_grad=True)                      # :240-241
        with ops.overlap_region():              # finishing reductions deferred and batched until the optimiser needs them
            with ops.lazy_wgrads():             # the decoder's weight gradients run on the side stream beside the encoder's backward pass
                d_hidden, d_feats = self.decoder.backward(dlogits)                     # :243
            self.transformer.backward(d_hidden, d_feats)
        return probs

    def _repack(self):
        repack_all(self)

    def _graph_state(self):
        bn = [b for m in self.modules() if isinstance(m, BatchNormalization) for b in (m.moving_mean_p, m.moving_variance_p)]
        return [self.flat.flat] + self.optimizer.state_tensors() + bn

    def train_step(self, x, y):
        """One optimisation step (:235-246): -> (loss, probs).  Loss = sum of per-pixel CCE / GLOBAL batch size.
        After ``capture_graph`` the returned loss and probabilities are the captured step's static buffers: valid until the next call
        (a clone of the scalar would be one more dispatch - 4.6 us - behind every replay)."""
        x, y = self._prep_x(x), self._prep_y(y)
        if self._graph is not None:
            return self._loss[0], self._graph_replay(x, y)
        probs = self._train_body(x, y)
        return self._loss[0].clone(), probs

    def __call__(self, x, *args, **kwargs):
        return self.forward(x)
