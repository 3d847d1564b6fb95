"""Model wrapper and training step: the surface of the reference's ``VisionTransformer.py`` (:192-254).

``VisionTransformer(batch_size, img_size, num_classes, learning_rate, weight_decay)`` exposes ``forward``, ``step``,
``train_step`` and ``compute_loss`` exactly as the drivers call them (MainNumpy.py:45,95; MainParallel.py:130,158):
``train_step(x, y) -> (loss, probs)`` with x NHWC float32/float64 ``[B,H,W,C]`` and y float32 soft one-hot
``[B,H,W,classes]``.

Differences that are parameters here and literals in the reference: the input channel count (10 at
VisionTransformer.py:100,198) and the 16x5 token grid (:90) which becomes (H/16, W/16).
``use_vit=False`` is BASELINE config 2 ("ResNeSt encoder + Decoder.py, no ViT"): the patch embedding feeds the
decoder directly.  The 8-layer ViT bottleneck (:9-189) is SURVEY.md §8f rank 1 ("next") and is not built yet.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .Decoder import DecoderCup
from .flat import AdamClip, FlatParams
from .layers import BatchNormalization, Conv2D
from .ResNest import ResNest, cardinal, residual_S
from .ops import BF16, roundup

input_size = (256, 80)   # VisionTransformer.py:7


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
            elif isinstance(m, Conv2D) and m.wp_f is not None:
                jobs += m.pack_jobs()
        dev = next(root.parameters()).device
        table = (ops.make_pack_table(jobs, dev), len(jobs))
        object.__setattr__(root, "_pack_table", table)
    ops.pack_weights_batched(*table)


class Embeddings(nn.Module):
    """VisionTransformer.py:81-124: ResNest hybrid model + 1x1 patch embedding (+ constant zero position term)."""

    def __init__(self, img_size, hidden_size=512, dropout_rate=0.0, wDecay=None, in_channels=10):
        super().__init__()
        self.img_size, self.hidden_size, self.wDecay = img_size, hidden_size, wDecay
        self.grid_size = (img_size[0] // 16, img_size[1] // 16)      # (16, 5) at 256x80 (:90)
        self.seq_len = self.grid_size[0] * self.grid_size[1]
        self.hybrid_model = ResNest(img_size[0], img_size[1], in_channels, radix=3, ksize=3, kpaths=3)   # :100
        self.patch_embeddings = Conv2D(512, hidden_size, 1, init="glorot")                                # :106

    def forward(self, x):
        x4, features = self.hybrid_model.forward(x)                                   # :113
        e = self.patch_embeddings.forward(x4)                                         # :114
        B = e.shape[0]
        return e.reshape(B, self.seq_len, self.hidden_size), features                 # :116 (+ zeros, :118; dropout 0)

    def backward(self, d_hidden, d_feats):
        B = d_hidden.shape[0]
        gh, gw = self.grid_size
        d_x4 = self.patch_embeddings.backward(d_hidden.reshape(B, gh, gw, self.hidden_size))
        self.hybrid_model.backward(d_x4, d_feats)


class Transformer(nn.Module):
    """VisionTransformer.py:177-189."""

    def __init__(self, img_size, wDecay=None, in_channels=10, use_vit=False):
        super().__init__()
        self.embeddings = Embeddings(img_size=img_size, in_channels=in_channels)
        self.encoder = None
        if use_vit:
            raise NotImplementedError("ViT bottleneck (VisionTransformer.py:9-189) is SURVEY.md §8f 'next'; use use_vit=False")

    def forward(self, input_ids):
        embedding_output, features = self.embeddings.forward(input_ids)
        return embedding_output, [], features

    def backward(self, d_hidden, d_feats):
        self.embeddings.backward(d_hidden, d_feats)


class VisionTransformer(nn.Module):
    def __init__(self, batch_size, img_size=(256, 80), num_classes=3, learning_rate=1e-3, weight_decay=1e-4, *,
                 in_channels: int = 10, use_vit: bool = False, device: Optional[str] = None, seed: Optional[int] = 0):
        super().__init__()
        if seed is not None:
            torch.manual_seed(seed)
        assert img_size[0] % 16 == 0 and img_size[1] % 16 == 0, "H and W must be multiples of 16 (four 2x2 poolings)"
        self.num_classes = num_classes
        self.img_size = tuple(img_size)
        self.transformer = Transformer(img_size, in_channels=in_channels, use_vit=use_vit)
        self.decoder = DecoderCup(num_classes, grid=(img_size[0] // 16, img_size[1] // 16))
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
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.grad_sync = None                   # set by the data-parallel wrapper: callable(flat_grad)
        self._graph = None

    # ------------------------------------------------------------------ parameters in / out (Keras names)
    @property
    def visionModel(self):
        """The drivers only use ``.layers`` and ``.save`` on it (MainNumpy.py:172,177)."""
        return self

    def save(self, path):
        torch.save(self.export_params(), path)

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
                assert tuple(t.shape) == tuple(v.shape), f"{k}: {tuple(t.shape)} vs {tuple(v.shape)}"
                t.data.copy_(v.to(torch.float32))
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
        hidden, attn_weights, features = self.transformer.forward(self._prep_x(x))
        return self.decoder.forward(hidden, features), attn_weights

    def _forward_loss(self, x, y, with_grad: bool):
        hidden, _, features = self.transformer.forward(x)
        logits = self.decoder.forward(hidden, features, return_logits=True)
        B, H, W, _ = logits.shape
        probs = torch.empty((B, H, W, self.num_classes), dtype=torch.float32, device=self.device)
        dlogits = ops.new_act(B, H, W, 8, self.device) if with_grad else None
        ops.fill_f32(self._loss, 0.0)
        ops.softmax_loss(logits, y, probs, self._loss, dlogits, HW=H * W, C_classes=self.num_classes, loss_kind=0,
                         label_smoothing=0.1, clip_eps=1e-7, inv_global_batch=1.0 / float(self.batch_size))   # :205,:227
        return probs, dlogits

    def compute_loss(self, y_true, y_pred):
        raise NotImplementedError("the loss is fused with the head softmax (usseg_softmax_loss_fwd_bwd); use step()/train_step()")

    def step(self, x, y):
        """Evaluation step (:248-254): -> (loss, probs)."""
        probs, _ = self._forward_loss(self._prep_x(x), self._prep_y(y), with_grad=False)
        return self._loss.clone().reshape(()), probs

    def _grad_body(self, x, y):
        """zero grads, forward, loss, backward (+ the per-replica clip when gradients are exchanged afterwards)."""
        self.flat.zero_grad()
        probs, dlogits = self._forward_loss(x, y, with_grad=True)                      # :240-241
        d_hidden, d_feats = self.decoder.backward(dlogits)                             # :243
        self.transformer.backward(d_hidden, d_feats)
        if self.grad_sync is not None:
            self.optimizer.clip_local()                                                # per-replica clip (:244) BEFORE the exchange
        return probs

    def _update_body(self):
        self.optimizer.apply(already_clipped=self.grad_sync is not None)               # (:244-)245 clip + Adam
        repack_all(self)

    def _train_body(self, x, y):
        probs = self._grad_body(x, y)
        if self.grad_sync is not None:
            self.grad_sync(self.flat.grad)                                             # SUM all-reduce inside apply_gradients
        self._update_body()
        return probs

    def train_step(self, x, y):
        """One optimisation step (:235-246): -> (loss, probs).  Loss = sum of per-pixel CCE / GLOBAL batch size."""
        x, y = self._prep_x(x), self._prep_y(y)
        if self._graph is not None:
            return self._graph_step(x, y)
        probs = self._train_body(x, y)
        return self._loss.clone().reshape(()), probs

    # ------------------------------------------------------------------ HIP-graph replay of the whole step
    def capture_graph(self, x, y, warmup: int = 2):
        """Capture ``train_step`` for inputs of this shape into HIP graphs (the step is ~400 launches of 5-70 us).
        Single GPU: one graph.  Data parallel: two graphs (gradients, update) with the RCCL all-reduce between them."""
        x, y = self._prep_x(x), self._prep_y(y)
        self._gx, self._gy = x.clone(), y.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._train_body(self._gx, self._gy)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.grad_sync is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._gprobs = self._train_body(self._gx, self._gy)
            self._graph = (g, None)
        else:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._gprobs = self._grad_body(self._gx, self._gy)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update_body()
            self._graph = (g1, g2)

    def _graph_step(self, x, y):
        self._gx.copy_(x)
        self._gy.copy_(y)
        g1, g2 = self._graph
        g1.replay()
        if g2 is not None:
            self.grad_sync(self.flat.grad)
            g2.replay()
        return self._loss.clone().reshape(()), self._gprobs

    def __call__(self, x, *args, **kwargs):
        return self.forward(x)
