"""Flat parameter / gradient storage and the clip-by-global-norm + Adam step.

All trainable variables of a model live in ONE fp32 device buffer (and their gradients, Adam moments in
three more of the same shape), so that
  * the global gradient norm, the clip and the Adam update are one kernel each
    (VisionTransformer.py:244-245; TBI_ResNest.py:46),
  * the data-parallel gradient exchange is ONE RCCL all-reduce over the flat gradient buffer
    (MainParallel.py:130 -> apply_gradients under MirroredStrategy),
  * kernels accumulate parameter gradients straight into the flat gradient buffer.
Every variable starts on a 32-byte boundary and is followed by zero padding up to a multiple of 8 floats, so
kernels may read per-channel vectors up to the physical (padded) channel count.  Modules may ask for several
variables to be laid out back to back ("adjacent groups", used by the grouped split-attention kernels).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class FlatParams:
    def __init__(self, module: nn.Module, device):
        self.device = torch.device(device)
        named = list(module.named_parameters())
        self.names = [n for n, _ in named]
        params = [p for _, p in named]
        pid = {id(p): i for i, p in enumerate(params)}
        offsets: Dict[int, int] = {}
        off = 0
        # adjacency groups first
        for m in module.modules():
            fn = getattr(m, "adjacent_params", None)
            if fn is None:
                continue
            for group, extra_pad in fn():
                for p in group:
                    assert id(p) in pid and id(p) not in offsets, "parameter in two adjacency groups"
                    offsets[id(p)] = off
                    off += p.numel()
                off = _pad8(off + extra_pad)
        for p in params:
            if id(p) in offsets:
                continue
            offsets[id(p)] = off
            off = _pad8(off + p.numel())
        self.total = _pad8(off)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.params = params
        self.offsets = [offsets[id(p)] for p in params]
        for p, o in zip(params, self.offsets):
            n = p.numel()
            view = self.flat[o:o + n].view(p.shape)
            view.copy_(p.data.to(torch.float32))
            p.data = view
            p.requires_grad_(False)
            p.grad = self.grad[o:o + n].view(p.shape)
        for m in module.modules():
            for k, b in list(m._buffers.items()):
                if b is not None:
                    m._buffers[k] = b.to(self.device)
        self.n_trainable = sum(p.numel() for p in params)
        for m in module.modules():
            fn = getattr(m, "on_finalize", None)
            if fn is not None:
                fn(self.device)

    def zero_grad(self):
        ops.fill_f32(self.grad, 0.0)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: p.data.detach().clone() for n, p in zip(self.names, self.params)}


class AdamClip:
    """tf.clip_by_global_norm(grads, clip) followed by tf.optimizers.Adam(lr) (Keras defaults: b1 .9, b2 .999, eps 1e-7).

    ``clip_norm=None`` disables clipping (TBI_ResNest.py:46 applies raw gradients).
    The step counter and the bias-corrected learning rate live on the device so that a captured HIP graph of the
    whole training step can be replayed.
    """

    def __init__(self, flat: FlatParams, lr: float = 1e-3, clip_norm: Optional[float] = 1.0, beta1=0.9, beta2=0.999, eps=1e-7):
        self.flat, self.lr, self.clip_norm = flat, float(lr), clip_norm
        self.b1, self.b2, self.eps = beta1, beta2, eps
        dev = flat.device
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_t_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(ops.ACC_FLOATS, dtype=torch.float32, device=dev)   # [0] = squared global norm (reproducible accumulator)

    def global_norm_sq(self):
        ops.sumsq(self.flat.grad, self.sumsq)        # sumsq[0] is overwritten: no zero-fill launch
        return self.sumsq

    def clip_local(self):
        """Per-replica clip, BEFORE the data-parallel all-reduce (VisionTransformer.py:244 precedes :245)."""
        if self.clip_norm is not None:
            self.global_norm_sq()
            ops.scale_by_clip(self.flat.grad, self.sumsq, float(self.clip_norm))

    def advance(self):
        """step += 1 and the bias-corrected learning rate of that step (device side: replayable from a HIP graph)."""
        ops.adam_advance(self.step_dev, self.lr_t_dev, self.lr, self.b1, self.b2)

    def apply_range(self, lo: int, hi: int, clip: float = 0.0):
        """Adam update of flat[lo:hi] (elementwise, so any partition of the buffer gives the same result as one launch)."""
        f = self.flat
        ops.adam_clip_step(f.flat[lo:hi], f.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.sumsq, clip, self.lr_t_dev, self.b1, self.b2,
                           self.eps)

    def apply(self, already_clipped: bool = False):
        clip = 0.0
        if self.clip_norm is not None and not already_clipped:
            # squared global norm and the step counter in one launch
            ops.sumsq_advance(self.flat.grad, self.sumsq, self.step_dev, self.lr_t_dev, self.lr, self.b1, self.b2)
            clip = float(self.clip_norm)
        else:
            self.advance()
        self.apply_range(0, self.flat.total, clip)

    def state_tensors(self):
        return [self.m, self.v, self.step_dev, self.lr_t_dev]

    def state_dict(self):
        """Adam moments in the flat layout + the step counter (what Keras saves with a compiled model)."""
        return {"__adam_m__": self.m.detach().clone(), "__adam_v__": self.v.detach().clone(), "__adam_step__": self.step_dev.detach().clone()}

    def load_state_dict(self, d):
        if not d:
            return
        self.m.copy_(d["__adam_m__"].to(self.m.device))
        self.v.copy_(d["__adam_v__"].to(self.v.device))
        self.step_dev.copy_(d["__adam_step__"].to(self.step_dev.device))
