"""Augmentations of the reference's ``DataAugs.py`` (shift / clip / noisy / imageReduc / dataAug) on the GPU.

The reference runs them as pure-Python O(H*W) loops per sample on the host (DataAugs.py:13-23,33-37,60-72) - tens of
milliseconds per image, ahead of a 5 ms training step.  Here the random DRAWS stay on the host, in the reference's order
(Python ``random``: r, t, then per clip box four draws, then per shift three, DataAugs.py:83-101) so that a seeded run
picks the same boxes and shifts, and the whole batch is transformed by ONE fused kernel (``usseg_augment``) that follows
the reference as executed: the loops' ``si - 1`` bounds and the dilation of ``imageReduc`` that never fires
(DataAugs.py:63) included.  Only ``noisy`` differs in the stream of its Gaussian (NumPy's generator on the host there, a
counter-based generator on the device here); a noise field can be injected for bit-exact comparisons.
"""
from __future__ import annotations

import ctypes as C
import random
from typing import List, Optional

import torch

from . import _lib as L
from . import ops


def draw(rng=random) -> dict:
    """The draws of one ``dataAug`` call (DataAugs.py:83-101), in the reference's order."""
    r = rng.randint(0, 100000)
    t = rng.randint(0, 100000)
    p = {"reduc": r % 3 != 0, "reduc_t": t % 7 + 2, "clips": [], "shift": None, "noise": bool(t % 3), "seed": r * 100003 + t}
    for _ in range(r % 3):
        p["clips"].append((rng.randint(0, 256), rng.randint(0, 80), rng.randint(20, 40), rng.randint(10, 20)))   # :28-31
    if t % 2:
        p["shift"] = (rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 1))                                  # :7-9
    return p


def _sample_table(params: List[dict], device) -> torch.Tensor:
    arr = (L.AugSample * len(params))()
    for i, p in enumerate(params):
        s = arr[i]
        s.do_reduc, s.nclip = int(p["reduc"]), len(p["clips"])
        for k, box in enumerate(p["clips"]):
            for q in range(4):
                s.clip[k][q] = box[q]
        if p["shift"] is not None:
            s.do_shift, (s.shift_r, s.shift_c, s.shift_dir) = 1, p["shift"]
        s.do_noise, s.seed = int(p["noise"]), int(p.get("seed", 0)) & ((1 << 63) - 1)
    raw = bytes(arr)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def dataAug_batch(x: torch.Tensor, y: torch.Tensor, params: Optional[List[dict]] = None, num_classes: int = 3, noise: Optional[torch.Tensor] = None,
                  rng=random, want_f32: bool = False):
    """x [B,H,W,C] fp32/fp64 and y [B,H,W] fp32 ON THE DEVICE -> (x_bf16 [B,H,W,roundup(C,8)], y_vec [B,H,W,num_classes] fp32
    [, x_f32, y_aug]): dataAug per sample (DataAugs.py:82-102) followed by label2vec (Dataset_2.py:112) in one launch."""
    assert x.is_cuda and y.is_cuda and x.dim() == 4 and y.dim() == 3 and x.dtype in (torch.float32, torch.float64) and y.dtype == torch.float32
    x, y = x.contiguous(), y.contiguous()
    B, H, W, Cc = x.shape
    if params is None:
        params = [draw(rng) for _ in range(B)]
    table = _sample_table(params, x.device)
    cp = ops.roundup(Cc, 8)
    xo = torch.empty((B, H, W, cp), dtype=torch.bfloat16, device=x.device)
    yv = torch.empty((B, H, W, num_classes), dtype=torch.float32, device=x.device)
    xf = torch.empty((B, H, W, Cc), dtype=torch.float32, device=x.device) if want_f32 else None
    ya = torch.empty((B, H, W), dtype=torch.float32, device=x.device) if want_f32 else None
    d = L.AugDesc(B, H, W, Cc, cp, num_classes)
    L.check(L.load().usseg_augment(C.byref(d), table.data_ptr(), x.data_ptr(), 1 if x.dtype == torch.float64 else 0, y.data_ptr(),
                                   None if noise is None else noise.data_ptr(), xo.data_ptr(), None if xf is None else xf.data_ptr(),
                                   None if ya is None else ya.data_ptr(), yv.data_ptr(), torch.cuda.current_stream().cuda_stream), "augment")
    return (xo, yv, xf, ya) if want_f32 else (xo, yv)


def dataAug(image: torch.Tensor, label: torch.Tensor):
    """Single-sample form with the reference's signature (DataAugs.py:82): image [H,W,C], label [H,W] -> (image fp32, label)."""
    _, _, xf, ya = dataAug_batch(image[None], label[None].float(), want_f32=True)
    return xf[0], ya[0]
