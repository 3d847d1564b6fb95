"""The reference's NumPy batch provider (``Dataset_2.py``) with the per-batch work on the GPU.

Same constructor and methods as ``Dataset_2.py:23-133`` (``Dataset(train_path, val_path, num_classes)``,
``next_train(batch_size, fix)``, ``next_test(batch_size)``, ``reset_idx``, attributes ``x_tr, y_tr, num_tr, height, width,
channel, num_classes`` ...).  Differences, all deliberate:
  * the ``.npy`` files are loaded with ``allow_pickle=False`` (the reference passes True, Dataset_2.py:28; plain float
    arrays need no unpickling) and kept resident in HBM (the full training set is a few GB of the 288);
  * ``next_train`` augments the whole batch with one fused kernel (``DataAugs.dataAug_batch``) and returns DEVICE tensors
    that ``VisionTransformer.train_step`` takes as they are: x already cast to the model's bf16 NHWC input, y the soft
    class maps of ``label2vec`` (Dataset_2.py:6-20,112);
  * ``label2vec`` is the device kernel ``usseg_label2vec``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import ops
from .DataAugs import dataAug_batch


def label2vec(label: torch.Tensor, num_classes: int) -> torch.Tensor:
    """Dataset_2.py:6-20 on the device: label [B,H,W] fp32 -> [B,H,W,num_classes] fp32."""
    label = label.contiguous().float()
    out = torch.empty(tuple(label.shape) + (num_classes,), dtype=torch.float32, device=label.device)
    L.check(L.load().usseg_label2vec(label.data_ptr(), label.numel(), num_classes, out.data_ptr(), torch.cuda.current_stream().cuda_stream),
            "label2vec")
    return out


class Dataset(object):
    def __init__(self, train_path=None, val_path=None, num_classes=3, device="cuda", train_data=None, val_data=None):
        """``train_data`` / ``val_data``: arrays of the on-disk layout [N,1,H,W,12] (label, 10 displacement channels, bMode;
        Dataset_2.py:33-43) may be passed instead of paths."""
        train_data = np.load(train_path, allow_pickle=False) if train_data is None else np.asarray(train_data)
        val_data = np.load(val_path, allow_pickle=False) if val_data is None else np.asarray(val_data)
        self.device = torch.device(device)
        to_dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
        # label = layer 0; the last layer (bMode) is dropped (Dataset_2.py:33-43)
        self.y_tr = to_dev(train_data[:, 0, :, :, 0], torch.float32)
        self.y_te = to_dev(val_data[:, 0, :, :, 0], torch.float32)
        self.x_tr = to_dev(train_data[:, 0, :, :, 1:-1], torch.float64)
        self.x_te = to_dev(val_data[:, 0, :, :, 1:-1], torch.float64)
        self.num_tr, self.num_te = self.x_tr.shape[0], self.x_te.shape[0]
        self.idx_tr, self.idx_te = 0, 0
        self.height, self.width, self.channel = (int(v) for v in self.x_te.shape[1:])
        self.min_val, self.max_val = float(self.x_te[0].min()), float(self.x_te[0].max())
        self.num_classes = num_classes

    def reset_idx(self):
        self.idx_tr, self.idx_te = 0, 0

    def next_train(self, batch_size=1, fix=False, params=None):
        """Dataset_2.py:88-114 -> (x bf16 [B,H,W,roundup(C,8)], y fp32 [B,H,W,num_classes], terminator), all on the device."""
        start, end = self.idx_tr, self.idx_tr + batch_size
        x_tr, y_tr = self.x_tr[start:end], self.y_tr[start:end]
        terminator = False
        if end >= self.num_tr:
            terminator = True
            self.idx_tr = 0
        else:
            self.idx_tr = end
        if fix:
            self.idx_tr = start
        if x_tr.shape[0] != batch_size:
            x_tr, y_tr = self.x_tr[-1 - batch_size:-1], self.y_tr[-1 - batch_size:-1]
        x, y = dataAug_batch(x_tr, y_tr, params=params, num_classes=self.num_classes)      # :106-112
        return x, y, terminator

    def next_test(self, batch_size=1):
        """Dataset_2.py:117-133 (no augmentation)."""
        start, end = self.idx_te, self.idx_te + batch_size
        x_te, y_te = self.x_te[start:end], self.y_te[start:end]
        terminator = False
        if end >= self.num_te:
            terminator = True
            self.idx_te = 0
        else:
            self.idx_te = end
        if x_te.shape[0] != batch_size:
            x_te, y_te = self.x_te[-1 - batch_size:-1], self.y_te[-1 - batch_size:-1]
        return ops.cast_input(x_te.contiguous(), ops.roundup(self.channel, 8)), label2vec(y_te, self.num_classes), terminator
