"""The order of one optimisation step, shared by both model wrappers (VisionTransformer.py:235-246; TBI_ResNest.py:35-55)
and by the data-parallel path (MainParallel.py:117-146).

    zero gradients -> forward -> loss -> backward            (tape, VisionTransformer.py:237-243)
    [data parallel] per-replica clip-by-global-norm          (:244 - it precedes :245, so every replica clips ITS OWN gradient)
    [data parallel] SUM all-reduce of the flat gradient      (apply_gradients under MirroredStrategy, :245 / MainParallel.py:130)
    clip (single replica only) + Adam                        (:244-245)
    rebuild the bf16 GEMM operands from the fp32 masters

This file is host logic only (no kernel is launched from here directly): a wrapper supplies three hooks
(``_zero_grad``, ``_forward_backward``, ``_repack``) and an ``optimizer`` with ``clip_local / advance / apply / apply_range``.
That makes the ordering testable on CPU with gloo (tests/test_cpu_host.py drives THIS class with oracle arithmetic in the
hooks) and on one GPU with an in-process RCCL group (tests/test_gpu_step.py).

HIP-graph replay: the whole step is captured once per input shape and replayed (one graph; with a gradient exchange the
step is one graph up to the per-replica clip, the collective, then the update).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Tuple

import torch


class TrainStepDriver:
    grad_sync: Optional[Callable] = None     # set by MainParallel.MirroredTrainer: callable(flat_grad) [+ .chunks(flat_grad)]
    _graph = None

    # ------------------------------------------------------------------ hooks a wrapper implements
    def _zero_grad(self):
        raise NotImplementedError

    def _forward_backward(self, x, y):
        """forward, loss, backward: parameter gradients accumulated into ``flat.grad``; -> probabilities."""
        raise NotImplementedError

    def _repack(self):
        raise NotImplementedError

    def _graph_state(self) -> List[torch.Tensor]:
        """Tensors that the capture warm-up steps modify and ``capture_graph`` must put back (weights, Adam state, BN statistics)."""
        return []

    # ------------------------------------------------------------------ the step
    def _grad_body(self, x, y):
        self._zero_grad()
        probs = self._forward_backward(x, y)                       # :240-243
        if self.grad_sync is not None:
            self.optimizer.clip_local()                            # per-replica clip (:244) BEFORE the exchange
        return probs

    def _update_body(self):
        self.optimizer.apply(already_clipped=self.grad_sync is not None)      # (:244-)245 clip + Adam
        self._repack()

    def _sync_and_update(self):
        """Gradient exchange + update.  A ``grad_sync`` with ``chunks`` pipelines the exchange: chunk k+1 is on the wire
        (RCCL's stream) while the Adam kernel of chunk k runs - exact, because after the per-replica clip both the SUM and
        the Adam update are elementwise over the flat buffer."""
        gs = self.grad_sync
        chunks = getattr(gs, "chunks", None)
        if chunks is None:
            gs(self.flat.grad)
            self._update_body()
            return
        self.optimizer.advance()
        for lo, hi, wait in chunks(self.flat.grad):
            wait()
            self.optimizer.apply_range(lo, hi)
        self._repack()

    def _train_body(self, x, y):
        probs = self._grad_body(x, y)
        if self.grad_sync is not None:
            self._sync_and_update()
        else:
            self._update_body()
        return probs

    # ------------------------------------------------------------------ HIP-graph replay of the whole step
    def capture_graph(self, x, y, warmup: int = 2):
        """Capture the training step for inputs of this shape.  The ``warmup`` eager steps it needs (workspaces grow, lazy
        tables are built) run on a throw-away copy of the training state: weights, Adam moments / step counter and BatchNorm
        statistics are restored afterwards, so enabling the graph does not consume optimisation steps."""
        x, y = self._prep_x(x), self._prep_y(y)
        self._gx, self._gy = x.clone(), y.clone()
        state = self._graph_state()
        saved = [t.clone() for t in state]
        sync_saved = self.grad_sync
        self.grad_sync = None if sync_saved is None else _LocalSync()      # no collective during the warm-up (ranks stay in step)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._train_body(self._gx, self._gy)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.grad_sync = sync_saved
        for t, v in zip(state, saved):
            t.copy_(v)
        self._repack()
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        if self.grad_sync is None:
            with torch.cuda.graph(g1):
                self._gprobs = self._train_body(self._gx, self._gy)
            self._graph = (g1, None)
            return
        with torch.cuda.graph(g1):
            self._gprobs = self._grad_body(self._gx, self._gy)
        g2 = None
        if getattr(self.grad_sync, "chunks", None) is None:          # one collective -> the update is a second graph
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update_body()
        self._graph = (g1, g2)

    def graph_inputs(self):
        """The (x, y) buffers the captured step reads: write the next batch THERE and pass these tensors to ``train_step`` - no copy."""
        return self._gx, self._gy

    def _graph_replay(self, x, y):
        """-> the STATIC probability buffer of the captured step (overwritten by the next replay)."""
        # a producer that writes the batch straight into the captured step's input buffers (``graph_inputs()``: the device input
        # pipeline, a data loader with pinned staging) skips these two copies
        if x.data_ptr() != self._gx.data_ptr():
            self._gx.copy_(x)
        if y.data_ptr() != self._gy.data_ptr():
            self._gy.copy_(y)
        g1, g2 = self._graph
        g1.replay()
        if self.grad_sync is not None:
            if g2 is not None:
                self.grad_sync(self.flat.grad)
                g2.replay()
            else:
                self._sync_and_update()
        return self._gprobs


class _LocalSync:
    """Stands in for the exchange during graph warm-up: the data-parallel code path (clip_local, already-clipped Adam)
    without a collective."""

    def __call__(self, flat_grad):
        return None


def even_chunks(n: int, k: int, align: int = 1024) -> List[Tuple[int, int]]:
    """Split [0, n) into <= k contiguous ranges whose boundaries are multiples of ``align`` (the last one takes the tail)."""
    k = max(1, min(k, max(1, n // align)))
    per = (n + k - 1) // k
    per = (per + align - 1) // align * align
    out, lo = [], 0
    while lo < n:
        hi = min(n, lo + per)
        out.append((lo, hi))
        lo = hi
    return out
