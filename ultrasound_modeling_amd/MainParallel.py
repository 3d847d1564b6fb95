"""Single-node data parallelism with the semantics of the reference's ``MainParallel.py`` (MirroredStrategy).

Reference behaviour being mirrored (MainParallel.py:16,117-146,209-210):
  * variables are created once and mirrored on every GPU (:209-210)  -> rank 0's flat parameter buffer is broadcast;
  * each global batch is split into N contiguous per-replica slices (:127-128) -> ``shard_batch``;
  * every replica runs ``train_step`` on its slice with the loss divided by the GLOBAL batch size
    (VisionTransformer.py:227), clips ITS OWN gradients by global norm (:244), and ``apply_gradients`` then
    all-reduces them with SUM before the identical Adam update on every replica (:245 under :130);
  * scalar metrics are SUM-reduced (:131-134) and evaluation outputs gathered (:160,:163).
MI355X design: one process per GPU (torchrun), ONE RCCL all-reduce over the flat fp32 gradient buffer per step
(25 MB for the conv-only Arch B): on the fully connected xGMI node a single large collective uses all 7 links.
BatchNorm statistics are NOT synchronised (the reference uses plain BatchNormalization, ResNest.py:19).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment. -> (rank, world_size, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # USSEG_DIST_BACKEND=gloo: rehearsal of the multi-process path on a box with fewer GPUs than ranks (the ranks then
            # share a device and the exchange goes through host memory - correctness only, never a measurement)
            backend = os.environ.get("USSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(x, y, rank: int, world: int):
    """The contiguous per-replica slice of a global batch (experimental_distribute_dataset, MainParallel.py:128)."""
    B = x.shape[0]
    assert B % world == 0, "global batch must divide evenly over the replicas"
    n = B // world
    return x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]


class GradSync:
    """The gradient exchange of ``apply_gradients`` under MirroredStrategy: SUM all-reduce of the flat fp32 gradient buffer.

    ``nchunks == 1``: ONE collective over the whole buffer (the update then replays as a second HIP graph).
    ``nchunks > 1``: the buffer is exchanged as ``nchunks`` contiguous pieces issued back to back on RCCL's stream; the step
    driver (step.TrainStepDriver._sync_and_update) waits for piece k ON THE COMPUTE STREAM (an event wait, the host does not
    block) and runs the Adam kernel of that range while piece k+1 is still on the wire.  Exact: the per-replica clip already
    happened (VisionTransformer.py:244 precedes :245), and both the SUM and Adam are elementwise over the flat buffer.
    What can NOT be overlapped exactly is the exchange with the backward pass: the reference reduces c_r*g_r where the
    per-replica clip factor c_r = min(1, 1/||g_r||) needs the whole local gradient first."""

    def __init__(self, group=None, nchunks: int = 1, align: int = 1 << 16):
        self.group, self.nchunks, self.align = group, max(1, int(nchunks)), align
        if self.nchunks == 1:
            self.chunks = None               # the driver checks this attribute

    def __call__(self, flat_grad: torch.Tensor):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def chunks(self, flat_grad: torch.Tensor):
        from .step import even_chunks
        ranges = even_chunks(flat_grad.numel(), self.nchunks, self.align)
        works = [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for lo, hi in ranges]
        for (lo, hi), w in zip(ranges, works):
            yield lo, hi, w.wait


class MirroredTrainer:
    """Wraps a model that has ``flat`` (FlatParams), ``grad_sync`` and ``train_step`` (VisionTransformer / Arch A).

    ``chunks``: pieces the gradient exchange is pipelined in (None / 1 = ONE collective and a graph-replayed update - the default;
    k > 1 = k pieces with the Adam kernel of piece i under the wire time of piece i+1, see GradSync).  Measured on one GPU
    (RCCL group of one rank, so no wire time to hide): the chunked form's eagerly launched update costs Arch A +0.2 ms per step
    (7.88 vs 7.68 ms) against at most 0.12 ms of Adam time it could hide behind a 103 MB all-reduce - so it stays opt-in
    (``USSEG_DP_CHUNKS`` / ``bench.py --dp-chunks``) until it can be measured on an 8-GPU node."""

    def __init__(self, net, group=None, force: bool = False, chunks: Optional[int] = None):
        self.net, self.group = net, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > 1 or (force and dist.is_initialized()):
            dist.broadcast(net.flat.flat, src=0, group=group)          # mirrored variables (:209-210)
            for m in net.modules():
                for b in m._buffers.values():
                    if b is not None:
                        dist.broadcast(b, src=0, group=group)
            if hasattr(net, "repack"):
                net.repack()
            if chunks is None:
                chunks = int(os.environ.get("USSEG_DP_CHUNKS", "0")) or 1
            net.grad_sync = GradSync(group, chunks)

    def train_step(self, x_local, y_local):
        """One mirrored step on this replica's slice -> (global loss, local probs) (MainParallel.py:130-131)."""
        loss, probs = self.net.train_step(x_local, y_local)
        if self.world > 1:
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        return loss, probs

    def test_step(self, x_local, y_local):
        """mirrored_test_step (MainParallel.py:148-176): SUM-reduced loss (:159), the probabilities of all replicas gathered along
        the batch axis (:160) and the labels likewise (:163) -> (loss, probs [B_global,...], y [B_global,...])."""
        loss, probs = getattr(self.net, "eval_step", self.net.step)(x_local, y_local)
        y = y_local
        if self.world > 1:
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
            probs, y = self._gather(probs), self._gather(y_local.to(probs.device))
        return loss, probs, y

    def _gather(self, t):
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.cat(parts, dim=0)
