"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/usseg.h declares (no compute
calls without a GPU), the flat parameter layout logic, and the data-parallel step semantics on world_size-2 gloo."""
import os
import re
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import usseg_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ultrasound_modeling_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "usseg.h")).read()
    declared = set(re.findall(r"\b(usseg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libusseg_hip.so does not export {name}"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.usseg_version() >= 1


def test_bad_arguments_return_errors_without_touching_a_gpu():
    import ctypes as C
    from ultrasound_modeling_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(1, 4, 4, 7, 8, 8, 8, 3, 1, 0, 0.0, 0)       # Cin not a multiple of 8
    rc = lib.usseg_conv2d_fwd(C.byref(d), 16, 16, None, None, 0, 16, None)
    assert rc == -1 and b"multiples of 8" in lib.usseg_last_error()
    d = _lib.ConvDesc(1, 4, 4, 8, 8, 8, 8, 5, 1, 0, 0.0, 0)       # unsupported kernel size
    assert lib.usseg_conv2d_fwd(C.byref(d), 16, 16, None, None, 0, 16, None) == -1
    with pytest.raises(_lib.UssegError):
        _lib.check(-1, "conv2d_fwd")
    # the fused stage entry points: an unsupported channel configuration and a null pointer are argument errors, not launches
    cd = _lib.CardinalDesc(1, 8, 8, 48, 3, 3, 10, 16, 32, 64, 48, 16, 32, 64, 1e-3, 0.3)      # Cin = 48: no fused instantiation
    assert lib.usseg_cardinal_supported(C.byref(cd)) == 0
    args21 = [16] * 21
    assert lib.usseg_cardinal_fwd(C.byref(cd), *args21) == -1 and b"no fused kernel" in lib.usseg_last_error()
    cd = _lib.CardinalDesc(1, 8, 8, 32, 3, 3, 10, 16, 32, 64, 32, 16, 32, 64, 1e-3, 0.3)
    assert lib.usseg_cardinal_supported(C.byref(cd)) == 1
    bwd = [16, 32, 16, 64] + [16] * 12 + [3.0, 16, None, 80] + [16] * 11            # dcat is NULL
    assert lib.usseg_cardinal_bwd(C.byref(cd), *bwd) == -1 and b"null pointer" in lib.usseg_last_error()
    bwd[18] = 16
    bwd[19] = 72                                                                     # ldc < Up + Oc
    assert lib.usseg_cardinal_bwd(C.byref(cd), *bwd) == -1 and b"strides" in lib.usseg_last_error()


def test_product_has_no_cpu_fallback():
    from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        VisionTransformer(batch_size=2, img_size=(32, 32), in_channels=1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ultrasound_modeling_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "usseg_oracle" not in src and "import oracle" not in src, f


def test_flat_params_layout_alignment_and_adjacency():
    import torch.nn as nn
    from ultrasound_modeling_amd.flat import FlatParams

    class Leaf(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.w = nn.Parameter(torch.arange(n, dtype=torch.float32))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.c, self.d = Leaf(3), Leaf(5), Leaf(3), Leaf(3)

        def adjacent_params(self):
            return [([self.a.w, self.c.w, self.d.w], 0)]
    net = Net()
    fp = FlatParams(net, "cpu")
    # a, c, d are back to back; every group / lone variable starts on an 8-float boundary and is zero padded
    assert net.c.w.data_ptr() == net.a.w.data_ptr() + 12 and net.d.w.data_ptr() == net.c.w.data_ptr() + 12
    offs = {n: o for n, o in zip(fp.names, fp.offsets)}
    assert offs["a.w"] % 8 == 0 and offs["b.w"] % 8 == 0 and fp.total % 8 == 0
    assert fp.flat[offs["a.w"]:offs["a.w"] + 9].tolist() == [0, 1, 2, 0, 1, 2, 0, 1, 2]
    assert fp.flat[offs["a.w"] + 9:offs["a.w"] + 16].abs().sum() == 0      # zero gap
    assert fp.n_trainable == 14
    # gradients are views of one flat buffer
    net.b.w.grad.fill_(2.0)
    assert fp.grad[offs["b.w"]:offs["b.w"] + 5].tolist() == [2.0] * 5 and fp.grad.sum() == 10.0


def test_shard_batch_is_contiguous_split():
    from ultrasound_modeling_amd.MainParallel import shard_batch
    x, y = torch.arange(8).reshape(8, 1), torch.arange(8).reshape(8, 1) * 10
    xs, ys = shard_batch(x, y, 1, 4)
    assert xs.flatten().tolist() == [2, 3] and ys.flatten().tolist() == [20, 30]


# ------------------------------------------------------------------------------------------------ world_size-2 gloo
class _CpuFlat:
    def __init__(self, n):
        self.flat = torch.zeros(n)
        self.grad = torch.zeros(n)
        self.total = n


class _CpuAdamClip:
    """flat.AdamClip's interface (clip_local / advance / apply / apply_range) in fp32 torch-CPU arithmetic: the GPU kernels
    cannot run here, the ORDER in which step.TrainStepDriver calls them is what this test drives."""

    def __init__(self, flat, lr=1e-3, clip_norm=1.0):
        self.flat, self.lr, self.clip_norm = flat, lr, clip_norm
        self.m, self.v, self.step, self.lr_t = torch.zeros_like(flat.flat), torch.zeros_like(flat.flat), 0, 0.0
        self.calls = []

    def _scale(self):
        gn = float(self.flat.grad.double().norm())
        return self.clip_norm / max(gn, self.clip_norm)

    def clip_local(self):
        self.calls.append("clip_local")
        self.flat.grad.mul_(self._scale())

    def advance(self):
        self.calls.append("advance")
        self.step += 1
        self.lr_t = self.lr * (1 - 0.999 ** self.step) ** 0.5 / (1 - 0.9 ** self.step)

    def apply_range(self, lo, hi, clip=0.0):
        self.calls.append(("adam", lo, hi))
        g = self.flat.grad[lo:hi] * (self._scale() if clip > 0 else 1.0)
        m, v = self.m[lo:hi], self.v[lo:hi]
        m.mul_(0.9).add_(g, alpha=0.1)
        v.mul_(0.999).addcmul_(g, g, value=0.001)
        self.flat.flat[lo:hi].sub_(self.lr_t * m / (v.sqrt() + 1e-7))

    def apply(self, already_clipped=False):
        self.advance()
        self.apply_range(0, self.flat.total, 0.0 if already_clipped else self.clip_norm)


def _make_cpu_net():
    from ultrasound_modeling_amd.step import TrainStepDriver

    class _CpuNet(TrainStepDriver):
        """The PRODUCT's step driver (step.TrainStepDriver: _grad_body / _sync_and_update / _update_body / _train_body) with
        the oracle's forward / backward in the hooks and a torch-CPU Adam: what runs under gloo is the shipped ordering logic."""

        def __init__(self, P, global_batch):
            self.P, self.names = P, O.trainable_names(P)
            self.sizes = [P[n].numel() for n in self.names]
            self.flat = _CpuFlat(sum(self.sizes))
            self.flat.flat.copy_(torch.cat([P[n].reshape(-1) for n in self.names]).float())
            self.optimizer = _CpuAdamClip(self.flat)
            self.grad_sync, self.global_batch = None, global_batch
            self._buffers = {k: v for k, v in P.items() if k not in self.names}     # BN moving statistics
            self.repacks = 0

        def modules(self):
            return [self]

        def _views(self, buf):
            out, o = [], 0
            for n, sz in zip(self.names, self.sizes):
                out.append(buf[o:o + sz].reshape(self.P[n].shape))
                o += sz
            return out

        def _zero_grad(self):
            self.flat.grad.zero_()

        def _forward_backward(self, x, y):
            leaves = [t.double().clone().requires_grad_(True) for t in self._views(self.flat.flat)]
            Pl = dict(self.P)
            Pl.update(zip(self.names, leaves))
            probs = O.vision_transformer_forward(x, Pl, 3, 3, as_executed=False)
            self.loss = O.compute_loss(y, probs, self.global_batch)
            for dst, g in zip(self._views(self.flat.grad), torch.autograd.grad(self.loss, leaves)):
                dst.add_(g.float())
            return probs.detach()

        def _repack(self):
            self.repacks += 1

        def train_step(self, x, y):
            probs = self._train_body(x, y)
            return self.loss.detach().float(), probs

        def step(self, x, y):
            probs = O.vision_transformer_forward(x, dict(self.P, **dict(zip(self.names, [t.double() for t in self._views(self.flat.flat)]))), 3, 3,
                                                 as_executed=False)
            return O.compute_loss(y, probs, self.global_batch).float(), probs
    return _CpuNet


def _dp_worker(rank, world, port, tmp, chunks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    from ultrasound_modeling_amd.MainParallel import MirroredTrainer, init_distributed, shard_batch
    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.set_num_threads(2)
    P = O.init_vision_transformer_params(channel=1, seed=1 + rank, perturb=True)       # ranks start DIFFERENT ...
    net = _make_cpu_net()(P, global_batch=4)
    tr = MirroredTrainer(net, chunks=chunks)                                            # ... and rank 0's weights are mirrored
    assert (getattr(net.grad_sync, "chunks", None) is not None) == (chunks > 1)
    x, y = O.synthetic_batch(4, 32, 32, 1, seed=9)
    xs, ys = shard_batch(x, y, rank, world)
    loss, _ = tr.train_step(xs, ys)
    calls = net.optimizer.calls
    # the shipped order: per-replica clip BEFORE the exchange, then the already-clipped Adam (whole buffer or chunk by chunk)
    assert calls[0] == "clip_local" and calls[1] == "advance" and all(c[0] == "adam" for c in calls[2:]), calls
    assert [c[1] for c in calls[2:]] == sorted(c[1] for c in calls[2:]) and calls[2][1] == 0 and calls[-1][2] == net.flat.total
    assert len(calls) - 2 == (chunks if chunks > 1 else 1) and net.repacks == 1
    # mirrored_test_step (MainParallel.py:148-176): loss SUM-reduced, probabilities AND labels gathered along the batch axis
    tl, tp, ty = tr.test_step(xs, ys)
    assert tp.shape[0] == 4 and torch.equal(ty, y), "labels are gathered in rank order (MainParallel.py:163)"
    l_all, p_all = net.step(x, y)
    assert torch.allclose(tp, p_all, atol=1e-12) and abs(tl.item() - l_all.item()) < 1e-4 * abs(l_all.item())
    torch.save({"loss": loss, "P": {n: t.clone() for n, t in zip(net.names, net._views(net.flat.flat))}}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunks", [1, 3])
def test_mirrored_step_world2_gloo(tmp_path, chunks):
    """Two replicas driving the PRODUCT's step order (step.TrainStepDriver + MainParallel.MirroredTrainer / GradSync): weights
    mirrored from rank 0, batch split contiguously, loss / GLOBAL batch, per-replica clip, SUM all-reduce (one collective, or
    three pipelined chunks), identical Adam update everywhere (MainParallel.py:117-146,209-210; VisionTransformer.py:227,244-245)."""
    port = 29500 + (os.getpid() % 2000) + chunks
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), chunks), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for k in r0["P"]:
        assert torch.equal(r0["P"][k], r1["P"][k]), k                       # replicas stay in lock step
    # single-process emulation of the same semantics with the oracle's own clip / Adam
    P = {k: v.float().double() for k, v in O.init_vision_transformer_params(channel=1, seed=1, perturb=True).items()}
    x, y = O.synthetic_batch(4, 32, 32, 1, seed=9)
    names = O.trainable_names(P)
    clipped, losses = [], []
    for r in range(2):
        leaves = [P[n].clone().requires_grad_(True) for n in names]
        Pl = dict(P)
        Pl.update(zip(names, leaves))
        probs = O.vision_transformer_forward(x[2 * r:2 * r + 2], Pl, 3, 3, as_executed=False)
        loss = O.compute_loss(y[2 * r:2 * r + 2], probs, 4)                  # GLOBAL batch 4
        g = torch.autograd.grad(loss, leaves)
        clipped.append(O.clip_by_global_norm(g)[0])                          # clip BEFORE the exchange
        losses.append(loss.item())
    summed = [(a.float() + b.float()).double() for a, b in zip(*clipped)]    # the exchange happens in fp32
    new = [P[n].clone() for n in names]
    O.adam_step(new, summed, [torch.zeros_like(t) for t in new], [torch.zeros_like(t) for t in new], 1, 1e-3)
    assert abs(r0["loss"].item() - sum(losses)) < 1e-4 * abs(sum(losses))    # scalar SUM-reduce (MainParallel.py:131)
    for n, t in zip(names, new):
        assert torch.allclose(r0["P"][n].double(), t, rtol=1e-6, atol=2e-6), n   # first Adam step ~ lr*sign(g): entries with |g| near fp32 noise may differ


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` started by hand (no torchrun environment) must start its two ranks itself, as CHILD processes, and
    relay rank 0's single JSON line (the driver's N>1 scaling runs use torch.distributed.run; a maintainer types the short form).
    --dry-run swaps the model for a CPU stand-in over gloo: launch, batch split, exchange, max-over-ranks timing and reporting are
    the shipped code."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["dry_run"] is True
    assert d["config"]["per_gpu_batch"] == 16 and d["config"]["global_batch"] == 32 and d["config"]["dp_exchange_chunks"] == 1
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # a failing rank makes the launcher exit non-zero
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--dry-run", "--scaling", "strong",
                        "--arch", "T"], capture_output=True, text=True, timeout=600, env=dict(env, USSEG_BENCH_FAIL_RANK="1"))
    assert r.returncode != 0
