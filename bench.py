#!/usr/bin/env python3
"""Headline benchmark: segmentation TRAINING images/sec at 256x256 on N MI355X (BASELINE.json metric).

A "step" is one full training step (forward + backward + per-replica clip + gradient all-reduce + Adam + operand
repack) on a synthetic batch of 256x256x1 tiles that is already resident in HBM.

  --arch B (default)  BASELINE configs[1]: Arch B (ResNest.py r=3,k=3 encoder + patch embedding + Decoder.py DecoderCup,
                      no ViT), 16 images per GPU - the configuration the metric is quoted on.
  --arch A            BASELINE configs[2]: Arch A (TBI_ResNest.py model, r=3,k=4, my_loss_cat, Adam 5e-3), 32 images per GPU.
  --scaling weak (default)  the per-GPU batch is fixed, the global batch (which divides Arch B's loss) is batch*N;
  --scaling strong          the GLOBAL batch is fixed at 16 (B) / 32 (A) and split over the N replicas (MainParallel.py:127-128).

  python bench.py --gpus N --steps K --warmup W
      N>1 under torch.distributed.run (RANK / WORLD_SIZE in the environment): this process is one rank.
      N>1 WITHOUT that environment: this process only launches `python -m torch.distributed.run --nproc-per-node N bench.py ...`
      as a child (before anything touches the GPU - no exec of a GPU process), relays its output and exits with its code.
  --dry-run   plumbing rehearsal without a GPU: a 6-parameter per-pixel stand-in model on the CPU driven through the SAME
              step.TrainStepDriver / MainParallel.MirroredTrainer code over gloo; its `value` measures nothing ("dry_run": true).

Prints ONE JSON line on rank 0 (see the keys below).  Extra objects:
  roofline     - the dominant kernel family (the conv kernels that run every plain conv / tconv forward and
                 backward-data launch): algorithmic FLOPs of those launches in one step / their summed duration,
                 measured with HIP events recorded by the library on the launch stream.  `fused_tiles` times the
                 fused tile kernels (cardinal forward / backward, stem forward: convs + norms + activations in one
                 launch) with the conv FLOPs they carry, `family` is both together, `wgrad` the weight gradients.
  cpu_baseline - the CPU oracle (a PyTorch-CPU fp32 restatement of the reference path; TensorFlow, hence the
                 reference itself, cannot run here) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# kernel arguments in device memory (the runtime's default on this ROCm; measured on one box: 2.75 ms per step with it, 3.02 without - every one of the
# step's 137 dispatches reads its arguments first).  Set before the HIP runtime loads with torch; an explicit setting in the environment wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md
HBM_BYTES_PER_S = 8.0e12     # HBM3E peak, MI355X_MICROARCH.md
BASE_BATCH = {"B": 16, "A": 32, "T": 8, "S": 16}   # images per GPU at N=1 (BASELINE configs[1] / [2] / [3] / [4])
H = W = 256
C_IN = 1


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU share of this process: affinity mask, capped by the cgroup CPU quota (the GPU box gives 16 per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def algorithmic_flops(net, arch):
    """Algorithmic FLOPs (2*MACs, logical channel counts; Arch B's radix branches de-duplicated, Arch A's are distinct layers)
    of the conv / tconv layers for the batch of the last forward pass: (forward, igemm family = fwd + dgrad actually run, wgrad)."""
    from ultrasound_modeling_amd.layers import Conv2D
    from ultrasound_modeling_amd import ops
    fwd = igemm = fused = 0.0     # fused: the part of `igemm` that runs inside the fused tile kernels (cardinal fwd / bwd, stem fwd)
    attain = [0.0]      # seconds the fwd + dgrad launches would take on their own rooflines: max(FLOPs / MFMA peak, min bytes / HBM)

    def roof(f, M, cin, cout, passes):
        attain[0] += passes * max(f / (PEAK_BF16_TFLOPS * 1e12), 2.0 * M * (cin + cout) / HBM_BYTES_PER_S)
    managed = set()
    groups = []         # (saved input, [(1x1 conv, kxk conv), ...]) of every grouped split-attention launch pair
    if arch == "B":
        from ultrasound_modeling_amd.ResNest import residual_S
        root = net
        first = net.transformer.embeddings.hybrid_model.conv1
        from ultrasound_modeling_amd import ResNest as _RN
        enc = net.transformer.embeddings.hybrid_model
        fused_ids = {}      # id(conv) -> passes of it (forward = 1, backward-data = 1) that run inside a fused tile kernel
        if getattr(enc, "_stem_fused", False):
            fused_ids.update({id(enc.conv1): 1, id(enc.convtmp_1): 1, id(enc.convtmp_2): 1})
        from ultrasound_modeling_amd import VisionTransformer as _VT
        dec = net.decoder
        if _VT._FUSED_HEAD and dec.quad_head and dec.head.cin_p in (16, 72):      # head conv + softmax + loss (usseg_head_quad_softmax_loss)
            fused_ids[id(dec.head)] = 1
        for m in net.modules():
            if isinstance(m, residual_S):
                g_ = m._group
                x_ = g_._saved[0]
                if g_.fused_ok(m.convtmp_sc):      # forward: grouped 1x1 + grouped 3x3 + shortcut 1x1 in usseg_cardinal_fwd
                    bwd_in = (_RN._FUSED_CARDINAL_BWD and m.wcat_d is not None and x_.shape[0] * x_.shape[1] * x_.shape[2] < _RN._CARD_BWD_MAX_PX)
                    fused_ids[id(m.convtmp_sc)] = 1
                    for c in g_.cards:
                        fused_ids[id(c.conv1)] = 1
                        fused_ids[id(c.conv2)] = 2 if bwd_in else 1      # + the grouped 3x3's backward-data pass in usseg_cardinal_bwd
                groups.append((m._group._saved[0], [(c.conv1, c.conv2) for c in m._group.cards]))
                for c in m._group.cards:
                    managed.update((id(c.split.dense1), id(c.split.dense2)))
    else:
        root = net.resModel
        first = root.Conv1
        for st in root._build():
            for sl in st.slabs:
                groups.append((sl._saved[0], [(b[0], b[2]) for b in sl.br]))
                for a1, _, a2s in sl.att:
                    managed.update([id(a1)] + [id(c) for c in a2s])
    for x, pairs in groups:
        Bx, Hx, Wx, _, _ = ops.geom(x)
        M = Bx * Hx * Wx
        f1 = sum(2.0 * M * c1.cin * c1.cout for c1, _ in pairs)
        f2 = sum(2.0 * M * c2.k * c2.k * c2.cin * c2.cout for _, c2 in pairs)
        fwd += f1 + f2
        igemm += 2 * (f1 + f2)
        if arch == "B":
            fused += sum(2.0 * M * c1.cin * c1.cout * fused_ids.get(id(c1), 0) for c1, _ in pairs)
            fused += sum(2.0 * M * c2.k * c2.k * c2.cin * c2.cout * fused_ids.get(id(c2), 0) for _, c2 in pairs)
        roof(f1, M, pairs[0][0].cin, sum(c1.cout for c1, _ in pairs), 2)
        roof(f2, M, sum(c2.cin for _, c2 in pairs), sum(c2.cout for _, c2 in pairs), 2)
        for c1, c2 in pairs:
            managed.update((id(c1), id(c2)))
    for m in root.modules():
        if isinstance(m, Conv2D) and id(m) not in managed:
            Bx, Hx, Wx, _, _ = ops.geom(m._x)
            f = 2.0 * Bx * Hx * Wx * m.k * m.k * m.cin * m.cout
            fwd += f
            igemm += f if m is first else 2 * f      # the first layer needs no input gradient
            if arch == "B":
                fused += f * fused_ids.get(id(m), 0)
            Mo = Bx * Hx * Wx * (4 if getattr(m, "transposed", False) else 1)
            roof(f, 0.5 * (Bx * Hx * Wx + Mo), m.cin, m.cout, 1 if m is first else 2)
    algorithmic_flops.attainable_s = attain[0]
    algorithmic_flops.fused = fused
    return fwd, igemm, fwd


def pmc_traffic(arch, per_gpu_batch, launches_per_step):
    """HBM bytes per launch of the conv kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is; tools/pmc_traffic.py made the file from the same build as the
    kernel statistics, tools/profile_round.sh), or None.  The file is REJECTED when it does not describe the step that was just
    profiled live: other architecture / batch / image size, or another number of conv-family launches per step."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_arch{arch}.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("per_gpu_batch") != per_gpu_batch or d.get("hw") != H or d.get("arch") != arch:
                continue
            if abs(d["conv_family_launches_per_step"] - launches_per_step) > 0.01:
                log(f"pmc_traffic: {os.path.basename(path)} has {d['conv_family_launches_per_step']} conv launches per step, this build "
                    f"runs {launches_per_step}: stale, ignored")
                continue
            return {"bytes_per_launch": d["conv_family_bytes_per_step"] / max(d["conv_family_launches_per_step"], 1),
                    "bytes_per_step": d["conv_family_bytes_per_step"], "total_bytes_per_step_all_kernels": d.get("total_bytes_per_step"),
                    "source": "profiles/" + os.path.basename(path)}
        except Exception:
            continue
    return None


def pmc_mfma(arch):
    """MFMA utilisation from the committed rocprofv3 --pmc pass of the same workload (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x the
    dispatch's GRBM_GUI_ACTIVE / 8 cycles; tools/pmc_mfma.py, tools/profile_round.sh), newest round first, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_mfma_arch{arch}.json")), reverse=True):
        try:
            d = json.load(open(path))
            fam = d["families"].get("conv forward / backward-data")
            return {"conv_family": fam["mfma_busy_frac"], "weight_gradients": d["families"].get("weight gradients", {}).get("mfma_busy_frac"),
                    "whole_step": d["whole_step_mfma_busy_frac"], "source": "profiles/" + os.path.basename(path)}
        except Exception:
            continue
    return None


def _time_us(fn, n: int = 20, rounds: int = 3) -> float:
    """HIP-event time of one call: the best of `rounds` back-to-back batches of n (one batch once read 16x high right after the
    profiled eager steps of the same process)."""
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def layer_probe(dev):
    """The two 3x3 exemplars of BASELINE.md section 2, forward only, B=32: the HBM-bound stem conv 32->32 at 256x256
    (north_star's "3x3 conv fwd at 256x256 bs=32") and the MFMA-bound concats_2 256->512 at 16x16.  HIP-event timed."""
    from ultrasound_modeling_amd import ops
    from ultrasound_modeling_amd.flat import FlatParams
    from ultrasound_modeling_amd.layers import Conv2D
    out = []
    for name, B, HW, ci, co in (("stem convtmp_2 3x3 32->32 @256x256 B=32", 32, 256, 32, 32),
                                ("concats_2 3x3 256->512 @16x16 B=32", 32, 16, 256, 512)):
        layer = Conv2D(ci, co, 3)
        FlatParams(layer, dev)
        x = torch.randn(B, HW, HW, ci, device=dev).to(torch.bfloat16)
        y = ops.new_act(B, HW, HW, co, dev)
        for _ in range(3):
            layer.forward(x, out=y)
        us = _time_us(lambda: layer.forward(x, out=y))
        flops = 2.0 * B * HW * HW * 9 * ci * co
        byts = 2.0 * B * HW * HW * (ci + co) + 2.0 * 9 * ci * co
        tf = flops / us / 1e6
        roof = min(PEAK_BF16_TFLOPS, flops / byts * 8.0)          # min(MFMA peak, AI x 8 TB/s) in TFLOP/s
        out.append({"layer": name, "us": round(us, 2), "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4),
                    "ai_flop_per_byte": round(flops / byts, 1), "roofline_tflops": round(roof, 1), "frac_of_roofline": round(tf / roof, 4),
                    "gb_per_s": round(byts / us / 1e3, 1)})
    # the whole stem chain (ResNest.py:39-47: conv1 + act -> convtmp_1 + BN + act -> convtmp_2 -> BN + act -> pool) as ONE launch, B=32: the
    # fused context of north_star's "3x3 conv fwd at 256x256 bs=32".  Bytes = what a training step must move: x in (8 physical channels), the
    # three tensors the backward pass needs (y1, t1, pre-norm convtmp_2 output) and the pooled output out - no intermediate is re-read.
    B, HW = 32, 256
    c1, c2, c3 = Conv2D(8, 16, 3), Conv2D(16, 32, 3), Conv2D(32, 32, 3)
    mods = torch.nn.ModuleList([c1, c2, c3])
    FlatParams(mods, dev)
    x = torch.randn(B, HW, HW, 8, device=dev).to(torch.bfloat16)
    v32 = lambda f: torch.full((32,), f, device=dev)
    args = (x, c1.wp_f, c1.bias.data, c2.wp_f, v32(0.0), c3.wp_f, c3.bias.data, v32(1.0), v32(0.0), v32(0.0), v32(1.0), 1e-3, 0.3)
    for _ in range(3):
        ops.stem_fwd(*args)
    us = _time_us(lambda: ops.stem_fwd(*args), n=10)
    M = B * HW * HW
    flops = 2.0 * M * 9 * (1 * 16 + 16 * 32 + 32 * 32)
    byts = 2.0 * M * (8 + 16 + 32 + 32 + 32 / 4)
    tf = flops / us / 1e6
    roof = min(PEAK_BF16_TFLOPS, flops / byts * 8.0)
    out.append({"layer": "stem chain fwd (conv1 -> convtmp_1 + BN -> convtmp_2 -> BN + pool, all activations) @256x256 B=32, ONE launch, training outputs written",
                "us": round(us, 2), "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4), "ai_flop_per_byte": round(flops / byts, 1),
                "roofline_tflops": round(roof, 1), "frac_of_roofline": round(tf / roof, 4), "gb_per_s": round(byts / us / 1e3, 1)})
    # the decoder's three dilated 3x3 512->64 branches at 32x32 (Decoder.py:14-25) as ONE multi-job launch, B=32
    B, HW, ci, co = 32, 32, 512, 64
    convs = torch.nn.ModuleList([Conv2D(ci, co, 3, d) for d in (2, 4, 8)])
    FlatParams(convs, dev)
    x = torch.randn(B, HW, HW, ci, device=dev).to(torch.bfloat16)
    y = ops.new_act(B, HW, HW, 3 * co, dev)
    jobs = [(x, c.wp_f, c.bias.data, 3, c.dil, y[..., j * co:(j + 1) * co], 0, 0.0) for j, c in enumerate(convs)]
    for _ in range(3):
        ops.conv2d_fwd_multi(jobs)
    us = _time_us(lambda: ops.conv2d_fwd_multi(jobs))
    flops = 3 * 2.0 * B * HW * HW * 9 * ci * co
    byts = 2.0 * B * HW * HW * (ci + 3 * co) + 3 * 2.0 * 9 * ci * co
    tf = flops / us / 1e6
    roof = min(PEAK_BF16_TFLOPS, flops / byts * 8.0)
    out.append({"layer": "decoder b0: three dilated 3x3 512->64 (d=2,4,8) @32x32 B=32, one multi-job launch", "us": round(us, 2),
                "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4), "ai_flop_per_byte": round(flops / byts, 1),
                "roofline_tflops": round(roof, 1), "frac_of_roofline": round(tf / roof, 4), "gb_per_s": round(byts / us / 1e3, 1)})
    return out


def cpu_baseline(arch, seconds_budget: float = 25.0):
    """Time the CPU oracle (fp32, all host cores) on a bounded sample: full train steps at B=2, 256x256x1."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import usseg_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: timing the CPU oracle on {cores} threads")
    Bc = 2
    x, y = O.synthetic_batch(Bc, H, W, C_IN, seed=0, dtype=torch.float32)
    if arch == "B":
        P = O.init_vision_transformer_params(channel=C_IN, seed=0, dtype=torch.float32)
        st = {}
        step = lambda: O.train_step(x, y, P, st, Bc)
    else:
        P = O.init_archA_params(channel=C_IN, radix=3, kpaths=4, seed=0, dtype=torch.float32)
        names = O.trainable_names(P)
        m, v, it = [torch.zeros_like(P[n]) for n in names], [torch.zeros_like(P[n]) for n in names], [0]
        gen = torch.Generator().manual_seed(0)

        def step():      # TBI_ResNest.py:35-55: forward, my_loss_cat map, gradient of its sum, plain Adam(5e-3)
            leaves = [P[n].detach().clone().requires_grad_(True) for n in names]
            Pl = dict(P)
            Pl.update(zip(names, leaves))
            masks = [(torch.rand(Bc, H // 64 * 2 ** (i + 1), W // 64 * 2 ** (i + 1), 512, generator=gen) > 0.5).float() for i in range(3)]
            probs = O.archA_forward(x, Pl, 3, 4, dropout_masks=masks)
            grads = torch.autograd.grad(O.my_loss_cat(y, probs, H, W).sum(), leaves)
            it[0] += 1
            with torch.no_grad():
                new = [l.detach().clone() for l in leaves]
                O.adam_step(new, grads, m, v, it[0], 5e-3)
                for n, t in zip(names, new):
                    P[n] = t
    step()            # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 8:
            break
    return {"value": round(Bc * n / el, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} full train steps (fwd+bwd+{'clip+' if arch == 'B' else ''}Adam) of the Arch {arch} oracle, fp32 PyTorch-CPU, B={Bc}, 256x256x1, "
                      f"{torch.get_num_threads()} threads; TensorFlow (the reference) is not installable here"}


def launch_ranks(n: int) -> int:
    """`bench.py --gpus N` started by hand (no torchrun environment): start the N ranks as a child torch.distributed.run and
    relay its stdout / stderr / exit code.  Called before any torch.cuda / library call, so no GPU process is ever exec'd."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    return subprocess.run(cmd, env=env).returncode


def make_dry_net(global_batch, dev):
    """--dry-run: a per-pixel 1->3 affine + softmax + the Arch B loss on the CPU behind the product's step driver.  It exists to
    rehearse bench.py's rank launch, batch split, exchange, timing and JSON line where there is no GPU (tests/test_cpu_host.py)."""
    from ultrasound_modeling_amd.step import TrainStepDriver

    class _Flat:
        def __init__(self):
            self.flat = torch.tensor([0.3, -0.2, 0.1, 0.0, 0.1, -0.1], device=dev)
            self.grad = torch.zeros(6, device=dev)
            self.total = self.n_trainable = 6

    class _Adam:
        def __init__(self, flat):
            self.flat, self.m, self.v, self.t = flat, torch.zeros(6), torch.zeros(6), 0

        def _scale(self):
            return 1.0 / max(float(self.flat.grad.norm()), 1.0)

        def clip_local(self):
            self.flat.grad.mul_(self._scale())

        def advance(self):
            self.t += 1

        def apply_range(self, lo, hi, clip=0.0):
            g = self.flat.grad[lo:hi] * (self._scale() if clip > 0 else 1.0)
            self.m[lo:hi].mul_(0.9).add_(g, alpha=0.1)
            self.v[lo:hi].mul_(0.999).addcmul_(g, g, value=0.001)
            lr_t = 1e-3 * (1 - 0.999 ** self.t) ** 0.5 / (1 - 0.9 ** self.t)
            self.flat.flat[lo:hi].sub_(lr_t * self.m[lo:hi] / (self.v[lo:hi].sqrt() + 1e-7))

        def apply(self, already_clipped=False):
            self.advance()
            self.apply_range(0, 6, 0.0 if already_clipped else 1.0)

    class DryNet(TrainStepDriver):
        def __init__(self):
            self.flat = _Flat()
            self.optimizer = _Adam(self.flat)
            self._buffers = {}

        def modules(self):
            return [self]

        def _zero_grad(self):
            self.flat.grad.zero_()

        def _forward_backward(self, x, y):
            p = self.flat.flat.clone().requires_grad_(True)
            probs = torch.softmax(x * p[:3] + p[3:], dim=-1)
            ys = 0.9 * y + 0.1 / 3
            self.loss = -(ys * probs.clamp(1e-7, 1 - 1e-7).log()).sum() / global_batch
            self.flat.grad.add_(torch.autograd.grad(self.loss, p)[0])
            return probs.detach()

        def _repack(self):
            pass

        def train_step(self, x, y):
            probs = self._train_body(x, y)
            return self.loss.detach(), probs
    return DryNet()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", choices=["B", "A", "T", "S"], default="B",
                    help="B: ResNest.py+Decoder.py (configs[1], default); A: TBI_ResNest.py (configs[2]); T: TBI_TransUNet.py at 512x512 (configs[3], throughput only)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: fixed per-GPU batch; strong: fixed global batch split over the replicas")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the data-parallel code path (RCCL all-reduce around the update) even at N=1")
    ap.add_argument("--dp-chunks", type=int, default=0, help="pieces the gradient exchange is pipelined in (0 = automatic)")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra eager steps with per-launch HIP events (roofline leg)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo rehearsal of the launch + exchange + reporting plumbing with a stand-in model")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # started by hand: become the launcher (nothing has touched the GPU yet)
        sys.exit(launch_ranks(args.gpus))

    from ultrasound_modeling_amd.MainParallel import MirroredTrainer, init_distributed
    dry = args.dry_run
    rank, world, local = init_distributed("gloo" if dry else None)
    assert world == max(args.gpus, 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("USSEG_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank that dies must fail the whole launch
        raise SystemExit(3)
    if dry:
        dev = torch.device("cpu")
        args.no_graph, args.profile_steps, args.no_cpu_baseline = True, 0, True
    else:
        if os.environ.get("USSEG_BENCH_SHARE_GPU") == "1":      # rehearsal hook (with USSEG_DIST_BACKEND=gloo): more ranks than GPUs
            local %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    arch = args.arch
    base = BASE_BATCH[arch]
    global H, W
    if arch in ("T", "S"):           # configs[3] / [4] are quoted at 512x512; the roofline / CPU legs are defined for the headline configs only
        H = W = 512
        args.profile_steps, args.no_cpu_baseline = 0, True
    if args.scaling == "strong":
        assert base % world == 0, f"strong scaling: the global batch {base} must divide over {world} replicas"
        per_gpu = base // world
    else:
        per_gpu = base
    global_batch = per_gpu * world

    from ultrasound_modeling_amd import _lib
    if dry:
        net = make_dry_net(global_batch, dev)
    elif arch == "B":
        from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
        net = VisionTransformer(batch_size=global_batch, img_size=(H, W), in_channels=C_IN, device=str(dev), seed=0)
    elif arch == "T":
        from ultrasound_modeling_amd.TBI_TransUNet import VisionTransformer as TransUNet
        net = TransUNet(img_size=(H, W), batch_size=global_batch, in_channels=C_IN, device=str(dev), seed=0)
    elif arch == "S":
        from ultrasound_modeling_amd.SwinTransformer import SwinTransformerModel
        net = SwinTransformerModel(model_name="swin_tiny_512", img_size=(H, W), patch_size=(4, 4), in_chans=C_IN, embed_dim=96, depths=[2, 2, 6, 2],
                                   num_heads=[3, 6, 12, 24], window_size=8, device=str(dev), seed=0)
    else:
        from ultrasound_modeling_amd.TBI_ResNest import ResNest
        net = ResNest(H, W, C_IN, 3, ksize=3, radix=3, kpaths=4, learning_rate=5e-3, device=str(dev), seed=0)
    if args.force_dist and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    trainer = MirroredTrainer(net, force=args.force_dist, chunks=args.dp_chunks or None)

    g = torch.Generator().manual_seed(rank)
    x = torch.randn(per_gpu, H, W, C_IN, generator=g).clamp_(-1, 1).to(dev)
    lab = torch.rand(per_gpu, H // 16, W // 16, generator=g)
    lab = (lab > 0.70).float() + (lab > 0.95).float()
    lab = (lab + 0.9 * torch.rand(lab.shape, generator=g) * (lab >= 1)).repeat_interleave(16, 1).repeat_interleave(16, 2)
    c2 = torch.where(lab >= 1.05, (lab - 1).clamp(0, 1), torch.zeros_like(lab))
    y = torch.stack([(lab <= 0.95).float(), torch.where(lab > 0.95, 1 - c2, torch.zeros_like(lab)), c2], dim=-1).to(dev)

    if arch == "S":      # the reference defines no loss for SwinTransformer.py: the step is driven by a fixed upstream gradient of the pooled output
        y = torch.full((per_gpu, 768), 1.0 / (768 * global_batch), device=dev)
    use_graph = not args.no_graph                    # N > 1: the step up to the per-replica clip is one graph, then the RCCL exchange, then the update
    if dry:
        x, y = x[..., :1].float(), y.float()
    log(f"rank {rank}/{world}: Arch {arch} built ({net.flat.n_trainable} params), {per_gpu} images per GPU, warming up (graph={use_graph})")
    for _ in range(max(args.warmup, 1) if not use_graph else 1):
        trainer.train_step(x, y)
    if use_graph:
        net.capture_graph(x, y)
        gx, gy = net.graph_inputs()           # the synthetic batch lives in the captured step's own input buffers (inputs resident in HBM;
        if gx.shape == x.shape and gx.dtype == x.dtype and gy.shape == y.shape and gy.dtype == y.dtype:
            gx.copy_(x); gy.copy_(y)          # a real loader - ultrasound_modeling_amd/Dataset_2.py - writes each batch there)
            x, y = gx, gy
        for _ in range(args.warmup):
            trainer.train_step(x, y)

    def sync():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.train_step(x, y)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    loss_val = float(loss.item())
    log(f"timed region done: {el / args.steps * 1e3:.3f} ms/step")

    # ---- roofline leg: eager steps with the library's per-launch HIP events around the igemm / wgrad kernels
    roofline = None
    if rank == 0 and args.profile_steps > 0:
        lib = _lib.load()
        net._graph_saved, net._graph = net._graph, None
        sync_saved, net.grad_sync = net.grad_sync, None     # kernel timing only: no collective inside the profiled steps
        P = args.profile_steps
        from ultrasound_modeling_amd import ops as _ops
        side_saved, _ops._Side.enabled = _ops._Side.enabled, False   # per-kernel durations: no weight-gradient launches running beside them
        lazy_saved, _ops._LAZY = _ops._LAZY, False
        xp, yp = net._prep_x(x), net._prep_y(y)
        _lib.check(lib.usseg_prof_enable(7, 4096 * P), "prof_enable")
        for _ in range(P):
            net._train_body(xp, yp)
        torch.cuda.synchronize()
        _ops._Side.enabled, _ops._LAZY = side_saved, lazy_saved
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.usseg_prof_read(1, ctypes.byref(ms), ctypes.byref(n)), "prof_read")
        ig_ms, ig_n = ms.value / P, n.value // P
        _lib.check(lib.usseg_prof_read(2, ctypes.byref(ms), ctypes.byref(n)), "prof_read")
        wg_ms, wg_n = ms.value / P, n.value // P
        _lib.check(lib.usseg_prof_read(4, ctypes.byref(ms), ctypes.byref(n)), "prof_read")
        fu_ms, fu_n = ms.value / P, n.value // P
        lib.usseg_prof_disable()
        fwd_f, ig_f, wg_f = algorithmic_flops(net, arch)
        net._graph, net.grad_sync = net._graph_saved, sync_saved
        fu_f = algorithmic_flops.fused
        cv_f = ig_f - fu_f                      # FLOPs of the plain conv launches (what `frac` prices: comparable with rounds 1-2)
        achieved = cv_f / (ig_ms * 1e-3) / 1e12
        roofline = {"kernel": "conv_stream_kernel / conv_big_kernel / igemm_dma_kernel / igemm_kernel / conv_halo_kernel: every plain conv + tconv forward and "
                              "backward-data launch (single stream).  The fused tile kernels (cardinal_fwd / cardinal_bwd / stem_fwd: convs + LayerNorm / "
                              "BatchNorm + activation in one launch) are timed separately in `fused_tiles`; `family` is both together",
                    "bound": "mfma", "achieved": round(achieved, 2),
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                    "launches_per_step": ig_n, "avg_launch_us": round(ig_ms * 1e3 / max(ig_n, 1), 2),
                    "algorithmic_gflop_per_step": round(cv_f / 1e9, 2), "kernel_ms_per_step": round(ig_ms, 3),
                    "fused_tiles": {"launches_per_step": fu_n, "kernel_ms_per_step": round(fu_ms, 3), "algorithmic_gflop_per_step": round(fu_f / 1e9, 2),
                                    "achieved": round(fu_f / max(fu_ms * 1e-3, 1e-12) / 1e12, 2),
                                    "note": "conv FLOPs only; these launches also do the norms / activations / pools that were separate passes"},
                    "family": {"launches_per_step": ig_n + fu_n, "kernel_ms_per_step": round(ig_ms + fu_ms, 3), "algorithmic_gflop_per_step": round(ig_f / 1e9, 2),
                               "achieved": round(ig_f / ((ig_ms + fu_ms) * 1e-3) / 1e12, 2)},
                    "wgrad": {"launches_per_step": wg_n, "kernel_ms_per_step": round(wg_ms, 3),
                              "achieved": round(wg_f / (wg_ms * 1e-3) / 1e12, 2), "algorithmic_gflop_per_step": round(wg_f / 1e9, 2)},
                    "fwd_gflop_per_image": round(fwd_f / per_gpu / 1e9, 3),
                    # time-weighted: what the same launches would take if each ran on its own roofline (HBM-bound stem /
                    # stage-1 layers priced on bytes, the deep ones on FLOPs) over what they took
                    "attainable_ms_per_step": round(algorithmic_flops.attainable_s * 1e3, 3),
                    "frac_of_attainable": round(algorithmic_flops.attainable_s * 1e3 / max(ig_ms + fu_ms, 1e-9), 4)}
        roofline["traffic"] = pmc_traffic(arch, per_gpu, ig_n + fu_n)
        roofline["mfma_busy_frac"] = pmc_mfma(arch)
        roofline["layers"] = layer_probe(dev)

    if rank == 0:
        names = {"B": "BASELINE configs[1]: Arch B (ResNest.py r=3,k=3 + Decoder.py, no ViT) train step",
                 "A": "BASELINE configs[2]: Arch A (TBI_ResNest.py model r=3,k=4, my_loss_cat, Adam 5e-3) train step",
                 "T": "BASELINE configs[3]: TBI_TransUNet.py (ResNeSt encoder with BatchNorm + 8-layer ViT bottleneck, 1024 tokens + decoder) train step",
                 "S": "BASELINE configs[4]: SwinTransformer.py windowed-attention encoder (swin_tiny widths, 8x8 windows; forward + backward from a fixed "
                      "upstream gradient of the pooled output + Adam; the reference defines no loss for it)"}
        out = {"metric": f"segmentation training images/sec at {H}x{W}", "value": round(global_batch * args.steps / el, 2),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"{names[arch]}, {H}x{W}x1, {per_gpu} images per GPU, bf16 MFMA / fp32 accumulate"
                                      f"{' (BASELINE names fp16 for this config; the reference itself is fp32)' if arch == 'S' else ''}, fp32 master weights + Adam",
                          "arch": arch, "per_gpu_batch": per_gpu, "global_batch": global_batch, "parallelism": f"dp{world}",
                          "hip_graph": bool(use_graph), "dp_exchange_chunks": getattr(net.grad_sync, "nchunks", 0) if net.grad_sync is not None else 0,
                          "final_loss": round(loss_val, 4)},
               "roofline": roofline}
        if dry:
            out.update({"dry_run": True, "data": "synthetic (dry run: CPU stand-in model over gloo, the value measures nothing)", "dtype": "f32"})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arch)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
