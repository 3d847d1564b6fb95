"""Per-launch roofline table of one eager training step: every conv / weight-gradient host call is bracketed by HIP events on the
launch stream (torch's current stream) and priced with the FLOPs and minimum bytes of its shapes (PHYSICAL channel counts).

usage: python tools/launch_table.py [B|A|T] [reps]      (minimum over `reps` steps per call site; eager, so small launches carry a
few us of event overhead - use it to rank launches, not to quote them)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ultrasound_modeling_amd import ops  # noqa: E402

PEAK_TF, PEAK_TB = 2500.0, 8.0
rec, order = {}, []
_idx = [0]


def _time(name, shape, flops, nbytes, fn, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*a, **k)
    e1.record()
    i = _idx[0]
    _idx[0] += 1
    if i >= len(order):
        order.append((name, shape, flops, nbytes))
    rec.setdefault(i, []).append((e0, e1))
    return r


def g(t):
    return ops.geom(t)


def wrap():
    o = {n: getattr(ops, n) for n in dir(ops)}

    def conv2d_fwd(x, wp, bias, ksize, dilation, out, *a, **k):
        B, H, W, Ci, _ = g(x)
        Co = g(out)[3]
        M = B * H * W
        return _time("fwd", f"{ksize}x{ksize} d{dilation} {Ci}->{Co} @{H}x{W}", 2 * M * ksize * ksize * Ci * Co, 2 * M * (Ci + Co), o["conv2d_fwd"], x, wp, bias,
                     ksize, dilation, out, *a, **k)

    def conv2d_dgrad(dy, wp_d, ksize, dilation, dx, *a, **k):
        B, H, W, Co, _ = g(dy)
        Ci = g(dx)[3]
        M = B * H * W
        return _time("dgrad", f"{ksize}x{ksize} d{dilation} {Co}->{Ci} @{H}x{W}", 2 * M * ksize * ksize * Ci * Co, 2 * M * (Ci + Co), o["conv2d_dgrad"], dy, wp_d,
                     ksize, dilation, dx, *a, **k)

    def conv2d_fwd_multi(jobs):
        fl = by = 0
        for j in jobs:
            B, H, W, Ci, _ = g(j[0])
            Co = g(j[5])[3]
            fl += 2 * B * H * W * j[3] ** 2 * Ci * Co
            by += 2 * B * H * W * (Ci + Co)
        return _time("fwd_multi", f"{len(jobs)} jobs {j[3]}x{j[3]} {Ci}->{Co} @{H}x{W}", fl, by, o["conv2d_fwd_multi"], jobs)

    def conv2d_dgrad_multi(jobs):
        fl = by = 0
        for j in jobs:
            B, H, W, Co, _ = g(j[0])
            Ci = g(j[4])[3]
            fl += 2 * B * H * W * j[2] ** 2 * Ci * Co
            by += 2 * B * H * W * (Ci + Co)
        return _time("dgrad_multi", f"{len(jobs)} jobs {j[2]}x{j[2]} {Co}->{Ci} @{H}x{W}", fl, by, o["conv2d_dgrad_multi"], jobs)

    def conv2d_dgrad_branches(dy_cat, wp_cat, ksizes, dilations, ch_offs, Cb, dx, *a, **k):
        B, H, W, _, _ = g(dy_cat)
        Ci = g(dx)[3]
        M = B * H * W
        taps = sum(kk * kk for kk in ksizes)
        return _time("dgrad_branches", f"{len(ksizes)} branches {Cb}->{Ci} @{H}x{W}", 2 * M * taps * Ci * Cb, 2 * M * (Ci + len(ksizes) * Cb),
                     o["conv2d_dgrad_branches"], dy_cat, wp_cat, ksizes, dilations, ch_offs, Cb, dx, *a, **k)

    def _wg(name, key):
        def f(x, dy, ksize, *a, **k):
            B, H, W, Ci, _ = g(x)
            Co = g(dy)[3]
            M = B * H * W
            return _time(name, f"{ksize}x{ksize} {Ci}->{Co} @{H}x{W}", 2 * M * ksize * ksize * Ci * Co, 2 * M * (Ci + Co), o[key], x, dy, ksize, *a, **k)
        return f

    def conv2d_wgrad_multi(jobs):
        fl = by = 0
        for j in jobs:
            B, H, W, Ci, _ = g(j[0])
            Co = g(j[1])[3]
            fl += 2 * B * H * W * j[2] ** 2 * Ci * Co
            by += 2 * B * H * W * (Ci + Co)
        return _time("wgrad_multi", f"{len(jobs)} jobs {j[2]}x{j[2]} {Ci}->{Co} @{H}x{W}", fl, by, o["conv2d_wgrad_multi"], jobs)

    def tconv2d_fwd(x, wp, bias, ksize, out, *a, **k):
        B, H, W, Ci, _ = g(x)
        Co = g(out)[3]
        M = B * H * W
        return _time("tconv_fwd", f"{ksize}x{ksize} s2 {Ci}->{Co} @{H}x{W}", 2 * M * ksize * ksize * Ci * Co, 2 * M * (Ci + 4 * Co), o["tconv2d_fwd"], x, wp, bias,
                     ksize, out, *a, **k)

    def tconv2d_dgrad(dy, wp_d, ksize, dx, *a, **k):
        B, H, W, Ci, _ = g(dx)
        Co = g(dy)[3]
        M = B * H * W
        return _time("tconv_dgrad", f"{ksize}x{ksize} s2 {Co}->{Ci} @{H}x{W}", 2 * M * ksize * ksize * Ci * Co, 2 * M * (Ci + 4 * Co), o["tconv2d_dgrad"], dy, wp_d,
                     ksize, dx, *a, **k)

    def _twg(name, key):
        def f(x, dy, ksize, *a, **k):
            B, H, W, Ci, _ = g(x)
            Co = g(dy)[3]
            M = B * H * W
            return _time(name, f"{ksize}x{ksize} s2 {Ci}->{Co} @{H}x{W}", 2 * M * ksize * ksize * Ci * Co, 2 * M * (Ci + 4 * Co), o[key], x, dy, ksize, *a, **k)
        return f

    ops.conv2d_fwd, ops.conv2d_dgrad, ops.conv2d_fwd_multi, ops.conv2d_dgrad_multi = conv2d_fwd, conv2d_dgrad, conv2d_fwd_multi, conv2d_dgrad_multi
    ops.conv2d_dgrad_branches, ops.conv2d_wgrad_multi = conv2d_dgrad_branches, conv2d_wgrad_multi
    ops.conv2d_wgrad, ops.conv2d_wgrad_mapped = _wg("wgrad", "conv2d_wgrad"), _wg("wgrad", "conv2d_wgrad_mapped")
    ops.tconv2d_fwd, ops.tconv2d_dgrad = tconv2d_fwd, tconv2d_dgrad
    ops.tconv2d_wgrad, ops.tconv2d_wgrad_mapped = _twg("tconv_wgrad", "tconv2d_wgrad"), _twg("tconv_wgrad", "tconv2d_wgrad_mapped")


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "B"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    torch.cuda.set_device(0)
    per_gpu = bench.BASE_BATCH[arch]
    HW = 512 if arch == "T" else 256
    if arch == "B":
        from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
        net = VisionTransformer(batch_size=per_gpu, img_size=(HW, HW), in_channels=1, device="cuda:0", seed=0)
    elif arch == "T":
        from ultrasound_modeling_amd.TBI_TransUNet import VisionTransformer as TransUNet
        net = TransUNet(img_size=(HW, HW), batch_size=per_gpu, in_channels=1, device="cuda:0", seed=0)
    else:
        from ultrasound_modeling_amd.TBI_ResNest import ResNest
        net = ResNest(HW, HW, 1, 3, ksize=3, radix=3, kpaths=4, learning_rate=5e-3, device="cuda:0", seed=0)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(per_gpu, HW, HW, 1, generator=gen).clamp_(-1, 1).cuda()
    lab = torch.randint(0, 3, (per_gpu, HW, HW), generator=gen)
    y = torch.nn.functional.one_hot(lab, 3).float().cuda()
    step = lambda: net.train_step(x, y) if hasattr(net, "train_step") else net.step(x, y, train=True)
    for _ in range(2):
        step()
    wrap()
    for _ in range(reps):
        _idx[0] = 0
        step()
    torch.cuda.synchronize()
    print(f"# arch {arch}, {per_gpu} images, eager; min of {reps} event-timed calls; physical channel counts")
    print(f"# {'op':15s} {'shape':42s} {'us':>7s} {'GFLOP':>7s} {'TF/s':>6s} {'MB':>6s} {'TB/s':>5s}  bound  frac")
    tot = {}
    for i, (name, shape, fl, by) in enumerate(order):
        us = min(a.elapsed_time(b) for a, b in rec[i]) * 1e3
        tf, tb = fl / us / 1e6, by / us / 1e6
        t_m, t_h = fl / PEAK_TF / 1e6, by / PEAK_TB / 1e6
        bound, frac = ("mfma", t_m / us) if t_m >= t_h else ("hbm", t_h / us)
        print(f"  {name:15s} {shape:42s} {us:7.1f} {fl / 1e9:7.2f} {tf:6.0f} {by / 1e6:6.1f} {tb:5.2f}  {bound:5s} {frac:5.2f}")
        t = tot.setdefault(name, [0.0, 0.0, 0.0])
        t[0] += us
        t[1] += fl
        t[2] += max(t_m, t_h)
    for name, (us, fl, att) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"# {name:15s} {us:8.1f} us  {fl / 1e9:8.1f} GFLOP  {fl / us / 1e6:6.0f} TF/s  attainable {att:7.1f} us ({att / us:4.2f})")


if __name__ == "__main__":
    main()
