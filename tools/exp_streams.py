"""Experiment: do independent conv launches captured on forked streams overlap inside one HIP graph?"""
import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops

dev = torch.device("cuda:0")
B, H, Cin, Cb = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 512, int(sys.argv[3]) if len(sys.argv) > 3 else 64
x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
out = torch.empty(B, H, H, 4 * Cb, device=dev, dtype=torch.bfloat16)
ws = [torch.randn(max(16, Cb), 9 * Cin, device=dev).to(torch.bfloat16) * 0.02 for _ in range(3)]
w1 = torch.randn(max(16, Cb), Cin, device=dev).to(torch.bfloat16) * 0.02
bias = torch.zeros(Cb, device=dev)
side = [torch.cuda.Stream() for _ in range(3)]


def seq():
    ops.conv2d_fwd(x, w1, bias, 1, 1, out[..., 0:Cb])
    for i, d in enumerate((2, 4, 8)):
        ops.conv2d_fwd(x, ws[i], bias, 3, d, out[..., (i + 1) * Cb:(i + 2) * Cb])


def forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    evs = []
    for i, d in enumerate((2, 4, 8)):
        s = side[i]
        s.wait_event(ev)
        with torch.cuda.stream(s):
            ops.conv2d_fwd(x, ws[i], bias, 3, d, out[..., (i + 1) * Cb:(i + 2) * Cb])
            e = torch.cuda.Event()
            e.record(s)
            evs.append(e)
    ops.conv2d_fwd(x, w1, bias, 1, 1, out[..., 0:Cb])
    for e in evs:
        main.wait_event(e)


def bench(fn, name):
    fn(); torch.cuda.synchronize()
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs):
            for _ in range(10):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{name}: {dt * 1e6:.1f} us per 4-branch group; same={torch.equal(ref, out)}")


bench(seq, "sequential")
bench(forked, "forked   ")
