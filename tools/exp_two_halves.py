"""Experiment: is there idle capacity that two concurrent half-batch steps could use?  Builds one Arch B net at batch 16 and two at batch 8,
captures each training step as a HIP graph, and times (a) the 16-image step, (b) one 8-image step alone, (c) two 8-image steps replayed
on two streams at the same time.  (c) < (a) would mean that pipelining two micro-batches through one GPU pays.
python tools/exp_two_halves.py"""
import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd.VisionTransformer import VisionTransformer

dev = "cuda"
H = W = 256


def batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, H, W, 1, generator=g).clamp_(-1, 1).to(dev)
    lab = (torch.rand(n, H, W, generator=g) * 3).long().clamp(0, 2)
    y = torch.nn.functional.one_hot(lab, 3).float().to(dev)
    return x, y


def make(n, seed):
    net = VisionTransformer(batch_size=16, img_size=(H, W), in_channels=1, device=dev, seed=0)      # loss / global batch 16 either way
    x, y = batch(n, seed)
    net.train_step(x, y)
    net.capture_graph(x, y)
    return net, x, y


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full, xf, yf = make(16, 1)
a, xa, ya = make(8, 2)
b, xb, yb = make(8, 3)
print(f"16-image step: {timeit(lambda: full._graph[0].replay()):.3f} ms")
print(f" 8-image step alone: {timeit(lambda: a._graph[0].replay()):.3f} ms")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        a._graph[0].replay()
    with torch.cuda.stream(s2):
        b._graph[0].replay()


print(f"two 8-image steps on two streams: {timeit(both):.3f} ms per pair")
