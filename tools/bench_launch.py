import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
dev = torch.device("cuda:0")
x = torch.zeros(16, device=dev)
big = torch.zeros(64 << 20, device=dev)   # 256 MB
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (10 * reps)
print("trivial fill (1 WG):        %.2f us per launch" % (timeit(lambda: ops.fill_f32(x, 0.0), 200) * 1e6))
y = torch.zeros(1 << 20, device=dev)
print("fill 4 MB:                  %.2f us per launch" % (timeit(lambda: ops.fill_f32(y, 0.0), 200) * 1e6))
print("torch zero_ 64 B:           %.2f us per launch" % (timeit(lambda: x.zero_(), 200) * 1e6))
