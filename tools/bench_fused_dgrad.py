"""Fused four-branch backward-data (Decoder stage) timing in isolation: python tools/bench_fused_dgrad.py [B]"""
import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# name, H, Cin (dx channels), q (channels per branch)
CASES = [("b0 s1 dx512 q64 @32", 32, 512, 64), ("b0 s2 dx256 q64 @32", 32, 256, 64), ("b1 s1 dx256 q32 @64", 64, 256, 32),
         ("b1 s2 dx128 q32 @64", 64, 128, 32), ("b2 s1 dx128 q16 @128", 128, 128, 16), ("b2 s2 dx64 q16 @128", 128, 64, 16)]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * reps)


for name, H, Cin, q in CASES:
    draw = torch.randn(B, H, H, 4 * q, device=dev).to(torch.bfloat16)
    dx = torch.empty(B, H, H, Cin, device=dev, dtype=torch.bfloat16)
    wcat = (torch.randn((Cin + 15) // 16 * 16, 28 * q, device=dev) * 0.02).to(torch.bfloat16)
    t = timeit(lambda: ops.conv2d_dgrad_branches(draw, wcat, [1, 3, 3, 3], [1, 2, 4, 8], [0, q, 2 * q, 3 * q], q, dx))
    gf = 2.0 * B * H * H * 28 * q * Cin / 1e9
    mb = (draw.numel() + dx.numel()) * 2 / 1e6
    print(f"{name:24s} {gf:6.2f} GF {mb:6.1f} MB  {t*1e6:7.1f} us {gf/t/1e3:7.1f} TF/s  {mb/t/1e6:6.2f} TB/s", flush=True)
