"""Per-layer 3x3 conv timing (forward and backward-data) on the Arch B shapes: python tools/bench_conv.py [B]
Set USSEG_BIG=0 / USSEG_NO_HALO=1 to time the other kernels on the same shapes."""
import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# name, H, Cin, Cout, dilation
LAYERS = [
    ("concats_2 s1 32->64 @128", 128, 32, 64, 1), ("concats_2 s2 64->128 @64", 64, 64, 128, 1),
    ("concats_2 s3 128->256 @32", 32, 128, 256, 1), ("concats_2 s4 256->512 @16", 16, 256, 512, 1),
    ("conv_more 512->256 @16", 16, 512, 256, 1),
    ("b0 512->64 d2 @32", 32, 512, 64, 2), ("b0 512->64 d4 @32", 32, 512, 64, 4), ("b0 512->64 d8 @32", 32, 512, 64, 8),
    ("b0' 256->64 d2 @32", 32, 256, 64, 2), ("b0' 256->64 d8 @32", 32, 256, 64, 8),
    ("b1 256->32 d2 @64", 64, 256, 32, 2), ("b1 256->32 d8 @64", 64, 256, 32, 8), ("b1' 128->32 d4 @64", 64, 128, 32, 4),
    ("b2 128->16 d2 @128", 128, 128, 16, 2), ("b2' 64->16 d8 @128", 128, 64, 16, 8),
]
import os
if os.environ.get("STEM"):
    LAYERS = [("stem 8->32 @256", 256, 8, 32, 1), ("stem 32->32 @256", 256, 32, 32, 1), ("stem 32->64 @256", 256, 32, 64, 1),
              ("s1 32->32 @128", 128, 32, 32, 1), ("s1 32->16 @128", 128, 32, 16, 1)]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * reps)


for name, H, Cin, Cout, d in LAYERS:
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    wf = (torch.randn(max(16, (Cout + 15) // 16 * 16), 9 * Cin, device=dev) * 0.02).to(torch.bfloat16)
    wd = (torch.randn(max(16, (Cin + 15) // 16 * 16), 9 * Cout, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.zeros(Cout, device=dev)
    y = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(B, H, H, Cin, device=dev, dtype=torch.bfloat16)
    gf = 2.0 * B * H * H * 9 * Cin * Cout / 1e9
    tf = timeit(lambda: ops.conv2d_fwd(x, wf, bias, 3, d, y))
    tb = timeit(lambda: ops.conv2d_dgrad(dy, wd, 3, d, dx))
    print(f"{name:28s} {gf:6.2f} GF  fwd {tf*1e6:7.1f} us {gf/tf/1e3:7.1f} TF/s   dgrad {tb*1e6:7.1f} us {gf/tb/1e3:7.1f} TF/s", flush=True)
