"""Norm fwd/bwd bandwidth on the Arch B shapes: python tools/bench_norm.py"""
import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
dev = torch.device("cuda:0")
# name, M, C, Cphys, G, mode
CASES = [("stem BN 32 @256^2", 16 * 256 * 256, 32, 32, 1, 1), ("s1 LN1 9(3g)", 16 * 128 * 128, 9, 16, 3, 0), ("s1 LN2 30(3g)", 16 * 128 * 128, 30, 32, 3, 0),
         ("s1 sc LN 64", 16 * 128 * 128, 64, 64, 1, 0), ("b2 BN 64 @128^2", 16 * 128 * 128, 64, 64, 1, 1), ("s2 LN2 63(3g)", 16 * 64 * 64, 63, 64, 3, 0),
         ("s2 sc LN 128", 16 * 64 * 64, 128, 128, 1, 0), ("b1 BN 128 @64^2", 16 * 64 * 64, 128, 128, 1, 1), ("s3 sc LN 256", 16 * 32 * 32, 256, 256, 1, 0),
         ("b0 BN 256 @32^2", 16 * 32 * 32, 256, 256, 1, 1), ("s4 sc LN 512", 16 * 16 * 16, 512, 512, 1, 0)]
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
for name, M, C, Cp, G, mode in CASES:
    x = torch.randn(1, 1, M, Cp, device=dev).to(torch.bfloat16); dy = torch.randn_like(x.float()).to(torch.bfloat16)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    ga, be = torch.ones(Cp, device=dev), torch.zeros(Cp, device=dev)
    mean, var = torch.zeros(Cp, device=dev), torch.ones(Cp, device=dev)
    dg, db, dbi = torch.zeros(Cp, device=dev), torch.zeros(Cp, device=dev), torch.zeros(Cp, device=dev)
    tf = timeit(lambda: ops.norm_act_fwd(x, C, ga, be, y, mode, G, 1e-3, 1, 0.3, mean, var))
    tb = timeit(lambda: ops.norm_act_bwd(x, dy, C, ga, be, dx, dg, db, mode, G, 1e-3, 1, 0.3, mean, var, dbias=dbi))
    by = M * Cp * 2
    print(f"{name:20s} {by/1e6:7.1f} MB  fwd {tf*1e6:6.1f} us {2*by/tf/1e9:7.0f} GB/s   bwd(+finish) {tb*1e6:6.1f} us {3*by/tb/1e9:7.0f} GB/s", flush=True)
