"""Eager timing of the Arch A (TBI_ResNest.py) train step (diagnostic): python tools/time_archA.py B HW"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ultrasound_modeling_amd.TBI_ResNest import ResNest
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 256
net = ResNest(HW, HW, 1, 3, ksize=3, radix=3, kpaths=4, learning_rate=5e-3)
x = torch.randn(B, HW, HW, 1, device="cuda").clamp_(-1, 1)
y = torch.softmax(torch.randn(B, HW, HW, 3, device="cuda"), -1)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lm, acc, _ = net.step(x, y, train=True)
    torch.cuda.synchronize()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms  loss {lm.sum().item():.4f}", flush=True)
print(f"images/s {B / (time.perf_counter() - t0):.1f}")
if len(sys.argv) > 3 and sys.argv[3] == "graph":
    net.capture_graph(x, y)
    ts = []
    for i in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lm, acc, _ = net.step(x, y, train=True)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"graph: min {min(ts) * 1e3:.2f} ms  loss {lm.sum().item():.4f}  images/s {B / min(ts):.1f}")
