"""Timing of the full reference model (Arch B WITH the ViT bottleneck): python tools/time_vit.py B H W C"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
B, H, W, C = (int(v) for v in (sys.argv[1:5] + ["16", "256", "256", "1"][len(sys.argv) - 1:]))
net = VisionTransformer(batch_size=B, img_size=(H, W), in_channels=C, use_vit=True)
x = torch.randn(B, H, W, C, device="cuda").clamp_(-1, 1)
y = torch.softmax(torch.randn(B, H, W, 3, device="cuda"), -1)
net.train_step(x, y)
net.capture_graph(x, y)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _ = net.train_step(x, y)
    torch.cuda.synchronize()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms loss {loss.item():.2f}", flush=True)
print(f"images/s {B / (time.perf_counter() - t0):.1f} ({net.flat.n_trainable} params)")
