"""Throughput of the windowed-attention encoder (SwinTransformer.py, BASELINE configs[4]: 512x512, batch 16 per GPU, swin_tiny widths,
8x8 windows): forward + backward + Adam of the encoder, eager launches.  The reference defines no loss / decoder for this file (nothing
imports it), so the upstream gradient is a fixed tensor.  usage: python tools/time_swin.py [steps]"""
import sys, time, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd.SwinTransformer import SwinTransformerModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
net = SwinTransformerModel(model_name="swin_tiny_512", img_size=(512, 512), patch_size=(4, 4), in_chans=1, embed_dim=96, depths=[2, 2, 6, 2],
                           num_heads=[3, 6, 12, 24], window_size=8, seed=0)
x = torch.randn(16, 512, 512, 1).cuda()
g = (torch.ones(16, 768) / 768).cuda()


def step():
    net.train_step(x, g)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"swin_tiny @512x512 B=16: {dt * 1e3:.2f} ms/step = {16 / dt:.0f} images/s ({net.flat.n_trainable / 1e6:.1f} M parameters)")
