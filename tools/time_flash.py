"""Times the fused attention kernels at one shape (default BASELINE configs[3]: 8 images, 1024 tokens, 4 heads of 128) with HIP events.
usage: python tools/time_flash.py [B N heads]   (USSEG_FLASH_TILE=0|1|2 selects the tiling)"""
import math
import sys

import torch

sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops

B, N, nh = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1024, 4)
hs = 128 * nh
qkv = (torch.randn(B, N, 1, 3 * hs) * 0.8).to(torch.bfloat16).cuda()
d_out = torch.randn(B, N, 1, hs).to(torch.bfloat16).cuda()
out, dqkv = torch.empty_like(d_out), torch.empty_like(qkv)
o32 = torch.empty(B, N, hs, device="cuda")
lse, delta = torch.empty(B * nh, N, device="cuda"), torch.empty(B * nh, N, device="cuda")
scale = 1 / math.sqrt(nh)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 4 * B * nh * N * N * 128
tf = t(lambda: ops.flash_attn_fwd(qkv, nh, scale, out, lse, o32))
tb = t(lambda: ops.flash_attn_bwd(qkv, nh, scale, out, d_out, lse, delta, dqkv, o32))
print(f"B={B} N={N} heads={nh}: forward {tf:.1f} us ({fl / tf / 1e6:.0f} TF/s nominal), backward (dQ + dK/dV) {tb:.1f} us ({2.5 * fl / tb / 1e6:.0f} TF/s nominal)")
