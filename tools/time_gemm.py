"""Times one dense layer (a 1x1 convolution on [1, M, 1, K] tokens) forward / backward-data / weight gradient with HIP events.
usage: python tools/time_gemm.py M K N [reps]     (planner switches through the USSEG_* environment)"""
import sys

import torch

sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
from ultrasound_modeling_amd.flat import FlatParams
from ultrasound_modeling_amd.layers import Conv2D

M, K, N = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
layer = Conv2D(K, N, 1, init="glorot")
FlatParams(layer, "cuda")
layer.on_finalize("cuda")
x = torch.randn(1, M, 1, K).to(torch.bfloat16).cuda()
dy = torch.randn(1, M, 1, N).to(torch.bfloat16).cuda()


def t(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 2.0 * M * K * N
y = layer.forward(x)
tf = t(lambda: layer.forward(x))
td = t(lambda: layer.backward(dy, skip_wgrad=True, skip_bias=True))
tw = t(lambda: layer.backward(dy, need_dx=False, skip_bias=True))
print(f"M={M} K={K} N={N}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF/s)  dgrad {td:.1f} us ({fl / td / 1e6:.0f} TF/s)  wgrad+finish {tw:.1f} us ({fl / tw / 1e6:.0f} TF/s)")
