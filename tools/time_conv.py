"""Times one 3x3 convolution forward / backward-data / weight gradient with HIP events.
usage: python tools/time_conv.py B H W Cin Cout [dilation] [reps]     (planner switches through the USSEG_* environment)"""
import sys

import torch

sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
from ultrasound_modeling_amd.flat import FlatParams
from ultrasound_modeling_amd.layers import Conv2D

B, H, W, Cin, Cout = (int(a) for a in sys.argv[1:6])
d = int(sys.argv[6]) if len(sys.argv) > 6 else 1
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 30
layer = Conv2D(Cin, Cout, 3, dilation_rate=d)
FlatParams(layer, "cuda")
layer.on_finalize("cuda")
x = torch.randn(B, H, W, layer.cin_p).to(torch.bfloat16).cuda()
dy = torch.randn(B, H, W, layer.cout_p).to(torch.bfloat16).cuda()


def t(fn):
    """GPU time per call: ``reps`` calls captured into one HIP graph (no host launch gaps), replayed three times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


fl = 2.0 * B * H * W * Cin * Cout * 9
mb = B * H * W * (layer.cin_p + layer.cout_p) * 2 / 1e6
layer.forward(x)
tf = t(lambda: layer.forward(x))
td = t(lambda: layer.backward(dy, skip_wgrad=True, skip_bias=True))
tw = t(lambda: layer.backward(dy, need_dx=False, skip_bias=True))
print(f"{B}x{H}x{W} {Cin}->{Cout} d{d}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF/s, {mb / tf * 1e-6 * 1e6:.2f} GB/ms)  dgrad {td:.1f} us  wgrad+finish {tw:.1f} us   [{mb:.0f} MB min traffic]")
