import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import usseg_oracle as O
from ultrasound_modeling_amd import ops, ResNest as R
from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
orig = ops.conv2d_dgrad
def patched(dy, wp_d, ksize, dilation, dx, residual=None, accumulate=False):
    chk = (not accumulate) and residual is None
    if chk:
        dx.fill_(float("nan"))
    out = orig(dy, wp_d, ksize, dilation, dx, residual, accumulate)
    if chk:
        torch.cuda.synchronize()
        n = torch.isnan(dx.float())
        if n.any():
            per_c = n.sum(dim=(0, 1, 2))
            print(f"dgrad k{ksize} d{dilation} dy{tuple(dy.shape)} -> dx{tuple(dx.shape)} stride {dx.stride()}: {int(n.sum())} unwritten; channels {per_c.nonzero().flatten().tolist()[:12]} pixels-with-nan {int(n.any(-1).sum())}")
    return out
ops.conv2d_dgrad = patched
B, HW = int(sys.argv[1]), int(sys.argv[2])
net = VisionTransformer(batch_size=B, img_size=(HW, HW), in_channels=1, device="cuda:0", seed=0)
x, y = O.synthetic_batch(B, HW, HW, 1, seed=40)
net.train_step(x, y.float())
torch.cuda.synchronize()
print("done")
