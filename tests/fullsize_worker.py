"""Worker of tests/test_gpu_fullsize.py: ONE process = one kernel-planner configuration (the USSEG_* switches are read once
per process).  Builds the model of a BASELINE config at FULL size with seed 0, runs forward + loss + backward on the seeded
synthetic batch and writes loss / probabilities / the flat gradient to a file.

usage: python tests/fullsize_worker.py <archB|archA> <out.pt>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import torch


def build(arch, lr=2e-5):
    """``lr``: small, so that ONE Adam step (whose first update is ~lr*sign(g) for every weight) is a descent step on the batch."""
    import usseg_oracle as O
    if arch == "archB":      # BASELINE configs[1]: ResNeSt-50-style encoder + Decoder.py, 256x256, B=16
        from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
        net = VisionTransformer(batch_size=16, img_size=(256, 256), in_channels=1, seed=0, learning_rate=lr)
        x, y = O.synthetic_batch(16, 256, 256, 1, seed=21, dtype=torch.float32)
    else:                    # BASELINE configs[2] per GPU: TBI_ResNest.py model, 256x256, B=32
        from ultrasound_modeling_amd.TBI_ResNest import ResNest
        net = ResNest(256, 256, 1, 3, ksize=3, radix=3, kpaths=4, learning_rate=lr, seed=0)
        x, y = O.synthetic_batch(32, 256, 256, 1, seed=22, dtype=torch.float32)
    return net, x, y.float()


def grad_pass(net, x, y):
    """zero grads -> forward -> loss -> backward (no update); -> (loss scalar, probs)."""
    x, y = net._prep_x(x), net._prep_y(y)
    probs = net._grad_body(x, y)
    torch.cuda.synchronize()
    loss = net._loss[0].item() if hasattr(net, "_loss") else net._loss_map.sum().item()
    return loss, probs


if __name__ == "__main__":
    arch, out = sys.argv[1], sys.argv[2]
    net, x, y = build(arch)
    loss, probs = grad_pass(net, x, y)
    torch.save({"loss": loss, "probs": probs[:, ::4, ::4].float().cpu(), "grad": net.flat.grad.float().cpu()}, out)
    print(f"[worker {arch}] loss {loss:.4f}")
