"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot run here: no TensorFlow).

    python tests/golden/make_golden.py

archB_64x64x1.npz  - BASELINE config 1 ("single 64x64 1-ch tile, UNet forward ... plumbing"): Arch B (ResNest.py r=3,k=3
                     + patch embedding + DecoderCup, no ViT), B=1, perturbed-initialisation weights (seed 3): input,
                     encoder outputs, probabilities, loss, gradient norms of every variable and a few full gradients.
layers_2x16x16.npz - one tile per layer type (3x3 conv, dilated conv, 1x1, tconv k=3 / k=4, LN+LeakyReLU, BN-inference,
                     avg-pool, residual_S stage): inputs, weights, forward outputs, input- and weight-gradients.
The weights of the first file are re-created from the seed by usseg_oracle.init_*; `param_checksum` guards against a
PyTorch RNG change.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import usseg_oracle as O  # noqa: E402


def checksum(P):
    return float(sum(v.double().abs().sum() for v in P.values()))


def arch_b():
    P = {k: v.float().double() for k, v in O.init_vision_transformer_params(channel=1, seed=3, perturb=True).items()}
    x, y = O.synthetic_batch(1, 64, 64, 1, seed=21)
    xb = x.to(torch.bfloat16).double()          # the product casts its input to bf16 (ResNest.py:39 boundary)
    names = O.trainable_names(P)
    leaves = [P[n].clone().requires_grad_(True) for n in names]
    Pl = dict(P)
    Pl.update(zip(names, leaves))
    x4, feats = O.resnest_forward(xb, Pl, 3, 3, "transformer.embeddings.hybrid_model.")
    probs = O.vision_transformer_forward(xb, Pl, 3, 3)
    loss = O.compute_loss(y, probs, 1)
    grads = torch.autograd.grad(loss, leaves)
    out = dict(x=x.numpy(), y=y.numpy(), x_4=x4.detach().numpy(), x_3=feats[0].detach().numpy(), x_2=feats[1].detach().numpy(),
               x_1=feats[2].detach().numpy(), probs=probs.detach().numpy(), loss=np.float64(loss.item()),
               param_checksum=np.float64(checksum(P)), grad_names=np.array(names),
               grad_norms=np.array([g.norm().item() for g in grads]))
    for n in ("decoder.head.kernel", "transformer.embeddings.hybrid_model.conv1.kernel",
              "transformer.embeddings.hybrid_model.conv_2.cardinal_blocks.1.conv2.kernel", "decoder.blocks.2.bn2_3.gamma"):
        out["grad::" + n] = grads[names.index(n)].numpy()
    np.savez_compressed(os.path.join(HERE, "archB_64x64x1.npz"), **{k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.ndim > 0 else v) for k, v in out.items()})


def layers():
    g = torch.Generator().manual_seed(77)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * sc).to(torch.bfloat16).double()
    out = {}

    def record(name, fn, inputs):
        leaves = [t.clone().requires_grad_(True) for t in inputs]
        y = fn(*leaves)
        dy = r(*y.shape)
        grads = torch.autograd.grad((y * dy).sum(), leaves)
        out[name + "::y"], out[name + "::dy"] = y.detach().numpy(), dy.numpy()
        for i, (t, gr) in enumerate(zip(inputs, grads)):
            out[f"{name}::in{i}"], out[f"{name}::grad{i}"] = t.numpy(), gr.numpy()
    record("conv3x3", lambda x, w, b: O.conv2d_same(x, w, b), [r(2, 16, 16, 16), r(3, 3, 16, 32, sc=0.1), r(32, sc=0.3)])
    record("conv3x3_d4", lambda x, w, b: O.conv2d_same(x, w, b, 4), [r(2, 16, 16, 64, sc=0.5), r(3, 3, 64, 16, sc=0.05), r(16, sc=0.3)])
    record("conv1x1", lambda x, w, b: O.conv2d_same(x, w, b), [r(2, 16, 16, 32), r(1, 1, 32, 64, sc=0.2), r(64, sc=0.3)])
    record("tconv3", lambda x, w, b: O.conv2d_transpose_s2_same(x, w, b), [r(2, 8, 8, 24), r(3, 3, 16, 24, sc=0.1), r(16, sc=0.3)])
    record("tconv4", lambda x, w, b: O.conv2d_transpose_s2_same(x, w, b), [r(2, 8, 8, 24), r(4, 4, 8, 24, sc=0.1), r(8, sc=0.3)])
    record("ln_lrelu", lambda x, ga, be: O.leaky_relu(O.layer_norm(x, ga, be)), [r(2, 16, 16, 24), 1 + r(24, sc=0.2), r(24, sc=0.2)])
    mm, mv = r(32, sc=0.1), 1 + r(32, sc=0.1).abs()
    out["bn_lrelu::mean"], out["bn_lrelu::var"] = mm.numpy(), mv.numpy()
    record("bn_lrelu", lambda x, ga, be: O.leaky_relu(O.batch_norm(x, ga, be, mm, mv, False)), [r(2, 16, 16, 32), 1 + r(32, sc=0.2), r(32, sc=0.2)])
    record("avgpool", lambda x: O.avg_pool2(x), [r(2, 16, 16, 16)])
    np.savez_compressed(os.path.join(HERE, "layers_2x16x16.npz"), **{k: v.astype(np.float32) for k, v in out.items()})


if __name__ == "__main__":
    arch_b()
    layers()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
