"""GPU parity of the kernel-sharing atrous convolution (Decoder.py:150-346) against the oracle:
the layer as written (cumulative rates 1,3,7,15,31), as intended (1,2,4,8,16), and the KSACBlock in its executable 'sum' reading."""
import pytest
import torch

import usseg_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(t):
    return t.detach().to(torch.bfloat16).to(torch.float64)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def dev(t64):
    B, H, W, C = t64.shape
    cp = (C + 7) // 8 * 8
    out = torch.zeros(B, H, W, cp, dtype=torch.bfloat16)
    out[..., :C] = t64.detach().to(torch.bfloat16)
    return out.to(DEV)


def _load(layer, P, prefix):
    layer.conv.kernel.data.copy_(P[prefix + "kernel"].float())
    for r in layer.dilation_rates_list:
        bn = getattr(layer, f"bn_r_{r}")
        bn.gamma.data.copy_(P[f"{prefix}bn_r_{r}.gamma"].float()); bn.beta.data.copy_(P[f"{prefix}bn_r_{r}.beta"].float())
        bn.moving_mean_p[:bn.C] = P[f"{prefix}bn_r_{r}.moving_mean"].float().to(DEV)
        bn.moving_variance_p[:bn.C] = P[f"{prefix}bn_r_{r}.moving_variance"].float().to(DEV)
    layer.conv.repack()


@pytest.mark.parametrize("as_written,B,H,W,cin,cout", [(True, 2, 32, 32, 24, 16), (False, 2, 32, 32, 24, 16), (True, 1, 64, 32, 40, 24),
                                                       (False, 2, 16, 16, 64, 64)])
def test_kernel_sharing_conv(as_written, B, H, W, cin, cout):
    from ultrasound_modeling_amd import ops
    from ultrasound_modeling_amd.Decoder import KernelSharingConv
    from ultrasound_modeling_amd.flat import FlatParams
    g = torch.Generator().manual_seed(cin + cout)
    P = O.init_ksac_params(cin, cout, seed=3, perturb=True)
    P["kernel"] = bf(P["kernel"])
    layer = KernelSharingConv(cout, [3, 3], in_channels=cin, kernel_initializer="HeNormal", as_written=as_written)
    assert layer.dilations == ((1, 3, 7, 15, 31) if as_written else (1, 2, 4, 8, 16))
    fp = FlatParams(layer, DEV)
    _load(layer, P, "")
    x = bf(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64))
    dys = [bf(torch.randn(B, H, W, cout, generator=g, dtype=torch.float64)) for _ in range(5)]
    names = [k for k in P if not k.endswith("moving_mean") and not k.endswith("moving_variance")]
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    Pl = dict(P); Pl.update(leaves)
    xl = x.clone().requires_grad_(True)
    O.STORAGE_DTYPE = torch.bfloat16          # the layer stores the conv outputs in bf16 before BatchNorm + GELU: storage-emulating oracle
    try:
        refs = O.ksac_layer(xl, Pl, "", as_written=as_written)
        grads = torch.autograd.grad(refs, [xl] + [leaves[k] for k in names], dys)
    finally:
        O.STORAGE_DTYPE = None
    outs = layer(dev(x))
    assert isinstance(outs, list) and len(outs) == 5
    for j, (o, r) in enumerate(zip(outs, refs)):
        assert rel(o[..., :cout], bf(r)) < 2e-3, f"branch {j}"
    fp.zero_grad()
    with ops.overlap_region():
        dx = layer.backward([dev(d) for d in dys])
    torch.cuda.synchronize()
    assert rel(dx[..., :cin], bf(grads[0])) < 4e-3                  # five bf16 accumulation passes over dx
    got = {"kernel": layer.conv.kernel.grad}
    for r in layer.dilation_rates_list:
        got[f"bn_r_{r}.gamma"], got[f"bn_r_{r}.beta"] = getattr(layer, f"bn_r_{r}").gamma.grad, getattr(layer, f"bn_r_{r}").beta.grad
    for k, gref in zip(names, grads[1:]):
        assert rel(got[k], gref) < 4e-3, k


def test_ksac_block_sum_reading_and_the_reference_failure():
    from ultrasound_modeling_amd import ops
    from ultrasound_modeling_amd.Decoder import KSACBlock
    from ultrasound_modeling_amd.flat import FlatParams
    with pytest.raises(ValueError):
        blk0 = KSACBlock(16, in_channels=16, fuse=None)
        FlatParams(blk0, DEV)
        blk0(torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device=DEV))
    g = torch.Generator().manual_seed(5)
    oc, cin = 16, 24
    blk = KSACBlock(oc, in_channels=cin, skip_channels=oc, as_written=False)
    fp = FlatParams(blk, DEV)
    P1, P2 = O.init_ksac_params(2 * oc, oc, seed=1, perturb=True), O.init_ksac_params(oc, oc, seed=2, perturb=True)
    P1["kernel"], P2["kernel"] = bf(P1["kernel"]), bf(P2["kernel"])
    _load(blk.conv1, P1, ""); _load(blk.conv2, P2, "")
    wu = bf(torch.randn(3, 3, oc, cin, generator=g, dtype=torch.float64) * 0.1)
    bu = (torch.randn(oc, generator=g, dtype=torch.float64) * 0.1).float().double()
    blk.up.kernel.data.copy_(wu.float()); blk.up.bias.data.copy_(bu.float()); blk.up.repack()
    x = bf(torch.randn(2, 8, 8, cin, generator=g, dtype=torch.float64))
    skip = bf(torch.randn(2, 16, 16, oc, generator=g, dtype=torch.float64))
    dout = bf(torch.randn(2, 16, 16, oc, generator=g, dtype=torch.float64))
    xl, sl = x.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    O.STORAGE_DTYPE = torch.bfloat16          # the block stores bf16 between its stages: compare with the storage-emulating oracle
    try:
        cat = torch.cat([O.conv2d_transpose_s2_same(xl, wu, bu), sl], dim=3)
        y = O._q(sum(O.ksac_layer(cat, P1, "", as_written=False)))
        ref = O._q(sum(O.ksac_layer(y, P2, "", as_written=False)))
        gx, gs = torch.autograd.grad(ref, [xl, sl], dout)
    finally:
        O.STORAGE_DTYPE = None
    out = blk(dev(x), dev(skip))
    assert rel(out, ref) < 1e-2
    fp.zero_grad()
    with ops.overlap_region():
        dx, dskip = blk.backward(dev(dout))
    torch.cuda.synchronize()
    assert rel(dx[..., :cin], gx) < 3e-2 and rel(dskip, gs) < 3e-2
