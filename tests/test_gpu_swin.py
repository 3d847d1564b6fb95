"""GPU parity of the windowed-attention encoder (SwinTransformer.py, BASELINE configs[4]) against
oracle/swin_oracle.py: the new kernels one by one on bf16-representable inputs (1e-3 bars as in test_gpu_ops.py), a small model end
to end (bf16-depth bars as in test_gpu_model.py), and a property run at the configuration's own 512x512, batch 16."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import swin_oracle as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(t):
    return t.detach().to(torch.bfloat16).to(torch.float64)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def dev(t):
    return t.detach().to(torch.bfloat16).contiguous().to(DEV)


@pytest.mark.parametrize("B,H,W,C,heads,ws,shift", [(2, 8, 8, 64, 2, 4, 0), (2, 8, 16, 64, 2, 4, 2), (1, 16, 16, 96, 3, 8, 4), (2, 16, 16, 32, 2, 8, 0), (2, 32, 24, 64, 2, 8, 3), (1, 24, 16, 128, 4, 8, 0),
                                                     (8, 128, 128, 32, 1, 8, 3), (4, 128, 128, 64, 2, 8, 0),   # >= 1024 blocks of 2 windows: the multi-window backward
                                                     (3, 4, 4, 32, 4, 2, 1), (1, 8, 8, 128, 2, 4, 2)])
def test_window_attention_kernel(B, H, W, C, heads, ws, shift):
    from ultrasound_modeling_amd import ops
    g = torch.Generator().manual_seed(B * 100 + ws * 10 + shift)
    d = C // heads
    qkv = bf(torch.randn(B, H, W, 3 * C, generator=g, dtype=torch.float64))
    table = (0.5 * torch.randn((2 * ws - 1) ** 2, heads, generator=g, dtype=torch.float64)).float().double()
    dout = bf(torch.randn(B, H, W, C, generator=g, dtype=torch.float64))

    def ref(qkv_, table_):
        x = qkv_
        if shift:
            x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        xw = S.window_partition(x, ws).reshape(-1, ws * ws, 3 * C)
        Bn, N, _ = xw.shape
        q3 = xw.reshape(Bn, N, 3, heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = q3[0] * d ** -0.5, q3[1], q3[2]
        a = q @ k.transpose(-1, -2) + table_[S.relative_position_index(ws).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)[None]
        if shift:
            m = S.shift_mask(H, W, ws, shift)
            a = (a.reshape(-1, m.shape[0], heads, N, N) + m[None, :, None]).reshape(-1, heads, N, N)
        o = (torch.softmax(a, -1) @ v).transpose(1, 2).reshape(Bn, N, C)
        o = S.window_reverse(o.reshape(-1, ws, ws, C), ws, H, W)
        return torch.roll(o, shifts=(shift, shift), dims=(1, 2)) if shift else o
    ql, tl = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    want = ref(ql, tl)
    gq, gt = torch.autograd.grad(want, [ql, tl], dout)
    qd, td = dev(qkv), table.float().to(DEV)
    out = ops.window_attn_fwd(qd, td, heads, ws, shift, torch.empty(B, H, W, C, dtype=torch.bfloat16, device=DEV))
    assert rel(out, bf(want)) < 1e-3
    dtab = torch.zeros_like(td)
    dqkv = ops.window_attn_bwd(qd, dev(dout), td, heads, ws, shift, torch.empty_like(qd), dtab)
    torch.cuda.synchronize()
    assert rel(dqkv, bf(gq)) < 1e-3
    assert rel(dtab, gt) < 1e-3


@pytest.mark.parametrize("M,C", [(37, 96), (64, 512), (50, 768), (33, 1536), (17, 3072), (9, 4096)])
def test_wide_layer_norm(M, C):
    from ultrasound_modeling_amd import ops
    g = torch.Generator().manual_seed(C)
    x = bf(torch.randn(1, 1, M, C, generator=g, dtype=torch.float64) * 1.3 + 0.2)
    dy = bf(torch.randn(1, 1, M, C, generator=g, dtype=torch.float64))
    gam, bet = (1 + 0.2 * torch.randn(C, generator=g, dtype=torch.float64)).float().double(), (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).float().double()
    xl, gl, bl = (t.clone().requires_grad_(True) for t in (x, gam, bet))
    want = S.layer_norm(xl, gl, bl)
    gx, gg, gb = torch.autograd.grad(want, [xl, gl, bl], dy)
    xd = dev(x)
    out = ops.ln_wide_fwd(xd, gam.float().to(DEV), bet.float().to(DEV), 1e-5, torch.empty_like(xd))
    assert rel(out, bf(want)) < 1e-3
    dga, dbe = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.ln_wide_bwd(xd, dev(dy), gam.float().to(DEV), 1e-5, torch.empty_like(xd), dga, dbe)
    torch.cuda.synchronize()
    assert rel(dx, bf(gx)) < 1e-3 and rel(dga, gg) < 1e-3 and rel(dbe, gb) < 1e-3


def test_patchify_merge_and_token_mean():
    from ultrasound_modeling_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 24, 3, generator=g, dtype=torch.float64)
    p = ops.patchify(x.to(DEV), 4)                                     # [2,4,6,48]
    want = x.reshape(2, 4, 4, 6, 4, 3).permute(0, 1, 3, 2, 4, 5).reshape(2, 4, 6, 48)      # (ph, pw, c) channel order
    assert torch.equal(p.cpu(), want.float().to(torch.bfloat16))
    t = bf(torch.randn(2, 8, 6, 16, generator=g, dtype=torch.float64))
    m = torch.empty(2, 4, 3, 64, dtype=torch.bfloat16, device=DEV)
    ops.patch_merge(dev(t), m)
    want = torch.cat([t[:, 0::2, 0::2], t[:, 1::2, 0::2], t[:, 0::2, 1::2], t[:, 1::2, 1::2]], -1)
    assert torch.equal(m.cpu().double(), want)
    back = torch.zeros(2, 8, 6, 16, dtype=torch.bfloat16, device=DEV)
    ops.patch_merge(back, m, backward=True)
    assert torch.equal(back.cpu().double(), t)
    td = dev(t)
    mean = ops.token_mean_fwd(td)
    assert rel(mean, t.mean(dim=(1, 2))) < 1e-6
    dy = torch.randn(2, 16, generator=g).to(DEV)
    dx = ops.token_mean_bwd(dy, td)
    assert rel(dx, bf((dy.double().cpu() / 48)[:, None, None, :].expand(2, 8, 6, 16))) < 1e-6


def _name_map(net):
    out = {}
    for k, p in net.named_parameters():
        ok = k.replace("basic_layers.", "layers").replace("blocks.", "blocks")
        parts = ok.split(".")
        # "layers0.blocks1.attn.qkv.kernel" -> "layers0/blocks1/attn/qkv/kernel"
        out[k] = "/".join(parts)
    return out


def _load(net, P):
    for k, ok in _name_map(net).items():
        t = dict(net.named_parameters())[k]
        t.data.copy_(P[ok].float().reshape(t.shape))
    net.repack()


@pytest.mark.parametrize("ape", [False, True])
def test_small_swin_model_against_the_oracle(ape):
    from ultrasound_modeling_amd.SwinTransformer import SwinTransformerModel
    cfg = dict(patch_size=4, embed_dim=32, depths=[2, 2], num_heads=[2, 4], window_size=4, ape=ape, img_size=(64, 64))
    P = {k: v.float().double() for k, v in S.init_swin_params(cfg, in_chans=1, seed=5).items()}
    net = SwinTransformerModel(model_name="tiny_test", img_size=(64, 64), patch_size=(4, 4), in_chans=1, embed_dim=32, depths=[2, 2], num_heads=[2, 4],
                               window_size=4, ape=ape)
    assert set(_name_map(net).values()) == set(P)
    _load(net, P)
    g = torch.Generator().manual_seed(6)
    x = bf(torch.randn(2, 64, 64, 1, generator=g, dtype=torch.float64))
    d_out = torch.randn(2, 64, generator=g, dtype=torch.float64).float().double()
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    want, feats_r = S.swin_forward(x, leaves, cfg)
    grads = dict(zip(leaves, torch.autograd.grad(want, list(leaves.values()), d_out)))
    out, feats = net(x.to(DEV))
    assert tuple(out.shape) == (2, 64) and len(feats) == 1 and tuple(feats[0].shape) == (2, 256, 32)
    e_o, e_f = rel(out, want), rel(feats[0], feats_r[0])
    net.flat.zero_grad()
    net.backward(d_out.float().to(DEV))
    torch.cuda.synchronize()
    own = dict(net.named_parameters())
    errs = sorted((rel(own[k].grad.reshape(grads[ok].shape), grads[ok]), k) for k, ok in _name_map(net).items())
    print(f"swin small: out rel {e_o:.3e} stage-0 feature rel {e_f:.3e} grad median {errs[len(errs) // 2][0]:.3e} worst {errs[-1]}")
    assert e_o < 2e-2 and e_f < 2e-2
    assert errs[len(errs) // 2][0] < 3e-2 and errs[-1][0] < 1.5e-1


def test_swin_cfg5_512x512_batch16_properties():
    """BASELINE configs[4]: 512x512, batch 16 per GPU - swin_tiny widths (embed 96, depths 2/2/6/2, heads 3/6/12/24) with 8x8
    windows on the 128x128 token grid: forward + backward finite, bit-reproducible, and the include_top head runs."""
    from ultrasound_modeling_amd.SwinTransformer import SwinTransformerModel
    kw = dict(img_size=(512, 512), patch_size=(4, 4), in_chans=1, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=8)
    x = torch.randn(16, 512, 512, 1, generator=torch.Generator().manual_seed(1))
    res = []
    for _ in range(2):
        net = SwinTransformerModel(model_name="swin_tiny_512", seed=0, **kw)
        out, feats = net(x)
        assert tuple(out.shape) == (16, 768) and [tuple(f.shape) for f in feats] == [(16, 16384, 96), (16, 4096, 192), (16, 1024, 384)]
        net.flat.zero_grad()
        net.backward(torch.ones(16, 768, device=DEV) / 768)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and torch.isfinite(net.flat.grad).all() and net.flat.grad.abs().max().item() > 0
        res.append((out.clone(), net.flat.grad.clone()))
        del net
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    top = SwinTransformerModel(model_name="t", include_top=True, num_classes=10, img_size=(64, 64), in_chans=1, embed_dim=32, depths=[2, 2],
                               num_heads=[2, 4], window_size=4)
    logits, _ = top(torch.randn(2, 64, 64, 1))
    assert tuple(logits.shape) == (2, 10)
    top.flat.zero_grad()
    top.backward(torch.ones(2, 10))
    assert torch.isfinite(top.flat.grad).all() and top.head.kernel.grad.abs().max().item() > 0


def test_swin_train_step_graph_replay_equals_eager():
    """The shared step driver (step.TrainStepDriver) on the Swin encoder: three optimisation steps driven by upstream gradients, eager vs
    HIP-graph replay (what ``bench.py --arch S`` times) - outputs, parameters and Adam state bit for bit; capture consumes no step."""
    from ultrasound_modeling_amd.SwinTransformer import SwinTransformerModel
    kw = dict(model_name="t", img_size=(64, 64), patch_size=(4, 4), in_chans=1, embed_dim=96, depths=[2, 2], num_heads=[3, 6], window_size=8, seed=3)
    eager, graph = SwinTransformerModel(**kw), SwinTransformerModel(**kw)
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(2, 64, 64, 1, generator=g) for _ in range(3)]
    ds = [torch.randn(2, 192, generator=g) for _ in range(3)]

    def state(n):
        torch.cuda.synchronize()
        return n.flat.flat.clone(), n.optimizer.m.clone(), n.optimizer.v.clone(), int(n.optimizer.step_dev.item())
    graph.capture_graph(xs[0], ds[0])
    for a, b in zip(state(eager), state(graph)):
        assert (a == b) if isinstance(a, int) else torch.equal(a, b), "capture must not change the training state"
    for i in range(3):
        l0, o0 = eager.train_step(xs[i], ds[i])
        l1, o1 = graph.train_step(xs[i], ds[i])
        torch.cuda.synchronize()
        assert torch.equal(o0, o1) and l0.item() == l1.item(), i
        for a, b in zip(state(eager), state(graph)):
            assert (a == b) if isinstance(a, int) else torch.equal(a, b), f"step {i}"
    assert state(graph)[3] == 3 and not torch.equal(state(graph)[0], SwinTransformerModel(**kw).flat.flat)
