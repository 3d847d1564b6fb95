"""CPU tests: the oracle against the committed golden fixtures and against an independent NumPy-loop implementation
of every primitive (direct index arithmetic, no framework convolution), so that the torch-based restatement and its
reading of the Keras padding / layout rules are checked twice."""
import os

import numpy as np
import pytest
import torch

import usseg_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
D = torch.float64


# ------------------------------------------------------------------------------------------------ NumPy loops
def np_conv2d_same(x, w, b, d=1):
    B, H, W, Ci = x.shape
    k, _, _, Co = w.shape
    pad = d * (k - 1) // 2
    y = np.zeros((B, H, W, Co))
    for kh in range(k):
        for kw in range(k):
            for h in range(H):
                ih = h + kh * d - pad
                if ih < 0 or ih >= H:
                    continue
                for ww in range(W):
                    iw = ww + kw * d - pad
                    if 0 <= iw < W:
                        y[:, h, ww, :] += x[:, ih, iw, :] @ w[kh, kw]
    return y + b


def np_tconv_s2_same(x, w, b):
    """out[2i + kh - pad] += x[i] * w[kh]  with pad = 0 (k=3, crop the end) / 1 (k=4)   """
    B, H, W, Ci = x.shape
    k, _, Co, _ = w.shape
    pad = 1 if k == 4 else 0
    y = np.zeros((B, 2 * H, 2 * W, Co))
    for i in range(H):
        for j in range(W):
            for kh in range(k):
                for kw in range(k):
                    oy, ox = 2 * i + kh - pad, 2 * j + kw - pad
                    if 0 <= oy < 2 * H and 0 <= ox < 2 * W:
                        y[:, oy, ox, :] += x[:, i, j, :] @ w[kh, kw].T
    return y + b


@pytest.mark.parametrize("k,d", [(3, 1), (3, 2), (3, 4), (1, 1)])
def test_conv_matches_numpy_loops(k, d):
    g = np.random.default_rng(0)
    x, w, b = g.standard_normal((2, 7, 6, 3)), g.standard_normal((k, k, 3, 4)), g.standard_normal(4)
    got = O.conv2d_same(torch.tensor(x), torch.tensor(w), torch.tensor(b), d).numpy()
    assert np.allclose(got, np_conv2d_same(x, w, b, d), atol=1e-12)


@pytest.mark.parametrize("k", [3, 4])
def test_tconv_matches_numpy_loops(k):
    g = np.random.default_rng(1)
    x, w, b = g.standard_normal((2, 4, 5, 3)), g.standard_normal((k, k, 2, 3)), g.standard_normal(2)
    got = O.conv2d_transpose_s2_same(torch.tensor(x), torch.tensor(w), torch.tensor(b)).numpy()
    assert np.allclose(got, np_tconv_s2_same(x, w, b), atol=1e-12)


def test_norms_pool_softmax_match_numpy():
    g = np.random.default_rng(2)
    x = g.standard_normal((2, 4, 6, 5))
    ga, be = g.standard_normal(5), g.standard_normal(5)
    mu, var = x.mean(-1, keepdims=True), x.var(-1, keepdims=True)
    assert np.allclose(O.layer_norm(torch.tensor(x), torch.tensor(ga), torch.tensor(be)).numpy(), (x - mu) / np.sqrt(var + 1e-3) * ga + be)
    pooled = (x[:, 0::2, 0::2] + x[:, 1::2, 0::2] + x[:, 0::2, 1::2] + x[:, 1::2, 1::2]) / 4
    assert np.allclose(O.avg_pool2(torch.tensor(x)).numpy(), pooled)
    e = np.exp(x - x.max(-1, keepdims=True))
    assert np.allclose(O.softmax_lastaxis(torch.tensor(x)).numpy(), e / e.sum(-1, keepdims=True))
    assert np.allclose(O.leaky_relu(torch.tensor(x)).numpy(), np.where(x >= 0, x, 0.3 * x))
    assert np.allclose(O.elu(torch.tensor(x)).numpy(), np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1))


def test_split_attention_matches_numpy():
    """ResNest.py:171-199 restated with plain array ops (softmax over the CHANNEL axis, same dense2 for each r)."""
    g = np.random.default_rng(3)
    R, C, Hd = 3, 6, 3
    ins = [g.standard_normal((2, 4, 4, C)) for _ in range(R)]
    w1, b1, w2, b2 = g.standard_normal((C, Hd)), g.standard_normal(Hd), g.standard_normal((Hd, C)), g.standard_normal(C)
    ga, be = g.standard_normal(Hd), g.standard_normal(Hd)
    pooled = sum(ins).mean(axis=(1, 2))
    h = pooled @ w1 + b1
    h = (h - h.mean(-1, keepdims=True)) / np.sqrt(h.var(-1, keepdims=True) + 1e-3) * ga + be
    h = np.where(h >= 0, h, 0.3 * h)
    z = h @ w2 + b2
    z = np.exp(z - z.max(-1, keepdims=True))
    z = z / z.sum(-1, keepdims=True)
    want = sum(t * z[:, None, None, :] for t in ins)
    P = {"dense1.kernel": torch.tensor(w1).reshape(1, 1, C, Hd), "dense1.bias": torch.tensor(b1), "dense1_bn.gamma": torch.tensor(ga),
         "dense1_bn.beta": torch.tensor(be), "dense2.kernel": torch.tensor(w2).reshape(1, 1, Hd, C), "dense2.bias": torch.tensor(b2)}
    got = O.split_attention([torch.tensor(t) for t in ins], P, "", R).numpy()
    assert np.allclose(got, want, atol=1e-12)


# ------------------------------------------------------------------------------------------------ golden fixtures
def test_arch_b_golden_fixture_regenerates():
    z = np.load(os.path.join(GOLD, "archB_64x64x1.npz"), allow_pickle=False)
    P = {k: v.float().double() for k, v in O.init_vision_transformer_params(channel=1, seed=3, perturb=True).items()}
    cs = float(sum(v.double().abs().sum() for v in P.values()))
    if abs(cs - float(z["param_checksum"])) > 1e-6 * cs:
        pytest.skip("PyTorch RNG differs from the one that generated the fixture")
    x = torch.tensor(z["x"]).double().to(torch.bfloat16).double()
    x4, feats = O.resnest_forward(x, P, 3, 3, "transformer.embeddings.hybrid_model.")
    assert np.allclose(x4.numpy(), z["x_4"], rtol=1e-5, atol=1e-5)
    assert np.allclose(feats[2].numpy(), z["x_1"], rtol=1e-5, atol=1e-5)
    probs = O.vision_transformer_forward(x, P, 3, 3)
    assert probs.shape == (1, 64, 64, 3) and np.allclose(probs.numpy(), z["probs"], rtol=1e-5, atol=1e-6)
    loss = O.compute_loss(torch.tensor(z["y"]).double(), probs, 1)
    assert abs(loss.item() - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    # as-executed (radix loop repeated) and de-duplicated modes agree
    probs2 = O.vision_transformer_forward(x, P, 3, 3, as_executed=False)
    assert torch.allclose(probs, probs2, atol=1e-12)


def test_layer_golden_fixture_regenerates():
    z = np.load(os.path.join(GOLD, "layers_2x16x16.npz"), allow_pickle=False)
    t = lambda k: torch.tensor(z[k]).double()
    assert np.allclose(O.conv2d_same(t("conv3x3::in0"), t("conv3x3::in1"), t("conv3x3::in2")).numpy(), z["conv3x3::y"], rtol=1e-5, atol=1e-5)
    assert np.allclose(O.conv2d_same(t("conv3x3_d4::in0"), t("conv3x3_d4::in1"), t("conv3x3_d4::in2"), 4).numpy(), z["conv3x3_d4::y"],
                       rtol=1e-5, atol=1e-5)
    assert np.allclose(O.conv2d_transpose_s2_same(t("tconv3::in0"), t("tconv3::in1"), t("tconv3::in2")).numpy(), z["tconv3::y"], rtol=1e-5,
                       atol=1e-5)
    assert np.allclose(O.conv2d_transpose_s2_same(t("tconv4::in0"), t("tconv4::in1"), t("tconv4::in2")).numpy(), z["tconv4::y"], rtol=1e-5,
                       atol=1e-5)


def test_train_step_lowers_loss_and_dedup_equals_as_executed_grads():
    P = {k: v.clone() for k, v in O.init_vision_transformer_params(channel=1, seed=5, perturb=True).items()}
    x, y = O.synthetic_batch(1, 32, 32, 1, seed=4)
    Pa, Pb = dict(P), dict(P)
    la, _, ga, _ = O.train_step(x, y, Pa, {}, 1, as_executed=True)
    lb, _, gb, _ = O.train_step(x, y, Pb, {}, 1, as_executed=False)
    assert abs(la.item() - lb.item()) < 1e-9 * abs(la.item())
    for k in ga:
        assert torch.allclose(ga[k], gb[k], rtol=1e-9, atol=1e-12), k   # gradients flow radix times into the shared weights
    st = {}
    l0 = O.train_step(x, y, P, st, 1)[0].item()
    for _ in range(3):
        l1 = O.train_step(x, y, P, st, 1)[0].item()
    assert l1 < l0
