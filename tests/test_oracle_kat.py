"""Known-answer tests that pin the CPU oracle (SURVEY.md §8c items i-vii) + finite-difference gradient checks.

The reference ships no tests or fixtures ("parity unpinned"), so these analytic cases are what anchors the
oracle's reading of the Keras semantics.  CPU only.
"""
import math

import numpy as np
import pytest
import torch

import usseg_oracle as O

D = torch.float64


def test_tconv3_same_delta_hits_2i_to_2i_plus_2_and_crops_end():
    """(i) k=3 s2 'same': out[2i+k] += x[i] w[k]; the last row/col (index 2H) is cropped (A.2)."""
    H = W = 4
    w = torch.arange(1, 10, dtype=D).reshape(3, 3, 1, 1)
    for (i, j) in [(1, 2), (3, 3), (0, 0)]:
        x = torch.zeros(1, H, W, 1, dtype=D)
        x[0, i, j, 0] = 1.0
        y = O.conv2d_transpose_s2_same(x, w)[0, :, :, 0]
        assert y.shape == (2 * H, 2 * W)
        want = torch.zeros(2 * H + 1, 2 * W + 1, dtype=D)
        want[2 * i:2 * i + 3, 2 * j:2 * j + 3] = w[:, :, 0, 0]
        assert torch.equal(y, want[:2 * H, :2 * W])


def test_tconv4_same_is_pad_1():
    """(i) k=4 s2 'same' == out[2i+k-1] += x[i] w[k]."""
    H = W = 3
    w = torch.arange(1, 17, dtype=D).reshape(4, 4, 1, 1)
    x = torch.zeros(1, H, W, 1, dtype=D)
    x[0, 1, 2, 0] = 1.0
    y = O.conv2d_transpose_s2_same(x, w)[0, :, :, 0]
    want = torch.zeros(2 * H + 2, 2 * W + 2, dtype=D)
    want[2 * 1:2 * 1 + 4, 2 * 2:2 * 2 + 4] = w[:, :, 0, 0]
    assert torch.equal(y, want[1:2 * H + 1, 1:2 * W + 1])


def test_tconv_kernel_layout_is_kh_kw_out_in():
    x = torch.zeros(1, 2, 2, 2, dtype=D)
    x[0, 0, 0, 1] = 1.0                       # input channel 1
    w = torch.zeros(3, 3, 3, 2, dtype=D)      # [kh,kw,Cout=3,Cin=2]
    w[1, 1, 2, 1] = 5.0                       # in 1 -> out 2 at the centre tap
    y = O.conv2d_transpose_s2_same(x, w)
    assert y[0, 1, 1, 2] == 5.0 and y.abs().sum() == 5.0


def test_cardinal_equals_radix_y_softmax_and_dedup_mode():
    """(ii) cardinal(x) == radix * y * softmax_c(dense2(...)) with the branch computed once (App. C.1)."""
    P = O.init_resnest_params(4, 3, 3, seed=1, perturb=True)
    x = torch.randn(2, 8, 8, 32, dtype=D, generator=torch.Generator().manual_seed(0))
    pre = "conv_1.cardinal_blocks.0."
    a = O.cardinal(x, P, pre, 3, as_executed=True)
    b = O.cardinal(x, P, pre, 3, as_executed=False)
    assert torch.allclose(a, b, rtol=0, atol=1e-12)
    y = O.leaky_relu(O.layer_norm(O.conv2d_same(x, P[pre + "conv1.kernel"], P[pre + "conv1.bias"]), P[pre + "conv1_bn.gamma"], P[pre + "conv1_bn.beta"]))
    y = O.leaky_relu(O.layer_norm(O.conv2d_same(y, P[pre + "conv2.kernel"], P[pre + "conv2.bias"]), P[pre + "conv2_bn.gamma"], P[pre + "conv2_bn.beta"]))
    g = (3 * y).mean(dim=(1, 2))[:, None, None, :]
    h = O.leaky_relu(O.layer_norm(O.conv2d_same(g, P[pre + "split.dense1.kernel"], P[pre + "split.dense1.bias"]),
                                  P[pre + "split.dense1_bn.gamma"], P[pre + "split.dense1_bn.beta"]))
    z = torch.softmax(O.conv2d_same(h, P[pre + "split.dense2.kernel"], P[pre + "split.dense2.bias"]), dim=-1)
    assert torch.allclose(a, 3 * y * z, rtol=1e-12, atol=1e-12)
    # softmax is over CHANNELS (App. C.2): the attention vector sums to 1 over the channel axis
    assert torch.allclose(z.sum(-1), torch.ones(2, 1, 1, dtype=D))


def test_bn_inference_fresh_stats():
    """(iii) BN inference with fresh moving stats == x / sqrt(1 + 1e-3)."""
    x = torch.randn(2, 3, 3, 5, dtype=D)
    y = O.batch_norm(x, torch.ones(5, dtype=D), torch.zeros(5, dtype=D), torch.zeros(5, dtype=D), torch.ones(5, dtype=D))
    assert torch.allclose(y, x / math.sqrt(1 + 1e-3), rtol=1e-14)


def test_bn_training_mode_stats_and_momentum():
    x = torch.randn(4, 3, 3, 2, dtype=D)
    y, (mm, mv) = O.batch_norm(x, torch.ones(2, dtype=D), torch.zeros(2, dtype=D), torch.zeros(2, dtype=D), torch.ones(2, dtype=D), training=True)
    mu, var = x.mean(dim=(0, 1, 2)), x.var(dim=(0, 1, 2), unbiased=False)
    assert torch.allclose(y, (x - mu) / torch.sqrt(var + 1e-3))
    assert torch.allclose(mm, 0.01 * mu) and torch.allclose(mv, 0.99 + 0.01 * var)


def test_layernorm_known_answers():
    """(iv) LN of a per-pixel constant vector returns beta; of [a,-a] returns +-a/sqrt(a^2+1e-3)."""
    beta = torch.tensor([0.5, -1.0, 2.0], dtype=D)
    y = O.layer_norm(torch.full((1, 2, 2, 3), 7.0, dtype=D), torch.ones(3, dtype=D) * 3, beta)
    assert torch.allclose(y, beta.expand(1, 2, 2, 3))
    a = 0.37
    y = O.layer_norm(torch.tensor([[[[a, -a]]]], dtype=D), torch.ones(2, dtype=D), torch.zeros(2, dtype=D))
    assert torch.allclose(y, torch.tensor([[[[a, -a]]]], dtype=D) / math.sqrt(a * a + 1e-3))


def test_decoder_cup_reshapes_at_256x80():
    """(v) at 256x80 the hidden state reshapes to [B,16,5,512] and x0 to [B,32,10,128],[B,64,20,32],[B,128,40,8]."""
    B, gh, gw = 1, 16, 5
    hidden = torch.arange(B * gh * gw * 512, dtype=D).reshape(B, gh * gw, 512)
    assert hidden.reshape(B, gh, gw, -1).shape == (1, 16, 5, 512)
    for i, want in enumerate([(1, 32, 10, 128), (1, 64, 20, 32), (1, 128, 40, 8)]):
        x0 = hidden.reshape(B, gh * 2 ** (i + 1), gw * 2 ** (i + 1), -1)
        assert x0.shape == want
        # raw row-major reinterpretation, NOT depth-to-space: consecutive memory stays consecutive
        assert torch.equal(x0.reshape(-1), hidden.reshape(-1))
    # a full decoder pass at the native size produces [B,256,80,3] probabilities
    P = O.init_decoder_params(3, 512, seed=0, dtype=D)
    feats = [torch.zeros(B, 32, 10, 256, dtype=D), torch.zeros(B, 64, 20, 128, dtype=D), torch.zeros(B, 128, 40, 64, dtype=D)]
    probs = O.decoder_cup(hidden * 1e-6, feats, P, (gh, gw))
    assert probs.shape == (1, 256, 80, 3) and torch.allclose(probs.sum(-1), torch.ones(1, 256, 80, dtype=D))


def test_cce_label_smoothing_known_answer():
    """(vi) CCE(label_smoothing=0.1) on one-hot / 3 classes: y' = [0.9333, 0.0333, 0.0333]."""
    y = torch.tensor([[[[1.0, 0.0, 0.0]]]], dtype=D)
    p = torch.tensor([[[[0.7, 0.2, 0.1]]]], dtype=D)
    want = -((0.9 + 0.1 / 3) * math.log(0.7) + (0.1 / 3) * math.log(0.2) + (0.1 / 3) * math.log(0.1))
    assert abs(O.cce_label_smoothing(y, p).item() - want) < 1e-14
    # compute_average_loss divides the SUM over pixels by the global batch (VisionTransformer.py:227)
    assert abs(O.compute_loss(y.expand(2, 4, 4, 3), p.expand(2, 4, 4, 3), 8).item() - want * 32 / 8) < 1e-12
    # clip at 1e-7
    p0 = torch.tensor([[[[1.0, 0.0, 0.0]]]], dtype=D)
    l0 = O.cce_label_smoothing(y, p0).item()
    assert abs(l0 - (-(0.9 + 0.1 / 3) * math.log(1 - 1e-7) - 2 * (0.1 / 3) * math.log(1e-7))) < 1e-12


@pytest.mark.parametrize("radix,kpaths,want", [
    (3, 3, [(3, 10, 5), (7, 21, 10), (14, 42, 21), (28, 85, 42)]),
    (3, 4, [(2, 8, 4), (5, 16, 8), (10, 32, 16), (21, 64, 32)]),
    (4, 4, [(2, 8, 4), (4, 16, 8), (8, 32, 16), (16, 64, 32)]),
])
def test_channel_arithmetic_table(radix, kpaths, want):
    """(vii) cv11 / cvkk / hidden for (radix,kpaths) in (3,3),(3,4),(4,4) (ResNest.py:73,120-121,160)."""
    got = [O.cardinal_channels(oc, radix, kpaths) for oc in (64, 128, 256, 512)]
    assert got == want


def test_param_counts_match_survey_b4():
    PB = O.init_vision_transformer_params(channel=1)
    nB = sum(v.numel() for k, v in PB.items() if k in O.trainable_names(PB))
    assert nB == 6270904                       # "6.27 M = 25.1 MB" 
    PA = O.init_archA_params()
    nA = sum(v.numel() for k, v in PA.items() if k in O.trainable_names(PA))
    assert abs(nA - 25.7e6) < 0.1e6            # "25.7 M = 103 MB"


def test_clip_and_adam_known_answers():
    g = [torch.tensor([3.0, 0.0], dtype=D), torch.tensor([4.0], dtype=D)]      # norm 5
    c, n = O.clip_by_global_norm(g, 1.0)
    assert abs(n.item() - 5) < 1e-14 and torch.allclose(c[0], g[0] / 5) and torch.allclose(c[1], g[1] / 5)
    c, _ = O.clip_by_global_norm([t * 0.1 for t in g], 1.0)                       # norm 0.5 < 1: unchanged
    assert torch.allclose(c[0], g[0] * 0.1)
    p, m, v = [torch.tensor([1.0], dtype=D)], [torch.zeros(1, dtype=D)], [torch.zeros(1, dtype=D)]
    O.adam_step(p, [torch.tensor([0.5], dtype=D)], m, v, 1, 1e-3)
    # first Adam step moves by ~lr in the gradient direction (Keras epsilon 1e-7 outside the sqrt)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert abs(p[0].item() - (1.0 - lr_t * 0.05 / (math.sqrt(0.001 * 0.25) + 1e-7))) < 1e-15


def test_my_loss_cat_matches_formula():
    B, H, W = 3, 4, 5
    gen = torch.Generator().manual_seed(0)
    y = torch.softmax(torch.randn(B, H, W, 3, dtype=D, generator=gen), -1)
    p = torch.softmax(torch.randn(B, H, W, 3, dtype=D, generator=gen), -1)
    out = O.my_loss_cat(y, p, H, W)
    assert out.shape == (H, W)
    want = torch.zeros(H, W, dtype=D)
    for c in range(3):
        want -= (y[..., c] * torch.log(p[..., c] + 1e-7)).sum(0) / (y[..., c].sum(0) + 1) / (H * W)
    assert torch.allclose(out, want, rtol=1e-13)


def test_label2vec_rule():
    lab = torch.tensor([0.0, 0.95, 0.96, 1.0, 1.05, 1.5, 2.0, 2.6], dtype=D).reshape(1, 1, 8)
    v = O.label2vec(lab)[0, 0]
    assert torch.allclose(v[:, 0], torch.tensor([1, 1, 0, 0, 0, 0, 0, 0], dtype=D))
    assert torch.allclose(v[:, 2], torch.tensor([0, 0, 0, 0, 0.05, 0.5, 1.0, 1.0], dtype=D), atol=1e-12)
    assert torch.allclose(v[:, 1], torch.tensor([0, 0, 1, 1, 0.95, 0.5, 0.0, 0.0], dtype=D), atol=1e-12)


# ------------------------------------------------------------------------------------------------ finite differences
def _fd_check(fn, inputs, eps=1e-6, n_probe=6, seed=0):
    leaves = [t.clone().requires_grad_(True) for t in inputs]
    out = fn(*leaves)
    gen = torch.Generator().manual_seed(seed)
    wgt = torch.randn(out.shape, dtype=D, generator=gen)
    (out * wgt).sum().backward()
    for li, leaf in enumerate(leaves):
        flat = leaf.detach().reshape(-1)
        idx = torch.randint(0, flat.numel(), (n_probe,), generator=gen)
        for i in idx.tolist():
            def f(delta):
                args = [t.clone() for t in inputs]
                a = args[li].reshape(-1)
                a[i] += delta
                return (fn(*args) * wgt).sum().item()
            num = (f(eps) - f(-eps)) / (2 * eps)
            ana = leaf.grad.reshape(-1)[i].item()
            assert abs(num - ana) <= 1e-5 * max(1.0, abs(num), abs(ana)), (li, i, num, ana)


def test_fd_conv_tconv_norms():
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, dtype=D, generator=g)
    _fd_check(lambda x, w, b: O.conv2d_same(x, w, b, 2), [r(1, 6, 6, 3), r(3, 3, 3, 4), r(4)])
    _fd_check(lambda x, w, b: O.conv2d_transpose_s2_same(x, w, b), [r(1, 3, 4, 3), r(3, 3, 2, 3), r(2)])
    _fd_check(lambda x, w, b: O.conv2d_transpose_s2_same(x, w, b), [r(1, 3, 2, 3), r(4, 4, 2, 3), r(2)])
    _fd_check(lambda x, ga, be: O.leaky_relu(O.layer_norm(x, ga, be)), [r(2, 3, 3, 5), r(5), r(5)])
    _fd_check(lambda x, ga, be: O.elu(O.batch_norm(x, ga, be, torch.zeros(5, dtype=D), torch.ones(5, dtype=D) * 1.3)), [r(2, 3, 3, 5), r(5), r(5)])


def test_fd_residual_S_stage():
    P = O.init_resnest_params(2, 3, 3, seed=2, perturb=True)
    keys = [k for k in P if k.startswith("conv_1.")]
    x = torch.randn(1, 4, 4, 32, dtype=D, generator=torch.Generator().manual_seed(1))

    def fn(xx, *vals):
        Pl = dict(P)
        Pl.update(dict(zip(keys, vals)))
        return O.residual_S(xx, Pl, "conv_1.", 3, 3, as_executed=True)
    _fd_check(fn, [x] + [P[k] for k in keys], n_probe=2)


def test_fd_loss_through_softmax():
    g = torch.Generator().manual_seed(9)
    y = torch.softmax(torch.randn(2, 3, 3, 3, dtype=D, generator=g), -1)
    _fd_check(lambda z: O.cce_label_smoothing(y, O.softmax_lastaxis(z)), [torch.randn(2, 3, 3, 3, dtype=D, generator=g)])
    _fd_check(lambda z: O.my_loss_cat(y, O.softmax_lastaxis(z), 3, 3), [torch.randn(2, 3, 3, 3, dtype=D, generator=g)])


def test_transunet_variant_of_the_oracle():
    """TBI_TransUNet.py: BatchNormalization for LayerNormalization (:304,426,465,472,503), conv_4 = 256 channels (:368), mean loss (:546)."""
    P = O.init_vision_transformer_params(channel=1, seed=0, use_vit=True, transunet=True)
    assert P["transformer.embeddings.hybrid_model.conv_4.concats_2.kernel"].shape == (3, 3, 126, 256)
    assert P["transformer.embeddings.patch_embeddings.kernel"].shape == (1, 1, 256, 512)
    for k in ("transformer.embeddings.hybrid_model.conv_1.cardinal_blocks.0.conv1_bn", "decoder.bn1",
              "transformer.embeddings.hybrid_model.conv_2.convtmp_scbn", "transformer.embeddings.hybrid_model.conv_3.cardinal_blocks.2.split.dense1_bn"):
        assert k + ".moving_mean" in P and k + ".moving_variance" in P, k
    # BN inference with fresh statistics is x / sqrt(1 + 1e-3): a cardinal block of the variant on a constant-free input
    x = torch.randn(1, 8, 8, 32, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    pre = "transformer.embeddings.hybrid_model.conv_1.cardinal_blocks.0."
    u = O.conv2d_same(x, P[pre + "conv1.kernel"], P[pre + "conv1.bias"])
    want = O.leaky_relu(u / (1 + 1e-3) ** 0.5)
    got = O.leaky_relu(O._norm(u, P, pre + "conv1_bn", "bn"))
    assert torch.allclose(got, want, atol=1e-12)
    # the loss is the MEAN over pixels (Keras default reduction), i.e. the newer model's sum / (B*H*W)
    xx, y = O.synthetic_batch(1, 32, 32, 1, seed=1)
    loss, probs, _, _ = O.train_step(xx, y, dict(P), {}, 1, use_vit=True, transunet=True)
    assert abs(loss.item() - O.cce_label_smoothing(y, probs).mean().item()) < 1e-12


def test_ksac_as_written_is_a_dilated_conv_with_cumulative_rates():
    """Decoder.py:266-288: the literal slice / pad / add restatement (with ``value`` re-assigned inside the loop over the rates,
    :280-285) equals ordinary 'same' dilated convolutions with the SHARED kernel at dilations (1, 3, 7, 15, 31)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 20, 18, 3, generator=g, dtype=torch.float64)     # every single rate (<= 16) must be smaller than H and W: the
    w = torch.randn(3, 3, 3, 4, generator=g, dtype=torch.float64)       # reference's slice + pad + EnsureShape fails otherwise (:280-286)
    assert O.ksac_effective_dilations() == (1, 3, 7, 15, 31) and O.ksac_effective_dilations(as_written=False) == (1, 2, 4, 8, 16)
    lit = O.kernel_sharing_conv2d_literal(x, w)
    fast = O.kernel_sharing_conv2d(x, w)
    for a, b in zip(lit, fast):
        assert torch.allclose(a, b, atol=1e-12)
    # only the first rate agrees with the as-intended reading
    intended = O.kernel_sharing_conv2d(x, w, as_written=False)
    assert torch.allclose(lit[0], intended[0], atol=1e-12) and not torch.allclose(lit[1], intended[1], atol=1e-6)
    # a delta input shows the taps of branch 1 at distance 3 (= 1 + 2), not 2
    d = torch.zeros(1, 9, 9, 1, dtype=torch.float64)
    d[0, 4, 4, 0] = 1.0
    y = O.kernel_sharing_conv2d_literal(d, torch.ones(3, 3, 1, 1, dtype=torch.float64), dilations=(1, 2))[1][0, :, :, 0]
    assert y[1, 1] == 1 and y[4, 7] == 1 and y[2, 2] == 0 and y.sum() == 9


def test_cce_cached_logits_switch():
    """The two readings of CategoricalCrossentropy on a Keras-softmax output (KERAS["cce_cached_logits"]): identical on ordinary pixels, and
    on a saturated pixel the probability path clips at 1e-7 (finite, -log(1e-7) per missing class) while the cached-logits path returns
    -sum y * log_softmax(logits) unclipped.  The product implements the probability path (the switch is False)."""
    assert O.KERAS["cce_cached_logits"] is False
    z = torch.tensor([[0.3, -1.2, 2.0], [40.0, -40.0, -40.0]], dtype=D)
    y = torch.tensor([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], dtype=D)
    p = O.softmax_lastaxis(z)
    clip = O.cce_label_smoothing(y, p)
    O.KERAS["cce_cached_logits"] = True
    try:
        logit = O.cce_label_smoothing(y, p, logits=z)
    finally:
        O.KERAS["cce_cached_logits"] = False
    assert abs(clip[0].item() - logit[0].item()) < 1e-12                      # ordinary pixel: the same number
    ys = 0.9 * y[1] + 0.1 / 3
    want_logit = -(ys * torch.log_softmax(z[1], -1)).sum().item()             # = 0.9333*80 + 0.0333*80 = 77.33
    assert abs(logit[1].item() - want_logit) < 1e-9 and abs(want_logit - (ys[1] + ys[2]).item() * 80.0) < 1e-6
    want_clip = -(ys[0] * math.log(1 - 1e-7) + (ys[1] + ys[2]) * math.log(1e-7)).item()     # 0.9667 * 16.118 = 15.58
    assert abs(clip[1].item() - want_clip) < 1e-9
