"""Known-answer tests of oracle/swin_oracle.py (SwinTransformer.py has no fixtures in the reference: parity unpinned)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import swin_oracle as S


def test_window_partition_and_reverse_are_inverse_for_square_windows():
    x = torch.arange(2 * 8 * 12 * 3, dtype=torch.float64).reshape(2, 8, 12, 3)
    w = S.window_partition(x, 4)
    assert w.shape == (2 * 2 * 3, 4, 4, 3)
    assert torch.equal(w[1], x[0, 0:4, 4:8])                       # second window of image 0: rows 0-3, columns 4-7
    assert torch.equal(S.window_reverse(w, 4, 8, 12), x)


def test_relative_position_index_and_mask():
    idx = S.relative_position_index(2)                             # SwinTransformer.py:84-93 for a 2x2 window: (2*2-1)^2 = 9 bins
    assert idx.shape == (4, 4) and idx.diagonal().tolist() == [4, 4, 4, 4]
    assert idx[0].tolist() == [4, 3, 1, 0] and idx[3].tolist() == [8, 7, 5, 4]
    m = S.shift_mask(8, 8, 4, 2)
    assert m.shape == (4, 16, 16) and float(m[0].abs().sum()) == 0.0          # top-left window lies in one region
    assert set(m.unique().tolist()) == {0.0, -100.0}
    # bottom-right window: 4 regions of 2x2 tokens -> each row sees its own 4 tokens
    assert (m[3] == 0).sum(dim=1).tolist() == [4] * 16


def test_block_window_rule_and_shapes():
    cfg = dict(patch_size=4, embed_dim=16, depths=[2, 2], num_heads=[2, 4], window_size=4)
    assert S.block_window((8, 8), 4, 0) == (4, 0) and S.block_window((8, 8), 4, 1) == (4, 2) and S.block_window((4, 4), 4, 1) == (4, 0)
    P = S.init_swin_params(cfg, in_chans=1, seed=0)
    x = torch.randn(2, 32, 32, 1, dtype=torch.float64)
    out, feats = S.swin_forward(x, P, cfg)
    assert out.shape == (2, 32) and [tuple(f.shape) for f in feats] == [(2, 64, 16)]
    # zero bias table + no shift: permuting the tokens INSIDE every window permutes the block output the same way
    for k in P:
        if k.endswith("relative_position_bias_table"):
            P[k].zero_()
    t = torch.randn(1, 16, 16, dtype=torch.float64)               # one 4x4 window, 16 channels
    perm = torch.randperm(16)
    a = S.swin_block(t, P, "layers0/blocks0/", (4, 4), 4, 0, 2)
    b = S.swin_block(t[:, perm], P, "layers0/blocks0/", (4, 4), 4, 0, 2)
    assert torch.allclose(a[:, perm], b, atol=1e-10)


def test_patch_merging_order():
    x = torch.arange(1 * 4 * 4 * 1, dtype=torch.float64).reshape(1, 16, 1)
    P = {"l/downsample/norm/gamma": torch.ones(4, dtype=torch.float64), "l/downsample/norm/beta": torch.zeros(4, dtype=torch.float64),
         "l/downsample/reduction/kernel": torch.eye(4, dtype=torch.float64)[:, :2]}
    y = S.patch_merging(x, P, "l/", (4, 4))
    # first output token gathers pixels (0,0), (1,0), (0,1), (1,1) = values 0, 4, 1, 5 in THAT order (:280-284)
    v = torch.tensor([0.0, 4.0, 1.0, 5.0], dtype=torch.float64)
    want = (v - v.mean()) / torch.sqrt(v.var(unbiased=False) + 1e-5)
    assert torch.allclose(y[0, 0], want[:2], atol=1e-12)
