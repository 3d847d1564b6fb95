"""Known-answer tests that pin the CPU restatement of the reference's input pipeline (oracle/input_oracle.py): the rules a
reading of Dataset_2.py / DataAugs.py could get wrong.  The reference ships no fixtures: parity unpinned."""
import random

import numpy as np

import input_oracle as IO


def test_label2vec_three_classes_thresholds():
    lab = np.array([[[0.0, 0.5, 0.95, 0.951, 1.0, 1.04, 1.05, 1.5, 2.0, 2.7]]], dtype=np.float32)
    v = IO.label2vec(lab, 3)[0, 0]
    assert v.shape == (10, 3)
    np.testing.assert_allclose(v[:3], [[1, 0, 0]] * 3)                       # <= 0.95 -> class 0
    np.testing.assert_allclose(v[3:6], [[0, 1, 0]] * 3)                      # (0.95, 1.05) -> class 1, class 2 still 0
    np.testing.assert_allclose(v[6], [0, 1 - 0.05, 0.05], atol=1e-6)         # >= 1.05: class 2 = l-1, class 1 = 1 - class 2
    np.testing.assert_allclose(v[7], [0, 0.5, 0.5], atol=1e-6)
    np.testing.assert_allclose(v[8], [0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(v[9], [0, 0, 1], atol=1e-6)                   # class 2 capped at 1
    v2 = IO.label2vec(lab, 2)[0, 0]
    np.testing.assert_allclose(v2[:, 0] + v2[:, 1], 1.0)


def test_shift_leaves_last_row_and_column_zero():
    img = np.arange(6 * 5 * 2, dtype=np.float64).reshape(6, 5, 2) + 1
    lab = np.arange(30, dtype=np.float32).reshape(6, 5) + 1
    l2, i2 = IO.shift(img, lab, 0, 0, 1)                                     # r = c = 0: identity except the last row/column
    assert (i2[-1] == 0).all() and (i2[:, -1] == 0).all() and (l2[-1] == 0).all() and (l2[:, -1] == 0).all()
    np.testing.assert_array_equal(i2[:-1, :-1], img[:-1, :-1])
    l3, i3 = IO.shift(img, lab, 2, 1, 1)
    np.testing.assert_array_equal(i3[0, 0], img[2, 1])
    assert (i3[4] == 0).all()                                                # row 4 + 2 is outside
    l4, i4 = IO.shift(img, lab, 2, 1, 0)
    np.testing.assert_array_equal(i4[3, 2], img[1, 1])
    assert (i4[:2] == 0).all() and (i4[:, 0] == 0).all()


def test_clip_box_is_open_and_skips_last_row_column():
    img = np.ones((50, 30, 3)); lab = np.ones((50, 30), dtype=np.float32)
    l2, i2 = IO.clip(img.copy(), lab.copy(), 10, 8, 3, 2)
    zero = np.argwhere(l2 == 0)
    assert zero[:, 0].min() == 8 and zero[:, 0].max() == 12 and zero[:, 1].min() == 7 and zero[:, 1].max() == 9   # strict inequalities
    l3, i3 = IO.clip(img.copy(), lab.copy(), 49, 29, 5, 5)
    assert (l3[49] == 1).all() and (l3[:, 29] == 1).all() and l3[48, 28] == 0                                      # loops stop at si-1


def test_image_reduc_as_executed_zeroes_image_where_label_is_zero():
    rng = np.random.default_rng(0)
    lab = (rng.random((12, 9)) > 0.5).astype(np.float32) * rng.random((12, 9)).astype(np.float32)
    img = rng.standard_normal((12, 9, 4))
    l2, i2 = IO.image_reduc(np.concatenate([lab[..., None], img], axis=2), 5)
    np.testing.assert_array_equal(l2, lab.astype(np.float64))                # the label comes back unchanged
    np.testing.assert_array_equal(i2[lab == 0], 0)
    np.testing.assert_array_equal(i2[lab != 0], img[lab != 0])


def test_draw_order_and_counts():
    rng = random.Random(7)
    p = [IO.draw_params(rng) for _ in range(200)]
    assert {len(q["clips"]) for q in p} == {0, 1, 2}
    assert any(q["shift"] is None for q in p) and any(q["shift"] is not None for q in p)
    for q in p:
        assert q["reduc"] == (len(q["clips"]) != 0)                           # both are functions of r % 3
        for (r, c, ra, ca) in q["clips"]:
            assert 0 <= r <= 256 and 0 <= c <= 80 and 20 <= ra <= 40 and 10 <= ca <= 20
        if q["shift"]:
            assert 0 <= q["shift"][0] <= 30 and 0 <= q["shift"][1] <= 12 and q["shift"][2] in (0, 1)
    # the device path's host-side draw consumes the generator identically
    from ultrasound_modeling_amd.DataAugs import draw
    a, b = random.Random(11), random.Random(11)
    for _ in range(50):
        q, d = IO.draw_params(a), draw(b)
        assert (q["reduc"], q["clips"], q["shift"], q["noise"]) == (d["reduc"], d["clips"], d["shift"], d["noise"])
