"""CPU restatement (NumPy) of the reference's host input pipeline: TEST INFRASTRUCTURE ONLY - PARITY UNPINNED.

Follows silverlight6/Ultrasound_Modeling ``Dataset_2.py:6-20`` (label2vec), ``Dataset_2.py:88-114`` (next_train) and
``DataAugs.py:6-102`` (shift / clip / noisy / imageReduc / dataAug) AS EXECUTED, including the quirks a reading of the
intent would miss:
  * the loops of shift and clip stop at ``si-1`` (DataAugs.py:13-14,33-34): the last row and column are never written
    (shift leaves them zero, clip leaves them untouched);
  * imageReduc's dilation never fires (``mask[i, j] > 1`` on a 0/1 mask, DataAugs.py:63), so after the first pass the mask
    is all zero, the label channel is returned unchanged and every image channel is zeroed where the LABEL is exactly 0
    (DataAugs.py:76-78: channel 0 of the concatenated array is the label);
  * dataAug draws r, t, then per clip (r, c, ra, ca), then per shift (r, c, direction) from Python's ``random`` in that
    order (DataAugs.py:83-101); the Gaussian noise comes from ``np.random.normal`` (DataAugs.py:50) and is passed in here
    so that tests can inject it.
Only tests/ may import this file.  The reference cannot be run here (TensorFlow absent) and ships no fixtures.
"""
import random

import numpy as np


def label2vec(label, num_classes):
    """Dataset_2.py:6-20.  label [B,H,W] float32 -> [B,H,W,num_classes]."""
    if num_classes == 3:
        class_2 = np.where(label >= 1.05, label - 1, 0)
        class_2 = np.where(class_2 > 1, 1, class_2)
        class_1 = np.expand_dims(np.where(label > 0.95, 1 - class_2, 0), axis=3)
        class_0 = np.expand_dims(np.where(label <= 0.95, 1, 0), axis=3)
        class_2 = np.expand_dims(class_2, axis=3)
        return np.concatenate((class_0, class_1, class_2), axis=3)
    class_1 = label
    class_0 = 1 - label
    return np.concatenate((np.expand_dims(class_0, axis=3), np.expand_dims(class_1, axis=3)), axis=3)


def shift(image, label, r, c, direction):
    """DataAugs.py:6-24 with the three draws passed in.  -> (label, image)."""
    si = image.shape
    mask2 = np.zeros((si[0], si[1], si[2]), dtype=np.float64)
    mask3 = np.zeros((si[0], si[1]), dtype=np.float64)
    for i in range(0, si[0] - 1):
        for j in range(0, si[1] - 1):
            ii, jj = (i + r, j + c) if direction else (i - r, j - c)
            if 0 <= ii < si[0] and 0 <= jj < si[1]:
                mask2[i, j, :] = image[ii, jj, :]
                mask3[i, j] = label[ii, jj]
    return mask3, mask2


def clip(image, label, r, c, ra, ca):
    """DataAugs.py:27-38 with the four draws passed in (in place, like the reference).  -> (label, image)."""
    si = image.shape
    for i in range(0, si[0] - 1):
        for j in range(0, si[1] - 1):
            if r + ra > i > r - ra and c + ca > j > c - ca:
                image[i, j, :] = 0
                label[i, j] = 0
    return label, image


def image_reduc(image_with_label, t):
    """DataAugs.py:54-79 as executed (see the module docstring).  -> (label, image)."""
    output = image_with_label
    si = output.shape
    mask = np.where(output[:, :, 0] < 0.1, 1, 0)
    mask2 = np.zeros((si[0], si[1]), dtype=np.int32)
    for _ in range(0, t):
        # `if mask[i, j] > 1` (DataAugs.py:63) is never true for a 0/1 mask: nothing is marked
        assert mask.max() <= 1
        mask = mask2
        mask2 = np.zeros((si[0], si[1]), dtype=np.int32)
    output[:, :, 0] = np.where(mask == 1, 0, output[:, :, 0])
    for k in range(1, si[2]):
        output[:, :, k] = np.where(output[:, :, 0] == 0, 0, output[:, :, k])
    return output[:, :, 0], output[:, :, 1:]


def draw_params(rng: random.Random, H=256, W=80):
    """The draws of one dataAug call (DataAugs.py:83-101) in the reference's order, as a dict the device path consumes."""
    r = rng.randint(0, 100000)
    t = rng.randint(0, 100000)
    p = {"reduc": r % 3 != 0, "reduc_t": t % 7 + 2, "clips": [], "shift": None, "noise": bool(t % 3)}
    for _ in range(r % 3):
        p["clips"].append((rng.randint(0, 256), rng.randint(0, 80), rng.randint(20, 40), rng.randint(10, 20)))   # DataAugs.py:28-31 (literal 256 / 80)
    if t % 2:
        p["shift"] = (rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 1))                                  # DataAugs.py:7-9
    return p


def data_aug(image, label, p, gauss=None):
    """DataAugs.py:82-102 with the draws ``p`` (draw_params) and the unit Gaussian field ``gauss`` [H,W,C] injected.
    image [H,W,C] float64, label [H,W] float32 -> (image, label)."""
    image, label = image.copy(), label.copy()
    if p["reduc"]:
        label, image = image_reduc(np.concatenate([np.expand_dims(label, axis=-1), image], axis=2), p["reduc_t"])
    for (r, c, ra, ca) in p["clips"]:
        label, image = clip(image, label, r, c, ra, ca)
    if p["shift"] is not None:
        label, image = shift(image, label, *p["shift"])
    if p["noise"] and gauss is not None:
        image = image + gauss / 5000                                                                              # DataAugs.py:41-51
    return image, label
