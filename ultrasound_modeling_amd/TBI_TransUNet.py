"""The surface of the reference's ``TBI_TransUNet.py`` (BASELINE configs[3]: ResNeSt encoder + ViT bottleneck + decoder,
512x512, batch 8, one GPU): an older, self-contained copy of the ResNest.py / Decoder.py / VisionTransformer.py model with

  * BatchNormalization (inference mode as driven) wherever the newer files use LayerNormalization
    (TBI_TransUNet.py:304 decoder bn1, :426 shortcut, :465 / :472 cardinal, :503 split attention),
  * ``conv_4 = residual_S(outchannel=256)`` (:368), so the 1x1 patch embedding maps 256 -> 512 (:110),
  * ``CategoricalCrossentropy(label_smoothing=0.1)`` with Keras' DEFAULT reduction - the mean over all B*H*W pixels (:546),
  * ``step(x, y, train=False) -> (loss, probabilities)`` (:571-588), clip-by-global-norm 1.0 and Adam, not jitted.

Everything runs on the same gfx950 kernels as the newer model (``VisionTransformer(..., transunet=True)``); the literal 16x5
token grid (:90,:547) is generalised to (H/16, W/16) as everywhere else.
"""
from __future__ import annotations

from typing import Optional

from .VisionTransformer import VisionTransformer as _VisionTransformer


class VisionTransformer(_VisionTransformer):
    """TBI_TransUNet.py:532-592.  Note the reference's argument order here: batch_size comes LAST (:533)."""

    def __init__(self, img_size=(256, 80), num_classes=3, learning_rate=1e-3, weight_decay=1e-4, batch_size=16, *,
                 in_channels: int = 10, device: Optional[str] = None, seed: Optional[int] = 0):
        super().__init__(batch_size, img_size, num_classes, learning_rate, weight_decay, in_channels=in_channels, use_vit=True,
                         device=device, seed=seed, transunet=True)

    def step(self, x, y, train=False):
        """TBI_TransUNet.py:571-588 -> (loss, probabilities); ``train=True`` also clips and applies the gradients."""
        if train:
            return self.train_step(x, y)
        return super().step(x, y)
