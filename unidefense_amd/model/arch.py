"""EfficientNet block table of the UniDefense backbone.

Restates the construction-time arithmetic of the reference (model/efficientnet/utils.py:84-129 round_filters /
round_repeats, :461-541 coefficients and block strings, :264-275 static 'SAME' pads; model/efficientnet/
model.py:189-215 block expansion and the SFConv selection rule).  The pads are fixed for the DESIGN
resolution (380 px for b4) whatever resolution is fed later, exactly like the reference.
"""
import math
from typing import Optional

# (repeat, kernel, stride, expand, in, out, se_ratio)           utils.py:506-514
_BASE_BLOCKS = [
    (1, 3, 1, 1, 32, 16, 0.25),
    (2, 3, 2, 6, 16, 24, 0.25),
    (2, 5, 2, 6, 24, 40, 0.25),
    (3, 3, 2, 6, 40, 80, 0.25),
    (3, 5, 1, 6, 80, 112, 0.25),
    (4, 5, 2, 6, 112, 192, 0.25),
    (1, 3, 1, 6, 192, 320, 0.25),
]

# width, depth, resolution, dropout                              utils.py:471-482
COEFFICIENTS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2),
    "efficientnet-b1": (1.0, 1.1, 240, 0.2),
    "efficientnet-b2": (1.1, 1.2, 260, 0.3),
    "efficientnet-b3": (1.2, 1.4, 300, 0.3),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4),
    "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5),
    "efficientnet-b7": (2.0, 3.1, 600, 0.5),
}

DELIMITER_DICT = {"efficientnet-b4": [2, 6, 10, 16, 22, 30, 32]}     # model/unidefense.py:22-24


def round_filters(filters, width, divisor=8):
    filters = filters * width
    new_filters = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new_filters < 0.9 * filters:
        new_filters += divisor
    return int(new_filters)


def round_repeats(repeats, depth):
    return int(math.ceil(depth * repeats))


def same_pad(size, k, s):
    """(lo, hi) zero padding of one axis for TF-'SAME' at image size `size`."""
    out = math.ceil(size / s)
    p = max((out - 1) * s + (k - 1) + 1 - size, 0)
    return p // 2, p - p // 2


class BlockSpec:
    __slots__ = ("cin", "cout", "cexp", "expand", "k", "stride", "pad", "cse", "sf_norm", "skip")

    def __init__(self, **kw):
        for k_, v in kw.items():
            setattr(self, k_, v)


def build_arch(name: str, freq_norm: Optional[str], image_size: Optional[int] = None):
    width, depth, res, _ = COEFFICIENTS[name]
    size = image_size or res
    stem_out = round_filters(32, width)
    lo, hi = same_pad(size, 3, 2)
    stem = dict(cout=stem_out, k=3, stride=2, pad=(lo, hi, lo, hi))
    size = math.ceil(size / 2)
    blocks = []
    nb = len(_BASE_BLOCKS)
    for bid, (r, k, s, e, i, o, se) in enumerate(_BASE_BLOCKS):
        i, o, r = round_filters(i, width), round_filters(o, width), round_repeats(r, depth)
        sf = freq_norm if bid not in (0, 1, nb - 1) else None
        for rep in range(r):
            cin = i if rep == 0 else o
            st = s if rep == 0 else 1
            lo, hi = same_pad(size, k, st)
            blocks.append(BlockSpec(cin=cin, cout=o, cexp=cin * e, expand=e, k=k, stride=st,
                                    pad=(lo, hi, lo, hi), cse=max(1, int(cin * se)), sf_norm=sf,
                                    skip=(st == 1 and cin == o)))
            if rep == 0:
                size = math.ceil(size / s)
    head = dict(cin=blocks[-1].cout, cout=round_filters(1280, width))
    return dict(stem=stem, blocks=blocks, head=head, bn_eps=1e-3, bn_momentum=1 - 0.99,
                drop_connect_rate=0.2)
