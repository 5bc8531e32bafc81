"""TEST INFRASTRUCTURE — CPU restatement of UniDefenseModelEb4 (functional, plain torch).

Every function cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  Works on any float dtype (tests use float32 for golden
comparison and float64 as the "exact" reference for the HIP kernels).

State is a flat ``dict`` with the reference's state-dict key names
(e.g. ``backbone._blocks.6._depthwise_conv.freq_conv.weight``).
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# Architecture table  (model/efficientnet/utils.py:84-129, 461-541; model.py:166-231)
# --------------------------------------------------------------------------------------
_BASE_BLOCKS = [  # (repeat, k, stride, expand, in, out, se)   utils.py:506-514
    (1, 3, 1, 1, 32, 16, 0.25),
    (2, 3, 2, 6, 16, 24, 0.25),
    (2, 5, 2, 6, 24, 40, 0.25),
    (3, 3, 2, 6, 40, 80, 0.25),
    (3, 5, 1, 6, 80, 112, 0.25),
    (4, 5, 2, 6, 112, 192, 0.25),
    (1, 3, 1, 6, 192, 320, 0.25),
]
_COEF = {  # width, depth, res, dropout      utils.py:471-482
    "efficientnet-b0": (1.0, 1.0, 224, 0.2),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4),
}
DELIMITER = {"efficientnet-b4": [2, 6, 10, 16, 22, 30, 32]}  # model/unidefense.py:22-24


def _round_filters(f, width, divisor=8):
    # utils.py:84-110
    f = f * width
    new_f = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new_f < 0.9 * f:
        new_f += divisor
    return int(new_f)


def _round_repeats(r, depth):
    return int(math.ceil(depth * r))  # utils.py:113-128


def _same_pad(size, k, s):
    """Static TF-'SAME' pad for a given *design* image size (utils.py:264-275, exp.py:26-39).
    Returns (lo, hi) for one axis."""
    o = math.ceil(size / s)
    p = max((o - 1) * s + (k - 1) + 1 - size, 0)
    return p // 2, p - p // 2


def eb4_arch(name: str = "efficientnet-b4", freq_norm: Optional[str] = "ortho"):
    """List of per-block dicts + stem/head info.  Pads are computed for the design
    resolution (380 for b4) exactly like the reference does at construction time,
    whatever resolution is fed later (SURVEY.md §7 'Static SAME padding')."""
    width, depth, res, _ = _COEF[name]
    size = res
    stem_out = _round_filters(32, width)
    lo, hi = _same_pad(size, 3, 2)
    arch = {"stem": dict(cin=3, cout=stem_out, k=3, s=2, pad=(lo, hi, lo, hi))}
    size = math.ceil(size / 2)
    blocks = []
    nb = len(_BASE_BLOCKS)
    for bid, (r, k, s, e, i, o, se) in enumerate(_BASE_BLOCKS):
        i, o, r = _round_filters(i, width), _round_filters(o, width), _round_repeats(r, depth)
        sf = freq_norm if bid not in (0, 1, nb - 1) else None  # model.py:205,214
        for rep in range(r):
            cin = i if rep == 0 else o
            st = s if rep == 0 else 1
            lo, hi = _same_pad(size, k, st)
            blocks.append(dict(cin=cin, cout=o, cexp=cin * e, expand=e, k=k, s=st,
                               pad=(lo, hi, lo, hi),  # (left, right, top, bottom)
                               cse=max(1, int(cin * se)), sf=sf,
                               skip=(st == 1 and cin == o)))
            if rep == 0:
                size = math.ceil(size / s)
    arch["blocks"] = blocks
    arch["head"] = dict(cin=blocks[-1]["cout"], cout=_round_filters(1280, width))
    arch["bn_eps"] = 1e-3
    arch["bn_mom"] = 1 - 0.99
    arch["drop_connect_rate"] = 0.2
    arch["delimiter"] = DELIMITER[name]
    return arch


# --------------------------------------------------------------------------------------
# Primitive ops
# --------------------------------------------------------------------------------------
def swish(x: Tensor) -> Tensor:
    """model/efficientnet/utils.py:66-82 (forward; autograd gives the same backward)."""
    return x * torch.sigmoid(x)


def batch_norm(x: Tensor, sd: Dict[str, Tensor], prefix: str, training: bool, eps: float,
               momentum: float = 0.01, update_running: bool = False) -> Tensor:
    """nn.BatchNorm2d/1d (model.py:67,77,91,186,222; unidefense.py:104): batch statistics
    (biased var) in training, running stats in eval."""
    w, b = sd.get(prefix + ".weight"), sd.get(prefix + ".bias")          # absent: norm(affine=False) (unidefense.py:38,61,116)
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        if update_running:
            n = x.numel() / x.shape[1]
            sd[prefix + ".running_mean"] = (1 - momentum) * sd[prefix + ".running_mean"] + momentum * mean.detach()
            sd[prefix + ".running_var"] = (1 - momentum) * sd[prefix + ".running_var"] + \
                momentum * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    xh = (x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + eps)
    return xh if w is None else xh * w.reshape(shape) + b.reshape(shape)


def instance_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """nn.InstanceNorm2d(affine=affine), no running stats (model/unidefense.py:54,61); w = b = None: affine=False."""
    mean = x.mean((2, 3), keepdim=True)
    var = x.var((2, 3), unbiased=False, keepdim=True)
    xh = (x - mean) / torch.sqrt(var + eps)
    return xh if w is None else xh * w.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)


def conv_static_same(x: Tensor, w: Tensor, stride: int, pad, groups: int = 1, bias=None) -> Tensor:
    """Conv2dStaticSamePadding.forward (model/efficientnet/utils.py:277-280): ZeroPad2d with
    the construction-time pads, then an unpadded conv."""
    if any(pad):
        x = F.pad(x, list(pad))
    return F.conv2d(x, w, bias, stride, 0, 1, groups)


def rfft2_cat(x: Tensor, norm) -> Tensor:
    """rfft2 then cat([re, im], dim=1)   (exp.py:55-56; unidefense.py:130-131)."""
    f = torch.fft.rfft2(x, norm=norm)
    return torch.cat([f.real, f.imag], dim=1)


def irfft2_split(y: Tensor, size, norm) -> Tensor:
    """split channels in two -> complex -> irfft2(s=size)   (exp.py:59-60; unidefense.py:142-145)."""
    re, im = torch.tensor_split(y, 2, dim=1)
    return torch.fft.irfft2(torch.complex(re.contiguous(), im.contiguous()), s=tuple(size), norm=norm)


def sfconv(x: Tensor, sd: Dict[str, Tensor], prefix: str, stride: int, pad, norm) -> Tensor:
    """SFConv2dStaticSamePadding.forward (model/efficientnet/exp.py:46-65), depthwise."""
    w = sd[prefix + ".weight"]
    spat = conv_static_same(x, w, stride, pad, groups=w.shape[0])
    fx = rfft2_cat(x, norm)
    fx = F.conv2d(fx, sd[prefix + ".freq_conv.weight"])
    fx = irfft2_split(fx, x.shape[-2:], norm)
    if fx.shape[-2:] != spat.shape[-2:]:
        fx = F.adaptive_avg_pool2d(fx, spat.shape[-2:])
    a = torch.sigmoid(sd[prefix + ".sf_coef"])
    return (1.0 - a) * spat + a * fx


def mbconv(x: Tensor, sd: Dict[str, Tensor], prefix: str, blk: dict, training: bool, eps: float,
           keep_mask: Optional[Tensor] = None, keep_prob: float = 1.0) -> Tensor:
    """MBConvBlock.forward (model/efficientnet/model.py:94-135).
    keep_mask: optional [N] 0/1 tensor = floor(keep_prob + U) of drop_connect (utils.py:131-156)."""
    inp = x
    if blk["expand"] != 1:
        x = F.conv2d(x, sd[prefix + "._expand_conv.weight"])
        x = swish(batch_norm(x, sd, prefix + "._bn0", training, eps))
    if blk["sf"] is not None:
        x = sfconv(x, sd, prefix + "._depthwise_conv", blk["s"], blk["pad"], blk["sf"])
    else:
        w = sd[prefix + "._depthwise_conv.weight"]
        x = conv_static_same(x, w, blk["s"], blk["pad"], groups=w.shape[0])
    x = swish(batch_norm(x, sd, prefix + "._bn1", training, eps))
    # squeeze-excite (model.py:117-122)
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.conv2d(s, sd[prefix + "._se_reduce.weight"], sd[prefix + "._se_reduce.bias"])
    s = swish(s)
    s = F.conv2d(s, sd[prefix + "._se_expand.weight"], sd[prefix + "._se_expand.bias"])
    x = torch.sigmoid(s) * x
    x = F.conv2d(x, sd[prefix + "._project_conv.weight"])
    x = batch_norm(x, sd, prefix + "._bn2", training, eps)
    if blk["skip"]:
        if training and keep_mask is not None:
            x = x / keep_prob * keep_mask.reshape(-1, 1, 1, 1).to(x.dtype)
        x = x + inp
    return x


def interpolate(x: Tensor, size) -> Tensor:
    """model/unidefense.py:16 — bilinear, align_corners=True."""
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)


def decoder_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, last: bool) -> Tensor:
    """dec_block{1,2,3} (model/unidefense.py:59-102): conv3x3-IN-swish, convT(k3,s2,p1,op1)-IN-swish,
    conv3x3-IN-swish [, conv3x3 -> tanh]."""
    g = sd.get                                           # conv biases (bias=True) and norm affines (affine=True) are optional keys
    x = F.conv2d(x, sd[prefix + ".0.weight"], g(prefix + ".0.bias"), 1, 1)
    x = swish(instance_norm(x, g(prefix + ".1.weight"), g(prefix + ".1.bias")))
    x = F.conv_transpose2d(x, sd[prefix + ".3.weight"], g(prefix + ".3.bias"), 2, 1, 1)
    x = swish(instance_norm(x, g(prefix + ".4.weight"), g(prefix + ".4.bias")))
    x = F.conv2d(x, sd[prefix + ".6.weight"], g(prefix + ".6.bias"), 1, 1)
    x = swish(instance_norm(x, g(prefix + ".7.weight"), g(prefix + ".7.bias")))
    if last:
        x = torch.tanh(F.conv2d(x, sd[prefix + ".9.weight"], g(prefix + ".9.bias"), 1, 1))
    return x


def dynamic_filter(x: Tensor, diff: Tensor, sd: Dict[str, Tensor], prefix: str, training: bool,
                   pad: int) -> Dict[str, Tensor]:
    """The EfficientNet model's filters use MemoryEfficientSwish (model/unidefense.py:56,115-118)."""
    return dynamic_filter_generic(x, diff, sd, prefix, training, pad, swish)


def dynamic_filter_generic(x: Tensor, diff: Tensor, sd: Dict[str, Tensor], prefix: str, training: bool,
                           pad: int, act) -> Dict[str, Tensor]:
    """FrequencyDynamicFilter / SpatialDynamicFilter .forward (model/modules.py:91-105, 120-134).
    att_norm = nn.BatchNorm2d with default eps 1e-5 (unidefense.py:55)."""
    p = F.conv2d(x, sd[prefix + ".layer1.0.weight"], sd.get(prefix + ".layer1.0.bias"), 1, pad)
    p = act(batch_norm(p, sd, prefix + ".layer1.1", training, 1e-5))
    pre = torch.cat([p.mean(1, keepdim=True), p.max(1, keepdim=True).values, diff], dim=1)
    mask = torch.sigmoid(F.conv2d(pre, sd[prefix + ".layer2.0.weight"], sd.get(prefix + ".layer2.0.bias")))
    return {"mask": mask, "out": mask * x, "proj": p}


def attention(pred: Tensor, x: Tensor, emb: Tensor, sd: Dict[str, Tensor], training: bool,
              norm, emb_keep: Optional[Tensor], drop_rate: float) -> Dict[str, Tensor]:
    """UniDefenseModelEb4.attention (model/unidefense.py:125-157).
    emb_keep: optional 0/1 mask (emb shape) for ``self.dropout(embedding.clone())`` (:155)."""
    size = emb.shape[-2:]
    pred = interpolate(pred, size)
    x = interpolate(x, size)
    freq_diff = torch.abs(rfft2_cat(pred, norm) - rfft2_cat(x, norm))
    emb_freq = rfft2_cat(emb, norm)
    ff = dynamic_filter(emb_freq, freq_diff, sd, "freq_filter", training, 0)
    freq_filtered = irfft2_split(ff["out"], size, norm)
    spat_diff = torch.abs(pred - x)
    sf = dynamic_filter(emb, spat_diff, sd, "spat_filter", training, 1)
    a = torch.sigmoid(sd["fuse_coef"])
    out = (1.0 - a) * sf["out"] + a * freq_filtered
    e = emb
    if training and drop_rate > 0:
        assert emb_keep is not None, "training with drop_rate>0 needs an explicit emb_keep mask"
        e = emb * emb_keep.to(emb.dtype) / (1.0 - drop_rate)
    out = out + e

    def top2_gap(p):   # torch.max's gradient is discontinuous where the two largest channels (nearly) tie
        t = p.detach().topk(2, dim=1).values
        return ((t[:, 0] - t[:, 1]) / t[:, 0].abs().clamp_min(1e-30)).min()
    return {"out": out, "freq_mask": ff["mask"], "spat_mask": sf["mask"],
            "max_gap": torch.minimum(top2_gap(ff["proj"]), top2_gap(sf["proj"]))}


def forward_eb4(sd: Dict[str, Tensor], x: Tensor, training: bool = False, drop_rate: float = 0.2,
                freq_norm: Optional[str] = "ortho", rng: Optional[dict] = None,
                arch: Optional[dict] = None) -> dict:
    """UniDefenseModelEb4.forward without input perturbation (model/unidefense.py:174-256, the
    `noise_x = x` branch :199-200).

    rng (training only) carries the explicit Bernoulli draws the reference takes from the torch
    global generator — all optional, default "keep everything":
      'drop_connect': {block_idx: keep[N] 0/1}      (utils.py:131-156, unidefense.py:165-171)
      'dec_keep'    : keep mask like x_b4           (F.dropout(x_b4, 0.2), unidefense.py:213)
      'emb_keep'    : keep mask like x_b5           (self.dropout(embedding.clone()), :155)
      'feat_keep'   : keep mask [N,1792]            (self.dropout(x_out) IN-PLACE, :230 — it also
                       overwrites loss_dict['factorization'], which aliases x_out, :229)
    """
    arch = arch or eb4_arch(freq_norm=freq_norm)
    rng = rng or {}
    eps = arch["bn_eps"]
    st = arch["stem"]
    # rng['noise_x']: the perturbed input of the second pass; it feeds the stem only, the attention residuals
    # and the reconstruction losses keep the clean x (model/unidefense.py:200, :219, :243-248)
    h = conv_static_same(rng.get("noise_x", x), sd["backbone._conv_stem.weight"], st["s"], st["pad"])
    h = swish(batch_norm(h, sd, "backbone._bn0", training, eps))
    nblk = len(arch["blocks"])
    delim = arch["delimiter"]

    def run(h, lo, hi):
        for idx in range(lo, hi):
            rate = arch["drop_connect_rate"] * float(idx) / nblk
            keep = rng.get("drop_connect", {}).get(idx) if training else None
            h = mbconv(h, sd, f"backbone._blocks.{idx}", arch["blocks"][idx], training, eps,
                       keep_mask=keep, keep_prob=1.0 - rate)
        return h

    x_b0 = run(h, 0, delim[0])
    x_b1 = run(x_b0, delim[0], delim[1])
    x_b2 = run(x_b1, delim[1], delim[2])
    x_b3 = run(x_b2, delim[2], delim[3])
    x_b4 = run(x_b3, delim[3], delim[4])

    d_in = x_b4
    if training:
        keep = rng.get("dec_keep")
        if keep is not None:
            d_in = x_b4 * keep.to(x.dtype) / (1.0 - 0.2)   # hard-coded p = 0.2 (:213)
    dec1 = decoder_block(d_in, sd, "dec_block1", False)
    dec2 = decoder_block(dec1, sd, "dec_block2", False)
    dec3 = decoder_block(dec2, sd, "dec_block3", True)

    x_b5 = run(x_b4, delim[4], delim[5])
    att = attention(dec3.detach(), x, x_b5, sd, training, freq_norm,
                    rng.get("emb_keep"), drop_rate if rng.get("emb_keep") is not None else 0.0)
    x_b6 = run(att["out"], delim[5], delim[6])

    h = F.conv2d(x_b6, sd["backbone._conv_head.weight"])
    h = swish(batch_norm(h, sd, "backbone._bn1", training, eps))
    pooled = h.mean((2, 3))
    h = batch_norm(pooled, sd, "bottleneck", training, 1e-5)
    if training and rng.get("feat_keep") is not None:
        h = h * rng["feat_keep"].to(h.dtype) / (1.0 - drop_rate)   # in-place in the reference
    loss_dict = {"factorization": h}
    loss_dict["triplet"] = [x_b4.mean((2, 3)), dec1.mean((2, 3)), dec2.mean((2, 3))]
    loss_dict["freq_mask"] = att["freq_mask"]
    loss_dict["spat_mask"] = att["spat_mask"]
    cls_out = F.linear(h, sd["classifier.fc.weight"], sd["classifier.fc.bias"])

    rec = interpolate(dec3, x.shape[-2:])
    loss_dict["spatial"] = torch.abs(rec - x).mean((1, 2, 3))
    tmp = torch.abs(rfft2_cat(rec, freq_norm) - rfft2_cat(x, freq_norm))
    t_re, t_im = tmp.tensor_split(2, dim=1)
    loss_dict["freq"] = (t_re + t_im).mean((1, 2, 3))
    return {"cls_out": cls_out, "rec": rec, "loss_dict": loss_dict, "_max_gap": att["max_gap"],
            "_feats": {"x_b0": x_b0, "x_b1": x_b1, "x_b2": x_b2, "x_b3": x_b3, "x_b4": x_b4,
                       "x_b5": x_b5, "dec1": dec1, "dec2": dec2, "dec3": dec3, "att_out": att["out"],
                       "x_b6": x_b6, "pooled": pooled}}


# --------------------------------------------------------------------------------------
# State-dict shapes (so that weights can be generated without the reference present)
# --------------------------------------------------------------------------------------
def eb4_state_shapes(num_classes: int = 2, freq_norm: Optional[str] = "ortho", bias: bool = False,
                     affine: bool = True) -> Dict[str, tuple]:
    """{key: shape} for UniDefenseModelEb4 — 802 keys with the default bias=False, affine=True (SURVEY.md §5 'Checkpoint');
    bias / affine: the constructor variants of model/unidefense.py:36-38 (decoder and filter convs, their norms)."""
    arch = eb4_arch(freq_norm=freq_norm)
    sh: Dict[str, tuple] = {}

    def bn(prefix, c):
        sh[prefix + ".weight"] = (c,)
        sh[prefix + ".bias"] = (c,)
        sh[prefix + ".running_mean"] = (c,)
        sh[prefix + ".running_var"] = (c,)
        sh[prefix + ".num_batches_tracked"] = ()

    st = arch["stem"]
    sh["backbone._conv_stem.weight"] = (st["cout"], 3, 3, 3)
    bn("backbone._bn0", st["cout"])
    for i, b in enumerate(arch["blocks"]):
        p = f"backbone._blocks.{i}"
        if b["expand"] != 1:
            sh[p + "._expand_conv.weight"] = (b["cexp"], b["cin"], 1, 1)
            bn(p + "._bn0", b["cexp"])
        sh[p + "._depthwise_conv.weight"] = (b["cexp"], 1, b["k"], b["k"])
        if b["sf"] is not None:
            sh[p + "._depthwise_conv.sf_coef"] = ()
            sh[p + "._depthwise_conv.freq_conv.weight"] = (2 * b["cexp"], 2 * b["cexp"], 1, 1)
        bn(p + "._bn1", b["cexp"])
        sh[p + "._se_reduce.weight"] = (b["cse"], b["cexp"], 1, 1)
        sh[p + "._se_reduce.bias"] = (b["cse"],)
        sh[p + "._se_expand.weight"] = (b["cexp"], b["cse"], 1, 1)
        sh[p + "._se_expand.bias"] = (b["cexp"],)
        sh[p + "._project_conv.weight"] = (b["cout"], b["cexp"], 1, 1)
        bn(p + "._bn2", b["cout"])
    hd = arch["head"]
    sh["backbone._conv_head.weight"] = (hd["cout"], hd["cin"], 1, 1)
    bn("backbone._bn1", hd["cout"])

    def dec(prefix, cin, cout, last):
        convs = [("0", (cout, cin, 3, 3)), ("3", (cout, cout, 3, 3)), ("6", (cout, cout, 3, 3))] + ([("9", (3, cout, 3, 3))] if last else [])
        for i, shp in convs:
            sh[f"{prefix}.{i}.weight"] = shp
            if bias:
                sh[f"{prefix}.{i}.bias"] = (shp[1] if i == "3" else shp[0],)          # ConvTranspose2d weight is [Cin, Cout, k, k]
        if affine:
            for i in ("1", "4", "7"):
                sh[f"{prefix}.{i}.weight"] = (cout,)
                sh[f"{prefix}.{i}.bias"] = (cout,)

    dec("dec_block1", 160, 80, False)
    dec("dec_block2", 80, 40, False)
    dec("dec_block3", 40, 20, True)
    bn("bottleneck", hd["cout"])
    sh["classifier.fc.weight"] = (num_classes, hd["cout"])
    sh["classifier.fc.bias"] = (num_classes,)
    d = 272
    def bn_filter(prefix, c):
        bn(prefix, c)
        if not affine:
            del sh[prefix + ".weight"], sh[prefix + ".bias"]
    sh["freq_filter.layer1.0.weight"] = (2 * d, 2 * d, 1, 1)
    bn_filter("freq_filter.layer1.1", 2 * d)
    sh["freq_filter.layer2.0.weight"] = (1, 8, 1, 1)
    sh["spat_filter.layer1.0.weight"] = (d, d, 3, 3)
    bn_filter("spat_filter.layer1.1", d)
    sh["spat_filter.layer2.0.weight"] = (1, 5, 1, 1)
    if bias:
        sh["freq_filter.layer1.0.bias"] = (2 * d,)
        sh["freq_filter.layer2.0.bias"] = (1,)
        sh["spat_filter.layer1.0.bias"] = (d,)
        sh["spat_filter.layer2.0.bias"] = (1,)
    sh["fuse_coef"] = ()
    return sh
