"""TEST INFRASTRUCTURE — CPU restatement of UniDefenseModelRes18 (functional, plain torch).

Reference: model/unidefense.py:259-436 (model), model/resnet/exp.py:21-54 (SFConv2d), :79-149 (BasicBlock),
:273-321 (make_blocks), :395-440 (stem), model/resnet/module_exp.py:8-32 (ExtractorRes18), :62-111 (embedders).
State keys follow the reference's state dict (177 keys).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .eb4 import batch_norm, instance_norm, rfft2_cat, irfft2_split, interpolate, dynamic_filter_generic

Tensor = torch.Tensor


def relu_site(x: Tensor, name: str, pins: Optional[dict], tie_tol: float = 1e-4) -> Tensor:
    """F.relu — or, when `pins` holds a recorded on/off pattern for this site, x * pattern after checking that the
    pattern departs from (x > 0) only on near-ties (|x| <= tie_tol * max|x|).  A ReLU network's gradient is
    piecewise constant in the activations' signs; two correct fp32 evaluations of a batch with ~1e7 units always
    disagree on a few units that sit within rounding of 0, and each such flip moves some weight gradients by
    ~1e-2.  Pinning the pattern to the implementation under test compares both gradients on the same linear piece
    (same idea as the injected dropout masks and the pinned max-pool winners)."""
    if not pins or name not in pins:
        return F.relu(x)
    m = pins[name].to(torch.bool)
    assert m.shape == x.shape, (name, m.shape, x.shape)
    pins.setdefault("_used", set()).add(name)
    dis = m != (x > 0)
    # bookkeeping for the tests: how many units the pinned pattern moves across 0, out of how many
    st = _dep(pins)
    st["relu_flips"] += int(dis.sum())
    st["relu_units"] += dis.numel()
    if bool(dis.any()):
        worst = (x.detach().abs()[dis].max() / x.detach().abs().max()).item()
        assert worst <= tie_tol, f"pinned ReLU pattern at {name} flips a unit that is not a near-tie ({worst:.3e})"
    return x * m.to(x.dtype)


def _dep(pins):
    """The departure counters of a pins dict (created on first use); None without pins."""
    if not pins:
        return None
    return pins.setdefault("_departures", {"relu_flips": 0, "relu_units": 0, "pool_moves": 0, "pool_windows": 0})


def sfconv2d(x: Tensor, sd: Dict[str, Tensor], prefix: str, stride: int, norm) -> Tensor:
    """SFConv2d.forward (model/resnet/exp.py:36-54): dense 3x3 conv (pad 1) + spectral 1x1 branch."""
    spat = F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride, 1)          # bias: the spatial branch only (exp.py:33,39)
    fx = rfft2_cat(x, norm)
    fx = F.conv2d(fx, sd[prefix + ".freq_conv.weight"])
    fx = irfft2_split(fx, x.shape[-2:], norm)
    if fx.shape[-2:] != spat.shape[-2:]:
        fx = F.adaptive_avg_pool2d(fx, spat.shape[-2:])
    a = torch.sigmoid(sd[prefix + ".sf_coef"])
    return (1.0 - a) * spat + a * fx


def _conv(x, sd, prefix, stride, norm):
    if prefix + ".sf_coef" in sd:
        return sfconv2d(x, sd, prefix, stride, norm)
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride, 1)


def basic_block(x: Tensor, sd, prefix: str, stride: int, training: bool, norm, pins=None) -> Tensor:
    """BasicBlock.forward (model/resnet/exp.py:127-149)."""
    sc = x
    y = _conv(x, sd, prefix + ".conv1", stride, norm)
    y = relu_site(batch_norm(y, sd, prefix + ".bn1", training, 1e-5), prefix + ".bn1", pins)
    y = _conv(y, sd, prefix + ".conv2", 1, norm)
    y = batch_norm(y, sd, prefix + ".bn2", training, 1e-5)
    if prefix + ".downsample.0.weight" in sd:
        sc = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0)
        sc = batch_norm(sc, sd, prefix + ".downsample.1", training, 1e-5)
    return relu_site(y + sc, prefix + ".add", pins)


def extractor(x: Tensor, sd, training: bool, norm, pins=None):
    """ExtractorRes18.forward (model/resnet/module_exp.py:22-32): 7x7/2 stem WITHOUT max-pool, layer1..3,
    concat of the avg-pooled layer1/layer2 outputs with layer3."""
    h = F.conv2d(x, sd["extractor.conv1.weight"], None, 2, 3)
    h = relu_site(batch_norm(h, sd, "extractor.bn1", training, 1e-5), "extractor.bn1", pins)
    p1 = basic_block(basic_block(h, sd, "extractor.layer1.0", 1, training, None, pins), sd, "extractor.layer1.1", 1,
                     training, None, pins)
    p2 = basic_block(basic_block(p1, sd, "extractor.layer2.0", 2, training, norm, pins), sd, "extractor.layer2.1", 1,
                     training, norm, pins)
    p3 = basic_block(basic_block(p2, sd, "extractor.layer3.0", 2, training, norm, pins), sd, "extractor.layer3.1", 1,
                     training, norm, pins)
    size = p3.shape[-2:]
    return p3, torch.cat([F.adaptive_avg_pool2d(p1, size), F.adaptive_avg_pool2d(p2, size), p3], dim=1)


def max_pool_3s2_pinned(z: Tensor, sel: Tensor, tie_tol: float = 1e-4, stats: Optional[dict] = None) -> Tensor:
    """F.max_pool2d(z, 3, 2, 1) with the winner of every window pinned to `sel` ([N,C,Ho,Wo], value kh*3+kw).
    A 3x3 max over ~2.6e5 windows always holds a few top-2 gaps near 1e-6, which two correct fp32
    evaluations resolve differently; pinning the selection (after checking that every pinned winner IS a maximum
    up to tie_tol) lets the gradients of both sides be compared for the same piecewise-linear branch."""
    n, c, h, w = z.shape
    u = F.unfold(z, 3, padding=1, stride=2).view(n, c, 9, -1)
    valid = F.unfold(torch.ones(1, 1, h, w, dtype=z.dtype), 3, padding=1, stride=2).view(1, 1, 9, -1) > 0
    u = torch.where(valid, u, torch.full_like(u, -1e30))
    y = u.gather(2, sel.reshape(n, c, 1, -1).long()).squeeze(2)
    if stats is not None:           # windows whose pinned winner is not this evaluation's own arg-max (near-ties only)
        stats["pool_moves"] += int((u.argmax(2) != sel.reshape(n, c, -1).long()).sum())
        stats["pool_windows"] += sel.numel()
    # relative to the activations' scale: after ~50 layers the fp32 path's values carry ~1e-5 relative error (observed
    # 1e-5 .. 2e-5 of max|z| on the UDR50 embedder pool, depending on the summation order of the kernels before it);
    # same near-tie tolerance as relu_site
    worst = (u.max(2).values - y).max().item() / max(z.detach().abs().max().item(), 1e-30)
    assert worst <= tie_tol, f"pinned max-pool selection is not an arg-max (off by {worst:.3e} of max|z|)"
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return y.view(n, c, ho, wo)


def emb_block1(x: Tensor, sd, training: bool, pool_sel: Optional[Tensor] = None, pins=None) -> Tensor:
    """EmbedderRes18Layer1.forward (module_exp.py:77-89); its SFConv2d has freq_norm=None (:68)."""
    o = F.conv2d(x, sd["emb_block1.conv1.weight"], sd.get("emb_block1.conv1.bias"), 2, 1)
    o = relu_site(batch_norm(o, sd, "emb_block1.norm1", training, 1e-5), "emb_block1.norm1", pins)
    o = sfconv2d(o, sd, "emb_block1.conv2", 1, None)
    o = batch_norm(o, sd, "emb_block1.norm2", training, 1e-5)
    idt = F.conv2d(x, sd["emb_block1.downsample.0.weight"], sd.get("emb_block1.downsample.0.bias"))
    idt = batch_norm(idt, sd, "emb_block1.downsample.1", training, 1e-5)
    idt = F.max_pool2d(idt, 3, 2, 1) if pool_sel is None else max_pool_3s2_pinned(idt, pool_sel, stats=_dep(pins))
    return relu_site(o + idt, "emb_block1.add", pins)


def emb_block2(x: Tensor, sd, training: bool, pins=None) -> Tensor:
    """EmbedderRes18Layer2.forward (module_exp.py:100-111)."""
    o = sfconv2d(x, sd, "emb_block2.conv1", 1, None)
    o = relu_site(batch_norm(o, sd, "emb_block2.norm1", training, 1e-5), "emb_block2.norm1", pins)
    o = F.conv2d(o, sd["emb_block2.conv2.weight"], sd.get("emb_block2.conv2.bias"), 1, 1)
    o = batch_norm(o, sd, "emb_block2.norm2", training, 1e-5)
    return relu_site(o + x, "emb_block2.add", pins)


def _dec(x, sd, prefix, idx_conv, transposed=False, pins=None):
    w, b = sd[f"{prefix}.{idx_conv}.weight"], sd.get(f"{prefix}.{idx_conv}.bias")
    x = F.conv_transpose2d(x, w, b, 2, 1, 1) if transposed else F.conv2d(x, w, b, 1, 1)
    return relu_site(instance_norm(x, sd.get(f"{prefix}.{idx_conv + 1}.weight"), sd.get(f"{prefix}.{idx_conv + 1}.bias")),
                     f"{prefix}.{idx_conv + 1}", pins)


def forward_r18(sd: Dict[str, Tensor], x: Tensor, training: bool = False, drop_rate: float = 0.2,
                freq_norm: Optional[str] = "ortho", rng: Optional[dict] = None) -> dict:
    """UniDefenseModelRes18.forward without input perturbation (model/unidefense.py:363-436).
    rng (training): 'dec_keep' like ext_feat (F.dropout p=0.2, :391), 'emb_keep' like emb_feat (:359),
    'feat_keep' [N,512] (:406; NOT in place here, so 'factorization' is the un-dropped feature)."""
    rng = rng or {}
    # rng['noise_x']: perturbed encoder input of the second pass (model/unidefense.py:372-392); the clean x
    # stays the target of the attention residuals and of the reconstruction losses
    pins = rng.get("relu_masks")        # optional {site: bool pattern} recorded from the implementation under test
    _, ext = extractor(rng.get("noise_x", x), sd, training, freq_norm, pins)
    d_in = ext
    if training and rng.get("dec_keep") is not None:
        d_in = ext * rng["dec_keep"].to(x.dtype) / 0.8
    d = _dec(d_in, sd, "dec_block1", 0, pins=pins)
    d = _dec(d, sd, "dec_block1", 3, transposed=True, pins=pins)
    dec1 = _dec(d, sd, "dec_block1", 6, pins=pins)
    d = _dec(dec1, sd, "dec_block2", 0, pins=pins)
    d = _dec(d, sd, "dec_block2", 3, transposed=True, pins=pins)
    d = _dec(d, sd, "dec_block2", 6, pins=pins)
    dec2 = torch.tanh(F.conv2d(d, sd["dec_block2.9.weight"], sd.get("dec_block2.9.bias"), 1, 1))

    emb = emb_block1(ext, sd, training, rng.get("pool_sel"), pins)
    # attention (model/unidefense.py:326-361): ReLU filters, att_depth 512
    size = emb.shape[-2:]
    pred = interpolate(dec2.detach(), size)
    xs = interpolate(x, size)
    freq_diff = torch.abs(rfft2_cat(pred, freq_norm) - rfft2_cat(xs, freq_norm))
    emb_freq = rfft2_cat(emb, freq_norm)
    ff = dynamic_filter_generic(emb_freq, freq_diff, sd, "freq_filter", training, 0,
                                lambda t: relu_site(t, "freq_filter.layer1.1", pins))
    freq_filtered = irfft2_split(ff["out"], size, freq_norm)
    sf = dynamic_filter_generic(emb, torch.abs(pred - xs), sd, "spat_filter", training, 1,
                                lambda t: relu_site(t, "spat_filter.layer1.1", pins))
    a = torch.sigmoid(sd["fuse_coef"])
    att = (1.0 - a) * sf["out"] + a * freq_filtered
    e = emb
    if training and rng.get("emb_keep") is not None:
        e = emb * rng["emb_keep"].to(x.dtype) / (1.0 - drop_rate)
    att = att + e

    h = emb_block2(att, sd, training, pins)
    h = h.mean((2, 3))
    fac = batch_norm(h, sd, "bottleneck", training, 1e-5)
    h = fac
    if training and rng.get("feat_keep") is not None:
        h = fac * rng["feat_keep"].to(x.dtype) / (1.0 - drop_rate)
    cls_out = F.linear(h, sd["classifier.fc.weight"], sd["classifier.fc.bias"])
    loss_dict = {"factorization": fac, "triplet": [ext.mean((2, 3)), dec1.mean((2, 3))],
                 "freq_mask": ff["mask"], "spat_mask": sf["mask"]}
    rec = interpolate(dec2, x.shape[-2:])
    loss_dict["spatial"] = torch.abs(rec - x).mean((1, 2, 3))
    tmp = torch.abs(rfft2_cat(rec, freq_norm) - rfft2_cat(x, freq_norm))
    t_re, t_im = tmp.tensor_split(2, dim=1)
    loss_dict["freq"] = (t_re + t_im).mean((1, 2, 3))

    def top2_gap(p):
        t = p.detach().topk(2, dim=1).values
        return ((t[:, 0] - t[:, 1]) / t[:, 0].abs().clamp_min(1e-30)).min()
    return {"cls_out": cls_out, "rec": rec, "loss_dict": loss_dict,
            "_max_gap": torch.minimum(top2_gap(ff["proj"]), top2_gap(sf["proj"])),
            "_feats": {"ext": ext, "emb": emb, "dec1": dec1, "dec2": dec2, "att_out": att}}


def r18_state_shapes(num_classes: int = 2, mid_depth: int = 448, bias: bool = False, affine: bool = True) -> Dict[str, tuple]:
    """bias / affine: the constructor variants of model/unidefense.py:268-270 (embedder, decoder and filter convs; their norms)"""
    sh: Dict[str, tuple] = {}

    def bn(p, c):
        sh[p + ".weight"] = (c,); sh[p + ".bias"] = (c,)
        sh[p + ".running_mean"] = (c,); sh[p + ".running_var"] = (c,); sh[p + ".num_batches_tracked"] = ()

    def sf(p, c):
        sh[p + ".weight"] = (c, c, 3, 3); sh[p + ".sf_coef"] = (); sh[p + ".freq_conv.weight"] = (2 * c, 2 * c, 1, 1)

    sh["fuse_coef"] = ()
    sh["extractor.conv1.weight"] = (64, 3, 7, 7)
    bn("extractor.bn1", 64)
    inpl = 64
    for li, planes in ((1, 64), (2, 128), (3, 256)):
        for bi in range(2):
            p = f"extractor.layer{li}.{bi}"
            cin = inpl if bi == 0 else planes
            use_sf = li > 1
            if use_sf and cin == planes:
                sf(p + ".conv1", planes)
            else:
                sh[p + ".conv1.weight"] = (planes, cin, 3, 3)
            bn(p + ".bn1", planes)
            if use_sf:
                sf(p + ".conv2", planes)
            else:
                sh[p + ".conv2.weight"] = (planes, planes, 3, 3)
            bn(p + ".bn2", planes)
            if bi == 0 and cin != planes:
                sh[p + ".downsample.0.weight"] = (planes, cin, 1, 1)
                bn(p + ".downsample.1", planes)
        inpl = planes
    def nrm(p, c):          # a norm the affine flag reaches
        bn(p, c)
        if not affine:
            del sh[p + ".weight"], sh[p + ".bias"]

    def cb(p, c):           # a conv the bias flag reaches
        if bias:
            sh[p + ".bias"] = (c,)

    sh["emb_block1.conv1.weight"] = (512, mid_depth, 3, 3); cb("emb_block1.conv1", 512)
    nrm("emb_block1.norm1", 512)
    sf("emb_block1.conv2", 512); cb("emb_block1.conv2", 512)
    nrm("emb_block1.norm2", 512)
    sh["emb_block1.downsample.0.weight"] = (512, mid_depth, 1, 1); cb("emb_block1.downsample.0", 512)
    nrm("emb_block1.downsample.1", 512)
    sf("emb_block2.conv1", 512); cb("emb_block2.conv1", 512)
    nrm("emb_block2.norm1", 512)
    sh["emb_block2.conv2.weight"] = (512, 512, 3, 3); cb("emb_block2.conv2", 512)
    nrm("emb_block2.norm2", 512)

    def dec(prefix, specs):
        for idx, (co, ci) in specs:
            sh[f"{prefix}.{idx}.weight"] = (co, ci, 3, 3)
            cb(f"{prefix}.{idx}", co)
            if idx != 9 and affine:
                sh[f"{prefix}.{idx + 1}.weight"] = (co,)
                sh[f"{prefix}.{idx + 1}.bias"] = (co,)

    dec("dec_block1", [(0, (128, mid_depth)), (3, (128, 128)), (6, (128, 128))])
    dec("dec_block2", [(0, (64, 128)), (3, (64, 64)), (6, (32, 64)), (9, (3, 32))])
    bn("bottleneck", 512)
    sh["classifier.fc.weight"] = (num_classes, 512)
    sh["classifier.fc.bias"] = (num_classes,)
    sh["freq_filter.layer1.0.weight"] = (1024, 1024, 1, 1); cb("freq_filter.layer1.0", 1024)
    nrm("freq_filter.layer1.1", 1024)
    sh["freq_filter.layer2.0.weight"] = (1, 8, 1, 1); cb("freq_filter.layer2.0", 1)
    sh["spat_filter.layer1.0.weight"] = (512, 512, 3, 3); cb("spat_filter.layer1.0", 512)
    nrm("spat_filter.layer1.1", 512)
    sh["spat_filter.layer2.0.weight"] = (1, 5, 1, 1); cb("spat_filter.layer2.0", 1)
    return sh
