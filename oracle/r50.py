"""TEST INFRASTRUCTURE — CPU restatement of UniDefenseModelRes50 (functional, plain torch).

Reference: model/unidefense.py:439-631 (model), model/resnet/exp.py:150-228 (Bottleneck), :273-321 (make_blocks:
SFConv only in stages >= 2, i.e. layer2/layer3), :395-440 (7x7/2 stem + 3x3/2 max-pool), model/resnet/
module_exp.py:34-59 (ExtractorRes50: stem, maxpool, layer1..3), :112-175 (EmbedderRes50Layer1/2).
State keys follow the reference's state dict.  The discrete decisions of the network (both 3x3/2 max-pools, every
ReLU pattern) can be pinned through rng['pool_sel'] = {'stem': .., 'emb': ..} and rng['relu_masks'] (see r18.py).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .eb4 import batch_norm, rfft2_cat, irfft2_split, interpolate, dynamic_filter_generic
from .r18 import relu_site, max_pool_3s2_pinned, _conv, _dec, _dep

Tensor = torch.Tensor
LAYERS = ((1, 64, 3, 1), (2, 128, 4, 2), (3, 256, 6, 2))          # (stage, planes, blocks, stride of block 0)


def _pool(z: Tensor, sel: Optional[Tensor], pins=None) -> Tensor:
    return F.max_pool2d(z, 3, 2, 1) if sel is None else max_pool_3s2_pinned(z, sel, stats=_dep(pins))


def bottleneck(x: Tensor, sd, prefix: str, stride: int, training: bool, norm, pins=None) -> Tensor:
    """Bottleneck.forward (model/resnet/exp.py:203-228): 1x1 -> 3x3 (stride; SFConv in stages >= 2) -> 1x1 (+ skip)."""
    sc = x
    y = F.conv2d(x, sd[prefix + ".conv1.weight"])
    y = relu_site(batch_norm(y, sd, prefix + ".bn1", training, 1e-5), prefix + ".bn1", pins)
    y = _conv(y, sd, prefix + ".conv2", stride, norm)
    y = relu_site(batch_norm(y, sd, prefix + ".bn2", training, 1e-5), prefix + ".bn2", pins)
    y = F.conv2d(y, sd[prefix + ".conv3.weight"])
    y = batch_norm(y, sd, prefix + ".bn3", training, 1e-5)
    if prefix + ".downsample.0.weight" in sd:
        sc = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0)
        sc = batch_norm(sc, sd, prefix + ".downsample.1", training, 1e-5)
    return relu_site(y + sc, prefix + ".add", pins)


def extractor(x: Tensor, sd, training: bool, norm, pins=None, pool_sel=None) -> Tensor:
    """ExtractorRes50.forward (model/resnet/module_exp.py:48-59)."""
    h = F.conv2d(x, sd["extractor.conv1.weight"], None, 2, 3)
    h = relu_site(batch_norm(h, sd, "extractor.bn1", training, 1e-5), "extractor.bn1", pins)
    h = _pool(h, pool_sel, pins)
    for li, planes, nblk, stride in LAYERS:
        for bi in range(nblk):
            h = bottleneck(h, sd, f"extractor.layer{li}.{bi}", stride if bi == 0 else 1, training,
                           norm if li > 1 else None, pins)
    return h


def emb_block1(x: Tensor, sd, training: bool, pool_sel=None, pins=None) -> Tensor:
    """EmbedderRes50Layer1.forward (module_exp.py:131-148); its SFConv2d has freq_norm=None and stride 2."""
    o = F.conv2d(x, sd["emb_block1.conv1.weight"], sd.get("emb_block1.conv1.bias"))
    o = relu_site(batch_norm(o, sd, "emb_block1.norm1", training, 1e-5), "emb_block1.norm1", pins)
    o = _conv(o, sd, "emb_block1.conv2", 2, None)
    o = relu_site(batch_norm(o, sd, "emb_block1.norm2", training, 1e-5), "emb_block1.norm2", pins)
    o = F.conv2d(o, sd["emb_block1.conv3.weight"], sd.get("emb_block1.conv3.bias"))
    o = batch_norm(o, sd, "emb_block1.norm3", training, 1e-5)
    idt = F.conv2d(x, sd["emb_block1.downsample.0.weight"], sd.get("emb_block1.downsample.0.bias"))
    idt = batch_norm(idt, sd, "emb_block1.downsample.1", training, 1e-5)
    idt = _pool(idt, pool_sel, pins)
    return relu_site(o + idt, "emb_block1.add", pins)


def emb_block2(x: Tensor, sd, training: bool, pins=None) -> Tensor:
    """EmbedderRes50Layer2.forward (module_exp.py:163-177)."""
    o = F.conv2d(x, sd["emb_block2.conv1.weight"], sd.get("emb_block2.conv1.bias"))
    o = relu_site(batch_norm(o, sd, "emb_block2.norm1", training, 1e-5), "emb_block2.norm1", pins)
    o = _conv(o, sd, "emb_block2.conv2", 1, None)
    o = relu_site(batch_norm(o, sd, "emb_block2.norm2", training, 1e-5), "emb_block2.norm2", pins)
    o = F.conv2d(o, sd["emb_block2.conv3.weight"], sd.get("emb_block2.conv3.bias"))
    o = batch_norm(o, sd, "emb_block2.norm3", training, 1e-5)
    return relu_site(o + x, "emb_block2.add", pins)


def forward_r50(sd: Dict[str, Tensor], x: Tensor, training: bool = False, drop_rate: float = 0.2,
                freq_norm: Optional[str] = "ortho", rng: Optional[dict] = None) -> dict:
    """UniDefenseModelRes50.forward (model/unidefense.py:556-631).  rng as in r18.forward_r18; 'pool_sel' is a dict
    {'stem': winners of the stem max-pool, 'emb': winners of emb_block1's}."""
    rng = rng or {}
    pins = rng.get("relu_masks")
    psel = rng.get("pool_sel") or {}
    ext = extractor(rng.get("noise_x", x), sd, training, freq_norm, pins, psel.get("stem"))
    d_in = ext
    if training and rng.get("dec_keep") is not None:
        d_in = ext * rng["dec_keep"].to(x.dtype) / 0.8
    d = _dec(d_in, sd, "dec_block1", 0, pins=pins)
    d = _dec(d, sd, "dec_block1", 3, transposed=True, pins=pins)
    dec1 = _dec(d, sd, "dec_block1", 6, pins=pins)
    d = _dec(dec1, sd, "dec_block2", 0, pins=pins)
    d = _dec(d, sd, "dec_block2", 3, transposed=True, pins=pins)
    dec2 = _dec(d, sd, "dec_block2", 6, pins=pins)
    d = _dec(dec2, sd, "dec_block3", 0, pins=pins)
    d = _dec(d, sd, "dec_block3", 3, transposed=True, pins=pins)
    d = _dec(d, sd, "dec_block3", 6, pins=pins)
    dec3 = torch.tanh(F.conv2d(d, sd["dec_block3.9.weight"], sd.get("dec_block3.9.bias"), 1, 1))

    emb = emb_block1(ext, sd, training, psel.get("emb"), pins)
    size = emb.shape[-2:]
    pred = interpolate(dec3.detach(), size)
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
    rec = interpolate(dec3, x.shape[-2:])
    loss_dict["spatial"] = torch.abs(rec - x).mean((1, 2, 3))
    tmp = torch.abs(rfft2_cat(rec, freq_norm) - rfft2_cat(x, freq_norm))
    t_re, t_im = tmp.tensor_split(2, dim=1)
    loss_dict["freq"] = (t_re + t_im).mean((1, 2, 3))

    def top2_gap(p):
        t = p.detach().topk(2, dim=1).values
        return ((t[:, 0] - t[:, 1]) / t[:, 0].abs().clamp_min(1e-30)).min()
    return {"cls_out": cls_out, "rec": rec, "loss_dict": loss_dict,
            "_max_gap": torch.minimum(top2_gap(ff["proj"]), top2_gap(sf["proj"])),
            "_feats": {"ext": ext, "emb": emb, "dec1": dec1, "dec3": dec3, "att_out": att}}


def r50_state_shapes(num_classes: int = 2, mid_depth: int = 1024, bias: bool = False, affine: bool = True) -> Dict[str, tuple]:
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
    for li, planes, nblk, _ in LAYERS:
        for bi in range(nblk):
            p = f"extractor.layer{li}.{bi}"
            cin = inpl if bi == 0 else planes * 4
            sh[p + ".conv1.weight"] = (planes, cin, 1, 1)
            bn(p + ".bn1", planes)
            if li > 1:
                sf(p + ".conv2", planes)
            else:
                sh[p + ".conv2.weight"] = (planes, planes, 3, 3)
            bn(p + ".bn2", planes)
            sh[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(p + ".bn3", planes * 4)
            if bi == 0:
                sh[p + ".downsample.0.weight"] = (planes * 4, cin, 1, 1)
                bn(p + ".downsample.1", planes * 4)
        inpl = planes * 4
    def nrm(p, c):          # a norm the affine flag reaches (model/unidefense.py:450,461-495)
        bn(p, c)
        if not affine:
            del sh[p + ".weight"], sh[p + ".bias"]

    def cb(p, c):           # a conv the bias flag reaches
        if bias:
            sh[p + ".bias"] = (c,)

    for blk, cin in (("emb_block1", mid_depth), ("emb_block2", 2048)):
        sh[f"{blk}.conv1.weight"] = (512, cin, 1, 1); cb(f"{blk}.conv1", 512)
        nrm(f"{blk}.norm1", 512)
        sf(f"{blk}.conv2", 512); cb(f"{blk}.conv2", 512)
        nrm(f"{blk}.norm2", 512)
        sh[f"{blk}.conv3.weight"] = (2048, 512, 1, 1); cb(f"{blk}.conv3", 2048)
        nrm(f"{blk}.norm3", 2048)
    sh["emb_block1.downsample.0.weight"] = (2048, mid_depth, 1, 1); cb("emb_block1.downsample.0", 2048)
    nrm("emb_block1.downsample.1", 2048)

    def dec(prefix, specs):
        for idx, (co, ci) in specs:
            sh[f"{prefix}.{idx}.weight"] = (ci, co, 3, 3) if idx == 3 else (co, ci, 3, 3)     # idx 3: ConvTranspose2d
            cb(f"{prefix}.{idx}", co)
            if not (prefix == "dec_block3" and idx == 9) and affine:
                sh[f"{prefix}.{idx + 1}.weight"] = (co,)
                sh[f"{prefix}.{idx + 1}.bias"] = (co,)

    dec("dec_block1", [(0, (256, mid_depth)), (3, (256, 256)), (6, (256, 256))])
    dec("dec_block2", [(0, (128, 256)), (3, (128, 128)), (6, (128, 128))])
    dec("dec_block3", [(0, (64, 128)), (3, (64, 64)), (6, (32, 64)), (9, (3, 32))])
    bn("bottleneck", 2048)
    sh["classifier.fc.weight"] = (num_classes, 2048)
    sh["classifier.fc.bias"] = (num_classes,)
    sh["freq_filter.layer1.0.weight"] = (4096, 4096, 1, 1); cb("freq_filter.layer1.0", 4096)
    nrm("freq_filter.layer1.1", 4096)
    sh["freq_filter.layer2.0.weight"] = (1, 8, 1, 1); cb("freq_filter.layer2.0", 1)
    sh["spat_filter.layer1.0.weight"] = (2048, 2048, 3, 3); cb("spat_filter.layer1.0", 2048)
    nrm("spat_filter.layer1.1", 2048)
    sh["spat_filter.layer2.0.weight"] = (1, 5, 1, 1); cb("spat_filter.layer2.0", 1)
    return sh
