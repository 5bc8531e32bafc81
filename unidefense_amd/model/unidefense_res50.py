"""UniDefenseModelRes50 on the MI355X HIP kernels.

Mirror of the reference's ``model/unidefense.py:439-631`` (+ ``model/resnet/module_exp.py:34-59,112-177`` and the
Bottleneck of ``model/resnet/exp.py:150-228``): same constructor kwargs, forward signature, return dict and the
374 state-dict keys.  The torch.nn modules are parameter containers only; compute runs through
``unidefense_amd.tape`` on the HIP kernels, activations are pixel-major [N,H,W,C].

Input sizes: the spectral branches use the in-register FFT kernels (sizes 8..64 and the mixed-radix 10/20/40/80),
i.e. 256x256 and 320x320 (BASELINE configs[3]) inputs; other sizes raise a clear error.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import kernels as K
from .. import tape as T
from .unidefense import Classifier, _FilterParams
from .unidefense_res import UniDefenseModelRes18, _SFConv2dParams


class _BottleneckParams(nn.Module):
    """model/resnet/exp.py:150-201 (cardinality 1, base width 64: width == planes)."""

    def __init__(self, inplanes, planes, stride, freq_norm, downsample):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _SFConv2dParams(planes, planes, 3, freq_norm) if freq_norm is not None else \
            nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        nn.init.zeros_(self.bn3.weight)          # zero_init_last (exp.py:200-201)


class _ExtractorRes50(nn.Module):
    """ExtractorRes50 (model/resnet/module_exp.py:34-47): stem, max-pool, layer1..3 of the custom resnet50;
    SFConv only in stages >= 2 (exp.py:303)."""

    def __init__(self, freq_norm):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, planes, nblk, stride in ((1, 64, 3, 1), (2, 128, 4, 2), (3, 256, 6, 2)):
            blocks = []
            for bi in range(nblk):
                blocks.append(_BottleneckParams(inpl if bi == 0 else planes * 4, planes, stride if bi == 0 else 1,
                                                freq_norm if li > 1 else None, bi == 0))
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
            inpl = planes * 4
        for m in self.modules():                # ResNet.init_weights (exp.py:451-461)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _Emb50Params(nn.Module):
    """EmbedderRes50Layer1 / Layer2 (module_exp.py:112-129, 151-161)."""

    def __init__(self, in_depth, downsample, bias=False, affine=True):
        super().__init__()
        self.conv1 = nn.Conv2d(in_depth, 512, 1, bias=bias)
        self.norm1 = nn.BatchNorm2d(512, affine=affine)
        self.conv2 = _SFConv2dParams(512, 512, 3, None, bias=bias)
        self.norm2 = nn.BatchNorm2d(512, affine=affine)
        self.conv3 = nn.Conv2d(512, 2048, 1, bias=bias)
        self.norm3 = nn.BatchNorm2d(2048, affine=affine)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(in_depth, 2048, 1, bias=bias), nn.BatchNorm2d(2048, affine=affine),
                                            nn.Identity())


def _dec(cin, c1, c2, last=None, bias=False, affine=True):
    mods = [nn.Conv2d(cin, c1, 3, 1, 1, bias=bias), nn.InstanceNorm2d(c1, affine=affine), nn.Identity(),
            nn.ConvTranspose2d(c1, c1, 3, 2, 1, output_padding=1, bias=bias), nn.InstanceNorm2d(c1, affine=affine),
            nn.Identity(),
            nn.Conv2d(c1, c2, 3, 1, 1, bias=bias), nn.InstanceNorm2d(c2, affine=affine), nn.Identity()]
    if last is not None:
        mods += [nn.Conv2d(c2, last, 3, 1, 1, bias=bias), nn.Identity()]
    return nn.Sequential(*mods)


class UniDefenseModelRes50(UniDefenseModelRes18):
    """UniDefense model with ResNet50 backbone (reference: model/unidefense.py:439-631)."""

    def __init__(self,
                 extractor="resnet50",
                 extractor_weights: Optional[str] = None,
                 mid_depth=1024,
                 bias: bool = False,
                 drop_rate: float = 0.2,
                 affine: bool = True,
                 num_classes: int = 2,
                 freq_norm: str = 'ortho',
                 **kwargs):
        nn.Module.__init__(self)
        if extractor != "resnet50":
            raise NotImplementedError("HIP path implements the reference's UDR50 config: resnet50")
        self.freq_norm = freq_norm
        self.drop_rate = drop_rate
        self.extractor = _ExtractorRes50(freq_norm)
        self.emb_block1 = _Emb50Params(mid_depth, True, bias, affine)
        self.emb_block2 = _Emb50Params(2048, False, bias, affine)
        self.dec_block1 = _dec(mid_depth, 256, 256, bias=bias, affine=affine)
        self.dec_block2 = _dec(256, 128, 128, bias=bias, affine=affine)
        self.dec_block3 = _dec(128, 64, 32, last=3, bias=bias, affine=affine)
        self.bottleneck = nn.BatchNorm1d(2048)
        self.bottleneck.bias.requires_grad_(False)
        nn.init.constant_(self.bottleneck.weight, 1.0)
        nn.init.constant_(self.bottleneck.bias, 0.0)
        self.classifier = Classifier(depth=2048, num_classes=num_classes)
        self.freq_filter = _FilterParams(2048 * 2, 1, 8, affine, bias)
        self.spat_filter = _FilterParams(2048, 3, 5, affine, bias)
        self.fuse_coef = nn.Parameter(torch.tensor(0.))
        if extractor_weights is not None:
            sd = torch.load(extractor_weights, map_location="cpu")
            ret = self.extractor.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("layer4", "fc"))},
                                                 strict=False)
            bad = [k for k in ret.missing_keys if "sf_coef" not in k and "freq_conv" not in k]
            if bad:
                raise RuntimeError(f"pretrained weights mismatch: missing {bad}")

    # ---------------------------------------------------------------------------------------
    def _bottleneck(self, tape, x, blk):
        """Bottleneck.forward (model/resnet/exp.py:203-228)."""
        y = T.conv1x1(tape, x, blk.conv1.weight)
        y = self._bn(tape, y, blk.bn1, 2)
        y = self._conv(tape, y, blk.conv2, blk.stride)
        y = self._bn(tape, y, blk.bn2, 2)
        y = T.conv1x1(tape, y, blk.conv3.weight)
        y = self._bn(tape, y, blk.bn3, 0)
        sc = x
        if blk.downsample is not None:
            sc = T.conv_dense_any(tape, x, blk.downsample[0].weight, blk.stride, 0) if blk.stride != 1 else \
                T.conv1x1(tape, x, blk.downsample[0].weight)
            sc = self._bn(tape, sc, blk.downsample[1], 0)
        return T.add_relu(tape, y, sc, site=self._block_name(blk) + ".add")

    def _embedder(self, tape, x, e, stride, name):
        o = T.bias_add(tape, T.conv1x1(tape, x, e.conv1.weight), e.conv1.bias)
        o = self._bn(tape, o, e.norm1, 2)
        o = self._conv(tape, o, e.conv2, stride)
        o = self._bn(tape, o, e.norm2, 2)
        o = T.bias_add(tape, T.conv1x1(tape, o, e.conv3.weight), e.conv3.bias)
        o = self._bn(tape, o, e.norm3, 0)
        sel = None
        if stride == 2:                      # EmbedderRes50Layer1: 1x1 conv + BN + 3x3/2 max-pool on the identity
            idt = T.bias_add(tape, T.conv1x1(tape, x, e.downsample[0].weight), e.downsample[0].bias)
            idt = self._bn(tape, idt, e.downsample[1], 0)
            idt, sel = T.maxpool3s2(tape, idt, return_arg=True)
        else:
            idt = x
        return T.add_relu(tape, o, idt, site=name + ".add"), sel

    def _run(self, x, tape, rng, noise_x=None):
        """The whole forward (model/unidefense.py:556-631) on HIP kernels.  x: [N,3,H,W] planes; noise_x: the
        perturbed encoder input (the clean x stays the target of the attention residuals and the losses)."""
        N, _, H, W = x.shape
        if H != W or H % 32:
            raise NotImplementedError(f"UDR50 on the HIP path takes square inputs whose side is a multiple of 32 (256: FFT sizes "
                                      f"64/32/16/8 and 320: 80/40/20/10 in registers; other sides through DFT matrices); got {H}x{W}")
        rng = self._prepare_rng(rng)
        ex = self.extractor
        x_pix = K.planes_to_pix(x if noise_x is None else noise_x)
        h = T.conv_dense_any(tape, x_pix, ex.conv1.weight, 2, 3, need_dx=False)
        h = self._bn(tape, h, ex.bn1, 2)
        h, sel_stem = T.maxpool3s2(tape, h, return_arg=True)
        for layer in (ex.layer1, ex.layer2, ex.layer3):
            for blk in layer:
                h = self._bottleneck(tape, h, blk)
        ext = h                                                          # [N, H/16, W/16, 1024]

        d_in = ext
        if self.training and getattr(self, "_dec_dropout", True):        # F.dropout(ext_feat, 0.2), :586 (tests may switch it off)
            d_in = T.dropout_mask(tape, ext, self._keep_mask(rng, "dec_keep", ext, 0.8), 0.2)
        d = self._dec(tape, d_in, self.dec_block1, 0)
        d = self._dec(tape, d, self.dec_block1, 3, transposed=True)
        dec1 = self._dec(tape, d, self.dec_block1, 6)
        d = self._dec(tape, dec1, self.dec_block2, 0)
        d = self._dec(tape, d, self.dec_block2, 3, transposed=True)
        d = self._dec(tape, d, self.dec_block2, 6)
        d = self._dec(tape, d, self.dec_block3, 0)
        d = self._dec(tape, d, self.dec_block3, 3, transposed=True)
        d = self._dec(tape, d, self.dec_block3, 6)
        d = T.bias_add(tape, T.conv_dense_any(tape, d, self.dec_block3[9].weight, 1, 1), self.dec_block3[9].bias)
        dec3 = T.tanh_to_planes(tape, d)                                 # [N,3,H/2,W/2]

        emb, sel_emb = self._embedder(tape, ext, self.emb_block1, 2, "emb_block1")

        # attention (model/unidefense.py:521-554) with ReLU filters
        n_, hh, ww, Cc = emb.shape
        norm = self.freq_norm
        pred = K.planes_to_pix(K.bilinear_fwd(dec3, hh, ww))
        xs = K.planes_to_pix(K.bilinear_fwd(x, hh, ww))
        sf, _ = T._fft_scales(hh, norm)
        freq_diff = K.absdiff(K.rfft2(pred, sf), K.rfft2(xs, sf))
        emb_freq = T.rfft2_cat(tape, emb, norm)
        ff = self.freq_filter
        proj = T.bias_add(tape, T.conv1x1(tape, emb_freq, ff.layer1[0].weight), ff.layer1[0].bias)
        proj = self._bn(tape, proj, ff.layer1[1], 2)
        f_out, freq_mask = T.dynamic_filter(tape, emb_freq, proj, freq_diff, ff.layer2[0].weight, ff.layer2[0].bias)
        freq_filtered = T.irfft2_split(tape, f_out, norm)
        spat_diff = K.absdiff(pred, xs)
        sfm = self.spat_filter
        proj = T.bias_add(tape, T.conv_dense_any(tape, emb, sfm.layer1[0].weight, 1, 1), sfm.layer1[0].bias)
        proj = self._bn(tape, proj, sfm.layer1[1], 2)
        s_out, spat_mask = T.dynamic_filter(tape, emb, proj, spat_diff, sfm.layer2[0].weight, sfm.layer2[0].bias)
        att = T.gate_mix(tape, s_out, freq_filtered, self.fuse_coef)
        e = emb
        if self.training and self.drop_rate > 0:
            e = T.dropout_mask(tape, emb, self._keep_mask(rng, "emb_keep", emb, 1.0 - self.drop_rate), self.drop_rate)
        att = T.add(tape, att, e)

        hfin, _ = self._embedder(tape, att, self.emb_block2, 1, "emb_block2")
        pooled = T.mean_hw(tape, hfin)
        fac = self._bn(tape, pooled, self.bottleneck, 0)
        feat = fac
        if self.training and self.drop_rate > 0:                         # nn.Dropout (not in place, :604)
            feat = T.dropout_mask(tape, fac, self._keep_mask(rng, "feat_keep", fac, 1.0 - self.drop_rate),
                                  self.drop_rate)
        cls_out = T.linear(tape, feat, self.classifier.fc.weight, self.classifier.fc.bias)
        t0 = T.mean_hw(tape, ext)
        t1 = T.mean_hw(tape, dec1)
        rec = T.bilinear(tape, dec3, H, W)
        spatial, freq = T.rec_losses(tape, rec, x, self.freq_norm)
        return {"cls_out": cls_out, "rec": rec, "factorization": fac, "triplet0": t0, "triplet1": t1,
                "freq_mask": freq_mask, "spat_mask": spat_mask, "spatial": spatial, "freq": freq,
                "_feats": {"ext": ext, "emb": emb, "dec1": dec1, "dec3": dec3, "att_out": att,
                           "pool_sel_stem": sel_stem, "pool_sel_emb": sel_emb}}
