"""UniDefenseModelRes18 on the MI355X HIP kernels.

Mirror of the reference's ``model/unidefense.py:259-436`` (+ ``model/resnet/module_exp.py:8-32,62-111`` and the
timm-style ResNet fork ``model/resnet/exp.py``): same constructor kwargs, forward signature, return dict and the
177 state-dict keys.  As in unidefense.py the torch.nn modules are parameter containers only; compute runs
through ``unidefense_amd.tape`` on the HIP kernels, activations are pixel-major [N,H,W,C].
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import kernels as K
from .. import tape as T
from .unidefense import Classifier, UniDefenseModelEb4, _FilterParams


class _SFConv2dParams(nn.Conv2d):
    """model/resnet/exp.py:21-34 (SFConv2d: dense conv weight + freq_conv + sf_coef)."""

    def __init__(self, cin, cout, k=3, freq_norm=None, bias=False):
        super().__init__(cin, cout, k, padding=1, bias=bias)
        self.freq_norm = freq_norm
        self.freq_conv = nn.Conv2d(cin * 2, cout * 2, kernel_size=1, bias=False)
        self.sf_coef = nn.Parameter(torch.tensor(-10.0))


def _conv3(cin, cout, freq_norm, sf):
    return _SFConv2dParams(cin, cout, 3, freq_norm) if sf else nn.Conv2d(cin, cout, 3, padding=1, bias=False)


class _BasicBlockParams(nn.Module):
    """model/resnet/exp.py:82-121."""

    def __init__(self, inplanes, planes, stride, freq_norm):
        super().__init__()
        self.stride = stride
        self.conv1 = _conv3(inplanes, planes, freq_norm, freq_norm is not None and inplanes == planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3(planes, planes, freq_norm, freq_norm is not None)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes))
        nn.init.zeros_(self.bn2.weight)          # zero_init_last (exp.py:124-125, 451-461)


class _ExtractorRes18(nn.Module):
    """ExtractorRes18 (model/resnet/module_exp.py:8-20): stem + layer1..3 of the custom resnet18."""

    def __init__(self, freq_norm):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(_BasicBlockParams(64, 64, 1, None), _BasicBlockParams(64, 64, 1, None))
        self.layer2 = nn.Sequential(_BasicBlockParams(64, 128, 2, freq_norm), _BasicBlockParams(128, 128, 1, freq_norm))
        self.layer3 = nn.Sequential(_BasicBlockParams(128, 256, 2, freq_norm), _BasicBlockParams(256, 256, 1, freq_norm))
        for m in self.modules():                # ResNet.init_weights (exp.py:451-461)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _Emb1Params(nn.Module):
    """EmbedderRes18Layer1 (module_exp.py:62-75)."""

    def __init__(self, in_depth, bias=False, affine=True):
        super().__init__()
        self.conv1 = nn.Conv2d(in_depth, 512, 3, 2, padding=1, bias=bias)
        self.norm1 = nn.BatchNorm2d(512, affine=affine)
        self.conv2 = _SFConv2dParams(512, 512, 3, None, bias=bias)
        self.norm2 = nn.BatchNorm2d(512, affine=affine)
        self.downsample = nn.Sequential(nn.Conv2d(in_depth, 512, 1, bias=bias), nn.BatchNorm2d(512, affine=affine), nn.Identity())


class _Emb2Params(nn.Module):
    """EmbedderRes18Layer2 (module_exp.py:92-98)."""

    def __init__(self, bias=False, affine=True):
        super().__init__()
        self.conv1 = _SFConv2dParams(512, 512, 3, None, bias=bias)
        self.norm1 = nn.BatchNorm2d(512, affine=affine)
        self.conv2 = nn.Conv2d(512, 512, 3, 1, padding=1, bias=bias)
        self.norm2 = nn.BatchNorm2d(512, affine=affine)


def _dec1(cin, bias=False, affine=True):
    return nn.Sequential(nn.Conv2d(cin, 128, 3, 1, 1, bias=bias), nn.InstanceNorm2d(128, affine=affine), nn.Identity(),
                         nn.ConvTranspose2d(128, 128, 3, 2, 1, output_padding=1, bias=bias),
                         nn.InstanceNorm2d(128, affine=affine), nn.Identity(),
                         nn.Conv2d(128, 128, 3, 1, 1, bias=bias), nn.InstanceNorm2d(128, affine=affine), nn.Identity())


def _dec2(bias=False, affine=True):
    return nn.Sequential(nn.Conv2d(128, 64, 3, 1, 1, bias=bias), nn.InstanceNorm2d(64, affine=affine), nn.Identity(),
                         nn.ConvTranspose2d(64, 64, 3, 2, 1, output_padding=1, bias=bias),
                         nn.InstanceNorm2d(64, affine=affine), nn.Identity(),
                         nn.Conv2d(64, 32, 3, 1, 1, bias=bias), nn.InstanceNorm2d(32, affine=affine), nn.Identity(),
                         nn.Conv2d(32, 3, 3, 1, 1, bias=bias), nn.Identity())


class UniDefenseModelRes18(nn.Module):
    """UniDefense model with ResNet18 backbone (reference: model/unidefense.py:259-436)."""

    path = "model/unidefense.py"
    _out_keys = ("cls_out", "rec", "factorization", "triplet0", "triplet1", "freq_mask", "spat_mask", "spatial", "freq")
    _triplet_keys = ("triplet0", "triplet1")
    # the generic forward (perturbation, autograd node, return dict) and helpers are shared with the Eb4 model
    forward = UniDefenseModelEb4.forward
    _bn = UniDefenseModelEb4._bn
    _sync_group = UniDefenseModelEb4._sync_group
    _keep_mask = UniDefenseModelEb4._keep_mask
    _to_pix_mask = staticmethod(UniDefenseModelEb4._to_pix_mask)

    def __init__(self,
                 extractor="resnet18",
                 extractor_weights: Optional[str] = None,
                 mid_depth=448,
                 bias: bool = False,
                 drop_rate: float = 0.2,
                 affine: bool = True,
                 num_classes: int = 2,
                 freq_norm: str = 'ortho',
                 **kwargs):
        super().__init__()
        if extractor != "resnet18":
            raise NotImplementedError("HIP path implements the reference's UDR18 config: resnet18")
        self.freq_norm = freq_norm
        self.drop_rate = drop_rate
        self.extractor = _ExtractorRes18(freq_norm)
        self.emb_block1 = _Emb1Params(mid_depth, bias, affine)
        self.emb_block2 = _Emb2Params(bias, affine)
        self.dec_block1 = _dec1(mid_depth, bias, affine)
        self.dec_block2 = _dec2(bias, affine)
        self.bottleneck = nn.BatchNorm1d(512)
        self.bottleneck.bias.requires_grad_(False)
        nn.init.constant_(self.bottleneck.weight, 1.0)
        nn.init.constant_(self.bottleneck.bias, 0.0)
        self.classifier = Classifier(num_classes=num_classes)
        self.freq_filter = _FilterParams(512 * 2, 1, 8, affine, bias)
        self.spat_filter = _FilterParams(512, 3, 5, affine, bias)
        self.fuse_coef = nn.Parameter(torch.tensor(0.))
        if extractor_weights is not None:
            sd = torch.load(extractor_weights, map_location="cpu")
            ret = self.extractor.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("layer4", "fc"))},
                                                 strict=False)
            bad = [k for k in ret.missing_keys if "sf_coef" not in k and "freq_conv" not in k]
            if bad:
                raise RuntimeError(f"pretrained weights mismatch: missing {bad}")

    # ---------------------------------------------------------------------------------------
    def _conv(self, tape, x, conv, stride):
        if isinstance(conv, _SFConv2dParams):
            return T.sfconv_dense(tape, x, conv.weight, conv.freq_conv.weight, conv.sf_coef, stride, conv.freq_norm, conv.bias)
        return T.bias_add(tape, T.conv_dense_any(tape, x, conv.weight, stride, conv.padding[0]), conv.bias)

    def _basic_block(self, tape, x, blk):
        """BasicBlock.forward (model/resnet/exp.py:127-149)."""
        y = self._conv(tape, x, blk.conv1, blk.stride)
        y = self._bn(tape, y, blk.bn1, 2)
        y = self._conv(tape, y, blk.conv2, 1)
        y = self._bn(tape, y, blk.bn2, 0)
        sc = x
        if blk.downsample is not None:
            sc = T.conv_dense_any(tape, x, blk.downsample[0].weight, blk.stride, 0)
            sc = self._bn(tape, sc, blk.downsample[1], 0)
        return T.add_relu(tape, y, sc, site=self._block_name(blk) + ".add")

    def _block_name(self, mod):
        names = getattr(self, "_mod_names", None)
        if names is None:
            names = self._mod_names = {id(m): n for n, m in self.named_modules()}
        return names[id(mod)]

    def _dec(self, tape, x, dec, idx, transposed=False):
        x = T.conv_transpose_s2(tape, x, dec[idx].weight) if transposed else \
            T.conv_dense_any(tape, x, dec[idx].weight, 1, 1)
        x = T.bias_add(tape, x, dec[idx].bias)
        return T.instancenorm_act(tape, x, dec[idx + 1].weight, dec[idx + 1].bias, dec[idx + 1].eps, 2)

    def _prepare_rng(self, rng):
        return {"drop_connect": {}, "_given": rng or {}}

    def _run(self, x, tape, rng, noise_x=None):
        """The whole forward (model/unidefense.py:389-436) on HIP kernels.  x: [N,3,H,W] planes; noise_x: the
        perturbed encoder input (the clean x stays the target of the attention residuals and the losses)."""
        N, _, H, W = x.shape
        rng = self._prepare_rng(rng)
        ex = self.extractor
        x_pix = K.planes_to_pix(x if noise_x is None else noise_x)
        h = T.conv_dense_any(tape, x_pix, ex.conv1.weight, 2, 3, need_dx=False)
        h = self._bn(tape, h, ex.bn1, 2)
        p1 = self._basic_block(tape, self._basic_block(tape, h, ex.layer1[0]), ex.layer1[1])
        p2 = self._basic_block(tape, self._basic_block(tape, p1, ex.layer2[0]), ex.layer2[1])
        p3 = self._basic_block(tape, self._basic_block(tape, p2, ex.layer3[0]), ex.layer3[1])
        ext = T.concat_channels(tape, [T.avgpool(tape, p1, p1.shape[1] // p3.shape[1]),
                                       T.avgpool(tape, p2, p2.shape[1] // p3.shape[1]), p3])      # [N,h,w,448]

        d_in = ext
        if self.training and getattr(self, "_dec_dropout", True):        # F.dropout(ext_feat, 0.2), :391 (tests may switch it off)
            d_in = T.dropout_mask(tape, ext, self._keep_mask(rng, "dec_keep", ext, 0.8), 0.2)
        d = self._dec(tape, d_in, self.dec_block1, 0)
        d = self._dec(tape, d, self.dec_block1, 3, transposed=True)
        dec1 = self._dec(tape, d, self.dec_block1, 6)
        d = self._dec(tape, dec1, self.dec_block2, 0)
        d = self._dec(tape, d, self.dec_block2, 3, transposed=True)
        d = self._dec(tape, d, self.dec_block2, 6)
        d = T.bias_add(tape, T.conv_dense_any(tape, d, self.dec_block2[9].weight, 1, 1), self.dec_block2[9].bias)
        dec2 = T.tanh_to_planes(tape, d)                                 # [N,3,H/2,W/2]

        # EmbedderRes18Layer1 (module_exp.py:77-89)
        e1 = self.emb_block1
        o = T.bias_add(tape, T.conv_dense_any(tape, ext, e1.conv1.weight, 2, 1), e1.conv1.bias)
        o = self._bn(tape, o, e1.norm1, 2)
        o = self._conv(tape, o, e1.conv2, 1)
        o = self._bn(tape, o, e1.norm2, 0)
        idt = T.bias_add(tape, T.conv1x1(tape, ext, e1.downsample[0].weight), e1.downsample[0].bias)
        idt = self._bn(tape, idt, e1.downsample[1], 0)
        idt, pool_sel = T.maxpool3s2(tape, idt, return_arg=True)
        emb = T.add_relu(tape, o, idt, site="emb_block1.add")

        # attention (model/unidefense.py:326-361) with ReLU filters
        n_, hh, ww, Cc = emb.shape
        norm = self.freq_norm
        pred = K.planes_to_pix(K.bilinear_fwd(dec2, hh, ww))
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

        # EmbedderRes18Layer2 (module_exp.py:100-111)
        e2 = self.emb_block2
        o = self._conv(tape, att, e2.conv1, 1)
        o = self._bn(tape, o, e2.norm1, 2)
        o = T.bias_add(tape, T.conv_dense_any(tape, o, e2.conv2.weight, 1, 1), e2.conv2.bias)
        o = self._bn(tape, o, e2.norm2, 0)
        h = T.add_relu(tape, o, att, site="emb_block2.add")

        pooled = T.mean_hw(tape, h)
        fac = self._bn(tape, pooled, self.bottleneck, 0)
        feat = fac
        if self.training and self.drop_rate > 0:                         # nn.Dropout (NOT in place here, :406)
            feat = T.dropout_mask(tape, fac, self._keep_mask(rng, "feat_keep", fac, 1.0 - self.drop_rate),
                                  self.drop_rate)
        cls_out = T.linear(tape, feat, self.classifier.fc.weight, self.classifier.fc.bias)
        t0 = T.mean_hw(tape, ext)
        t1 = T.mean_hw(tape, dec1)
        rec = T.bilinear(tape, dec2, H, W)
        spatial, freq = T.rec_losses(tape, rec, x, self.freq_norm)
        return {"cls_out": cls_out, "rec": rec, "factorization": fac, "triplet0": t0, "triplet1": t1,
                "freq_mask": freq_mask, "spat_mask": spat_mask, "spatial": spatial, "freq": freq,
                "_feats": {"ext": ext, "emb": emb, "dec1": dec1, "dec2": dec2, "att_out": att, "pool_sel": pool_sel}}
