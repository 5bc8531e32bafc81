"""Input perturbations of the second (consistency) pass of the train step — no-grad preprocessing of the
input batch, chosen with the torch global RNG exactly like the reference (model/unidefense.py:177-198).

Status: SURVEY.md §8(f) rank 1 ("next" after the hot path): these run as device-side torch ops for now
(they are outside the differentiated path and touch only the [N,3,H,W] input); dedicated HIP kernels
(EFDM sort-match, amplitude transfer on the 256x256 FFT) are the next widening step.

Semantics restated from:
  random_noise / random_blur / downscale      model/modules.py:7-21
  FrequencyStyleTransfer                      model/modules.py:35-55
  SpatialStyleTransfer (EFDM)                 model/modules.py:58-76
  coral colour transfer                       utils/operation.py:7-45  (incl. its use of svd's Vh as V)
"""
import torch
import torch.nn.functional as F


def random_noise(t, mean=0.0, std=1e-4):
    return torch.clip(t + torch.normal(mean, std, size=t.shape, device=t.device), -1.0, 1.0)


def random_blur(t, kernel_size=5):
    """torchvision.transforms.functional.gaussian_blur(t, (5,5)) — torchvision is absent from this image, so
    this follows its documented rule: sigma = 0.3*((k-1)*0.5 - 1) + 0.8, reflect padding, separable kernel.
    (Parity for this one perturbation is UNPINNED)"""
    k = kernel_size
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    xs = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k, device=t.device, dtype=t.dtype)
    pdf = torch.exp(-0.5 * (xs / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = torch.outer(k1, k1)
    c = t.shape[1]
    w = k2.expand(c, 1, k, k).contiguous()
    p = k // 2
    return F.conv2d(F.pad(t, [p, p, p, p], mode="reflect"), w, groups=c)


def downscale(t, bottleneck_scale=0.75):
    down = F.interpolate(t, scale_factor=bottleneck_scale, mode="nearest")
    return F.interpolate(down, size=t.shape[-2:], mode="nearest")


PERT_FUNCS = [random_noise, random_blur, downscale]


def freq_style_transfer(content, style):
    b = content.shape[0]
    lmda = (torch.rand((b, 1, 1, 1)) / 2.0 + 0.5).to(content)
    fa = torch.fft.rfft2(content, dim=(-2, -1), norm="ortho")
    fb = torch.fft.rfft2(style, dim=(-2, -1), norm="ortho")
    amp = lmda * torch.abs(fa) + (1.0 - lmda) * torch.abs(fb)
    mixed = amp * torch.exp(1j * torch.angle(fa))
    return torch.fft.irfft2(mixed, s=content.shape[-2:], dim=(-2, -1), norm="ortho")


def spatial_style_transfer(content, style):
    assert content.shape == style.shape
    b, c, h, w = content.shape
    lmda = (torch.rand((b, 1, 1)) / 2.0 + 0.5).to(content)
    cv = content.reshape(b, c, -1)
    _, idx = torch.sort(cv, dim=-1)
    sv, _ = torch.sort(style.reshape(b, c, -1), dim=-1)
    inv = idx.argsort(-1)
    out = cv + (1 - lmda) * sv.gather(-1, inv) - (1 - lmda) * cv
    return out.reshape(b, c, h, w)


def _mat_sqrt(x):
    u, d, vh = torch.linalg.svd(x)
    # the reference multiplies by the transpose of svd's THIRD output, i.e. by V (utils/operation.py:15-17)
    return u @ torch.diag_embed(d.pow(0.5)) @ vh.transpose(-1, -2)


def coral(source, target):
    """Batched version of utils/operation.py:20-45 over [N,3,H,W]."""
    n = source.shape[0]

    def stats(t):
        f = t.reshape(n, 3, -1)
        mean = f.mean(-1, keepdim=True)
        std = f.std(-1, keepdim=True)
        fn = (f - mean) / std
        cov = fn @ fn.transpose(1, 2) + torch.eye(3, dtype=t.dtype, device=t.device)
        return fn, mean, std, cov

    s_n, _, _, s_cov = stats(source)
    _, t_mean, t_std, t_cov = stats(target)
    # 3x3 factorisations on the host (tiny), the image-sized products on the device
    a = _mat_sqrt(t_cov.cpu()) @ torch.linalg.inv(_mat_sqrt(s_cov.cpu()))
    out = a.to(source.device) @ s_n
    return (out * t_std + t_mean).reshape(source.shape)


@torch.no_grad()
def perturb_input(x, pert_real_list, pert_fake_list, preserve_color):
    """model/unidefense.py:177-198 (the `need augmentation` branch)."""
    if torch.rand(1) > 0.5:
        sum_real, sum_fake = len(pert_real_list), len(pert_fake_list)
        x_real = x.narrow(0, 0, sum_real)
        x_fake = x.narrow(0, sum_real, sum_fake)
        x_s = torch.cat([x_real[pert_real_list.to(x.device)], x_fake[pert_fake_list.to(x.device)]], dim=0)
        if preserve_color:
            x_s = coral(x_s, x)
        rand = torch.randint(0, 2, size=(1,))
        fn = freq_style_transfer if rand == 0 else spatial_style_transfer
        return fn(x, x_s)
    rand = torch.randint(0, len(PERT_FUNCS), size=(1,))
    return PERT_FUNCS[rand](x)
