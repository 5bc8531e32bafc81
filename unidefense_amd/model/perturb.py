"""Input perturbations of the second (consistency) pass of the train step — no-grad preprocessing of the
input batch, chosen with the torch global (CPU) RNG exactly like the reference (model/unidefense.py:177-198):
same draws, same order, so a seeded run takes the reference's branches with the reference's lmda values
(tests/test_c_perturb.py replays the recorded seeds of tests/golden/perturb_n4.npz).

SURVEY.md §8(f) rank 1.  The arithmetic runs in HIP kernels (csrc/perturb.hip + the DFT-matrix GEMMs of ud_gemm)
through the C-ABI; there is no torch-op fallback.  What stays torch: the RNG draws (they DEFINE parity with the
reference), the style-partner row gather `x[perm]`, the noise field `torch.normal(.., device=x.device)` and the 3x3
SVDs of the colour transfer (host LAPACK, as in a CPU run of the reference; the reference's `_mat_sqrt` result
depends on the solver's singular-vector signs, so it is implementation-defined across devices).

Semantics restated from:
  random_noise / random_blur / downscale      model/modules.py:7-21  (pert_noise = std 1e-4, model/unidefense.py:17)
  FrequencyStyleTransfer                      model/modules.py:35-55
  SpatialStyleTransfer (EFDM)                 model/modules.py:58-76
  coral colour transfer                       utils/operation.py:7-45  (incl. its use of svd's Vh as V)
"""
import numpy as np
import torch

from .. import kernels as K

_INDEX_CACHE = {}


def _f32(t):
    return t.contiguous().to(torch.float32)


def random_noise(t, mean=0.0, std=1e-4):
    return torch.clip(t + torch.normal(mean, std, size=t.shape, device=t.device), -1.0, 1.0)


def random_blur(t, kernel_size=5):
    """torchvision.transforms.functional.gaussian_blur(t, (5,5)) (model/modules.py:15-16).  torchvision is a third-party
    dependency absent from the reference tree and from this image; its published algorithm — sigma = 0.3*((k-1)*0.5 - 1)
    + 0.8, normalised exp(-x^2 / 2 sigma^2) taps, reflect padding, one depthwise 5x5 conv = two separable passes — is what
    oracle/perturb.py restates (three independent forms, tests/test_c_perturb.py) and what this kernel is held to."""
    assert kernel_size == 5
    sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    xs = np.linspace(-2.0, 2.0, 5)
    pdf = np.exp(-0.5 * (xs / sigma) ** 2)
    k1 = (pdf / pdf.sum()).astype(np.float32)
    return K.blur5_reflect(_f32(t), k1[:3])


def _nearest_index(out_size, in_size, scale):
    """ATen's nearest source index: min(floor(dst * scale), in - 1) in float32; scale = 1/scale_factor when
    F.interpolate got a scale_factor, in/out when it got a size (UpSample.h)."""
    dst = np.arange(out_size, dtype=np.float32)
    return np.minimum(np.floor(dst * np.float32(scale)).astype(np.int64), in_size - 1)


def downscale_index(size, bottleneck_scale=0.75):
    """F.interpolate(nearest, scale_factor) followed by F.interpolate(nearest, size) composed along one axis."""
    d = int(np.floor(size * bottleneck_scale))
    return _nearest_index(d, size, 1.0 / bottleneck_scale)[_nearest_index(size, d, d / size)]


def downscale(t, bottleneck_scale=0.75):
    h, w = t.shape[-2:]
    idx = []
    for s in (h, w):
        key = (s, bottleneck_scale, t.device)
        if key not in _INDEX_CACHE:
            _INDEX_CACHE[key] = torch.from_numpy(downscale_index(s, bottleneck_scale).astype(np.int32)).to(t.device)
        idx.append(_INDEX_CACHE[key])
    return K.gather2d(_f32(t), idx[0], idx[1])


PERT_FUNCS = [random_noise, random_blur, downscale]


def freq_transfer_with(content, style, lmda):
    """lmda: [B] device fp32.  rfft2(ortho) of both, amplitude mix with the content's phase, irfft2(ortho)."""
    b, c, h, w = content.shape
    if h != w:
        raise ValueError("freq_style_transfer: square inputs only (the reference's configs are 128/256/320 square)")
    ya = K.dft_rfft2_planes(_f32(content).view(b * c, h, w), ortho=True)
    yb = K.dft_rfft2_planes(_f32(style).view(b * c, h, w), ortho=True)
    mixed = K.amp_mix(ya, yb, _f32(lmda).reshape(-1), h, c)
    return K.dft_rfft2_planes_adjoint(mixed, h, ortho=True).view(b, c, h, w)


def spatial_transfer_with(content, style, lmda):
    """lmda: [B] device fp32.  Exact feature-distribution matching per (sample, channel)."""
    assert content.shape == style.shape
    b, c, h, w = content.shape
    out = K.efdm(_f32(content).view(b * c, h * w), _f32(style).view(b * c, h * w), _f32(lmda).reshape(-1), c)
    return out.view(b, c, h, w)


def freq_style_transfer(content, style):
    lmda = torch.rand((content.shape[0], 1, 1, 1)) / 2.0 + 0.5          # larger = less perturbation
    return freq_transfer_with(content, style, lmda.to(content.device))


def spatial_style_transfer(content, style):
    lmda = torch.rand((content.shape[0], 1, 1)) / 2.0 + 0.5
    return spatial_transfer_with(content, style, lmda.to(content.device))


def _mat_sqrt(x):
    u, d, vh = torch.linalg.svd(x)
    # the reference multiplies by the transpose of svd's THIRD output, i.e. by V (utils/operation.py:15-17)
    return u @ torch.diag_embed(d.pow(0.5)) @ vh.transpose(-1, -2)


def _coral_stats(x):
    """mean [N,3], unbiased std [N,3], cov-of-normalised + I [N,3,3] from the device moments (fp64, host)."""
    n_pix = x.shape[-1] * x.shape[-2]
    m = K.coral_moments(_f32(x)).cpu()
    s1, s2 = m[:, :3], m[:, 3:]
    mean = s1 / n_pix
    raw = torch.zeros(x.shape[0], 3, 3, dtype=torch.float64)
    for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
        raw[:, i, j] = raw[:, j, i] = s2[:, k]
    central = raw - n_pix * mean[:, :, None] * mean[:, None, :]
    std = (torch.diagonal(central, dim1=1, dim2=2) / (n_pix - 1)).sqrt()
    cov = central / (std[:, :, None] * std[:, None, :]) + torch.eye(3, dtype=torch.float64)
    return mean, std, cov


def coral(source, target):
    """utils/operation.py:20-45 per sample over [N,3,H,W]: out = A (s - mean_s)/std_s * std_t + mean_t with
    A = sqrt(cov_t) inv(sqrt(cov_s)), applied as ONE affine colour map per sample."""
    s_mean, s_std, s_cov = _coral_stats(source)
    t_mean, t_std, t_cov = _coral_stats(target)
    a = (_mat_sqrt(t_cov.float()) @ torch.linalg.inv(_mat_sqrt(s_cov.float()))).double()
    lin = t_std[:, :, None] * a / s_std[:, None, :]
    off = t_mean - (lin @ s_mean[:, :, None])[:, :, 0]
    m = torch.cat([lin, off[:, :, None]], 2).float().contiguous().to(source.device)
    return K.affine3(_f32(source), m)


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
