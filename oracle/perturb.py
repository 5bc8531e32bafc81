"""TEST INFRASTRUCTURE — CPU restatement (numpy, float64 unless told otherwise) of the pass-2 input perturbations
of the reference's train step.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path (unidefense_amd/) never does.

Pinned against tests/golden/perturb_n4.npz (oracle/make_golden_perturb.py: the reference's own functions run in
this container) by tests/test_c_perturb.py.  Random draws (lmda, the noise field, the branch choices) are INPUTS
here; the reference takes them from torch's global CPU generator.

  downscale                 model/modules.py:19-21       F.interpolate(nearest) x0.75 then back to (H, W)
  freq_style_transfer       model/modules.py:35-55       rfft2(ortho) amplitude mix, content phase, irfft2
  spatial_style_transfer    model/modules.py:58-76       exact feature-distribution matching (sort / rank gather)
  coral                     utils/operation.py:7-45      colour transfer; _mat_sqrt multiplies by svd's V (":15-17")
  random_noise              model/modules.py:7-12        clip(x + noise, -1, 1), noise handed in
  gaussian_blur5            model/modules.py:15-16 random_blur = torchvision.transforms.functional.gaussian_blur(t, (5, 5)).
                            THIRD-PARTY SOURCE ABSENT: torchvision (README.md:66 pins 0.13.1) is not vendored in the
                            reference and not installed here, so its published algorithm is restated — sigma defaults to
                            0.3*((k-1)*0.5-1)+0.8 = 0.15 k + 0.35 (= 1.1), kernel1d = exp(-0.5 (x/sigma)^2) on
                            linspace(-(k-1)/2, (k-1)/2, k) normalised to sum 1, kernel2d = outer product, reflect padding
                            by k//2, one depthwise conv2d.  gaussian_blur5 evaluates it separably in float64;
                            gaussian_blur5_tv follows torchvision's own operation order in torch (2-D kernel built in the
                            image dtype, F.pad(reflect) + F.conv2d(groups=C)).  tests/test_c_perturb.py pins the two to
                            each other and to scipy.ndimage (an independent implementation of the same truncated Gaussian
                            with mirror boundaries); the reference's call site passes no sigma, so nothing else is open.
"""
import numpy as np


def nearest_index(out_size, in_size, scale):
    """ATen's nearest source index: floor(dst * scale) clipped to in_size-1, evaluated in float32 like
    upsample_nearest (aten/native/UpSample.h nearest_neighbor_compute_source_index).  `scale` is 1/scale_factor
    when interpolate() got a scale_factor, in/out when it got a size."""
    dst = np.arange(out_size, dtype=np.float32)
    return np.minimum(np.floor(dst * np.float32(scale)).astype(np.int64), in_size - 1)


def downscale(x, bottleneck_scale=0.75):
    n, c, h, w = x.shape
    hd, wd = int(np.floor(h * bottleneck_scale)), int(np.floor(w * bottleneck_scale))
    iy, ix = nearest_index(hd, h, 1.0 / bottleneck_scale), nearest_index(wd, w, 1.0 / bottleneck_scale)
    down = x[:, :, iy][:, :, :, ix]
    jy, jx = nearest_index(h, hd, hd / h), nearest_index(w, wd, wd / w)
    return down[:, :, jy][:, :, :, jx]


def downscale_index(size, bottleneck_scale=0.75):
    """The composed source index of downscale() along one axis (what the HIP kernel gathers with)."""
    d = int(np.floor(size * bottleneck_scale))
    return nearest_index(d, size, 1.0 / bottleneck_scale)[nearest_index(size, d, d / size)]


def freq_style_transfer(content, style, lmda):
    """lmda [B,1,1,1] in [0.5, 1): larger = less perturbation."""
    fa = np.fft.rfft2(content, axes=(-2, -1), norm="ortho")
    fb = np.fft.rfft2(style, axes=(-2, -1), norm="ortho")
    amp = lmda * np.abs(fa) + (1.0 - lmda) * np.abs(fb)
    mixed = amp * np.exp(1j * np.angle(fa))
    return np.fft.irfft2(mixed, s=content.shape[-2:], axes=(-2, -1), norm="ortho")


def spatial_style_transfer(content, style, lmda):
    """lmda [B,1,1].  value_style.gather(-1, argsort(argsort(content))) = the style value of equal rank."""
    b, c, h, w = content.shape
    cv = content.reshape(b, c, -1)
    idx = np.argsort(cv, axis=-1, kind="stable")
    sv = np.sort(style.reshape(b, c, -1), axis=-1)
    inv = np.argsort(idx, axis=-1, kind="stable")
    out = cv + (1 - lmda) * np.take_along_axis(sv, inv, -1) - (1 - lmda) * cv
    return out.reshape(b, c, h, w)


def mat_sqrt(x):
    u, d, vh = np.linalg.svd(x)
    return u @ np.diag(np.sqrt(d)) @ vh.T      # the reference calls svd's third output V and transposes it


def coral_one(source, target):
    def stats(t):
        f = t.reshape(3, -1)
        mean = f.mean(-1, keepdims=True)
        std = f.std(-1, ddof=1, keepdims=True)
        fn = (f - mean) / std
        return fn, mean, std, fn @ fn.T + np.eye(3, dtype=t.dtype)

    s_n, _, _, s_cov = stats(source)
    _, t_mean, t_std, t_cov = stats(target)
    out = mat_sqrt(t_cov) @ (np.linalg.inv(mat_sqrt(s_cov)) @ s_n)
    return (out * t_std + t_mean).reshape(source.shape)


def coral(source, target):
    """model/unidefense.py:186-190: per sample, coral(style_i, content_i)."""
    return np.stack([coral_one(s, t) for s, t in zip(source, target)], 0)


def random_noise(x, noise):
    return np.clip(x + noise, -1.0, 1.0)


def gaussian_blur5(x):
    k = 5
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    xs = np.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k)
    pdf = np.exp(-0.5 * (xs / sigma) ** 2)
    k1 = pdf / pdf.sum()
    p = k // 2
    xp = np.pad(x, [(0, 0), (0, 0), (p, p), (p, p)], mode="reflect")
    h, w = x.shape[-2:]
    rows = sum(k1[i] * xp[:, :, i:i + h, :] for i in range(k))
    return sum(k1[j] * rows[:, :, :, j:j + w] for j in range(k))


def gaussian_blur5_tv(x, kernel_size=(5, 5)):
    """torchvision 0.13.1 functional_tensor.gaussian_blur in its own operation order (torch tensor in, torch tensor out)."""
    import torch
    import torch.nn.functional as F
    sigma = [k * 0.15 + 0.35 for k in kernel_size]

    def k1d(k, sg):
        half = (k - 1) * 0.5
        xs = torch.linspace(-half, half, steps=k)
        pdf = torch.exp(-0.5 * (xs / sg).pow(2))
        return pdf / pdf.sum()
    kx, ky = k1d(kernel_size[0], sigma[0]).to(x.dtype), k1d(kernel_size[1], sigma[1]).to(x.dtype)
    k2 = torch.mm(ky[:, None], kx[None, :])
    c = x.shape[-3]
    pad = [kernel_size[0] // 2, kernel_size[0] // 2, kernel_size[1] // 2, kernel_size[1] // 2]
    return F.conv2d(F.pad(x, pad, mode="reflect"), k2.expand(c, 1, *k2.shape), groups=c)


def style_batch(x, pert_real, pert_fake):
    """model/unidefense.py:179-185: the style partner of every sample — reals permuted among reals, fakes
    among fakes (batch ordered [real...; fake...])."""
    nr = len(pert_real)
    return np.concatenate([x[:nr][np.asarray(pert_real)], x[nr:nr + len(pert_fake)][np.asarray(pert_fake)]], 0)
