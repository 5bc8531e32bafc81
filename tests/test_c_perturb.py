"""Pass-2 input perturbations (reference model/unidefense.py:177-198, model/modules.py:7-76,
utils/operation.py:7-45).

CPU: the oracle restatement (oracle/perturb.py) against the vectors recorded from the reference's own functions
(tests/golden/perturb_n4.npz, oracle/make_golden_perturb.py), function by function and branch by branch (the
branch cases replay the reference's global-RNG draws: torch.rand(1) > 0.5, torch.randint, lmda = rand/2 + 0.5).
GPU: the product (unidefense_amd.model.perturb, HIP kernels through the C-ABI) against the same vectors and, at the
full 256x256 bs-32 size, against the oracle.
"""
import os

import numpy as np
import pytest
import torch

from oracle import param_fill, perturb as OP


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "perturb_n4.npz"))
    n, size, seed = [int(v) for v in g["meta"]]
    x = param_fill.make_input(n, size, seed=seed)
    return g, x


def _style(g, x):
    return OP.style_batch(x, g["pert_real"], g["pert_fake"])


def _close(got, ref, tol, what):
    err = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max() / max(np.abs(ref).max(), 1e-30)
    from tests.margins import within
    assert within(what, err, tol), (what, err)
    return err


def test_oracle_functions_vs_reference_golden(golden_dir):
    g, xt = _load(golden_dir)
    x = xt.numpy().astype(np.float64)
    st = _style(g, x)
    lm = g["lmda"].astype(np.float64)
    _close(OP.freq_style_transfer(x, st, lm.reshape(-1, 1, 1, 1)), g["freq_transfer"], 2e-6, "freq")
    # rank matching only moves values around: exact up to the fp32 rounding of the reference's mix
    _close(OP.spatial_style_transfer(x, st, lm.reshape(-1, 1, 1)), g["spat_transfer"], 2e-7, "spat")
    _close(OP.coral(st, x), g["coral"], 2e-5, "coral")
    assert np.array_equal(OP.downscale(xt.numpy()), g["downscale"])
    for s in (128, 256, 320):
        assert np.array_equal(OP.downscale_index(s), g[f"downscale_index_{s}"]), s


def _replay_branch(g, x, name):
    """The draws of model/unidefense.py:177-198 from the seeded global generator, fed to the oracle."""
    seed, color = [int(v) for v in g[f"branch_{name}_seed"]]
    torch.manual_seed(seed)
    style_branch = bool(torch.rand(1) > 0.5)
    if style_branch:
        st = _style(g, x)
        if color:
            st = OP.coral(st, x)
        which = int(torch.randint(0, 2, size=(1,)))
        if which == 0:
            lm = (torch.rand((x.shape[0], 1, 1, 1)) / 2.0 + 0.5).numpy().astype(np.float64)
            return OP.freq_style_transfer(x, st, lm)
        lm = (torch.rand((x.shape[0], 1, 1)) / 2.0 + 0.5).numpy().astype(np.float64)
        return OP.spatial_style_transfer(x, st, lm)
    which = int(torch.randint(0, 3, size=(1,)))
    if which == 0:
        return OP.random_noise(x, torch.normal(0.0, 1e-4, size=x.shape).numpy())
    assert which == 2
    return OP.downscale(x)


@pytest.mark.parametrize("name,tol", [("noise", 1e-7), ("down", 0.0), ("freq", 2e-6), ("spat", 2e-7),
                                      ("freq_coral", 2e-5), ("spat_coral", 2e-5)])
def test_oracle_branches_vs_reference_golden(golden_dir, name, tol):
    g, xt = _load(golden_dir)
    _close(_replay_branch(g, xt.numpy().astype(np.float64), name), g[f"branch_{name}"], tol, name)


# ---------------------------------------------------------------------------------------------- GPU: the product

@pytest.mark.gpu
@pytest.mark.parametrize("name,tol", [("down", 0.0), ("freq", 1e-5), ("spat", 1e-6), ("freq_coral", 5e-5),
                                      ("spat_coral", 5e-5)])
def test_product_branches_vs_reference_golden(golden_dir, name, tol):
    """perturb_input on the device, with the reference's seeded draws, against the recorded perturbed batch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.model import perturb
    g, xt = _load(golden_dir)
    seed, color = [int(v) for v in g[f"branch_{name}_seed"]]
    torch.manual_seed(seed)
    out = perturb.perturb_input(xt.cuda(), torch.as_tensor(g["pert_real"]), torch.as_tensor(g["pert_fake"]), bool(color))
    _close(out.cpu().numpy(), g[f"branch_{name}"], tol, name)


@pytest.mark.gpu
def test_product_noise_branch_statistics(golden_dir):
    """The noise field comes from the DEVICE generator (model/modules.py:8-9 draws on tensor.device), so only its
    law can be checked: clip(x + N(0, 1e-4^2))."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.model import perturb
    x = param_fill.make_input(8, 256, seed=5).cuda()
    d = (perturb.random_noise(x) - x)
    inner = x.abs() < 0.999
    assert abs(d[inner].std().item() / 1e-4 - 1.0) < 0.02 and abs(d[inner].mean().item()) < 1e-6
    assert perturb.random_noise(x).abs().max().item() <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,size", [(32, 256), (6, 128), (4, 320)])
def test_product_full_size_vs_oracle(n, size):
    """BASELINE sizes: HIP kernels against the numpy oracle on the same draws."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.model import perturb
    x = param_fill.make_input(n, size, seed=77)
    perm = torch.randperm(n // 2, generator=torch.Generator().manual_seed(3))
    st = torch.cat([x[:n // 2][perm], x[n // 2:][perm]], 0)
    lm = torch.rand(n, generator=torch.Generator().manual_seed(4)) / 2.0 + 0.5
    xd, sd = x.cuda(), st.cuda()
    x64, s64, l64 = x.numpy().astype(np.float64), st.numpy().astype(np.float64), lm.numpy().astype(np.float64)
    assert np.array_equal(perturb.downscale(xd).cpu().numpy(), OP.downscale(x.numpy()))
    got = perturb.spatial_transfer_with(xd, sd, lm.cuda()).cpu().numpy()
    _close(got, OP.spatial_style_transfer(x64, s64, l64.reshape(-1, 1, 1)), 1e-6, "spat")
    got = perturb.freq_transfer_with(xd, sd, lm.cuda()).cpu().numpy()
    _close(got, OP.freq_style_transfer(x64, s64, l64.reshape(-1, 1, 1, 1)), 1e-5, "freq")


@pytest.mark.gpu
def test_spatial_transfer_with_ties():
    """8-bit images have many equal pixels; any assignment of the equal-rank style values among tied content
    pixels is a valid torch.sort outcome (it is not stable).  Invariants that hold for every outcome: per channel,
    the multiset of (out - lmda*content)/(1-lmda) equals the style's multiset, and ordering is weakly preserved."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.model import perturb
    g = torch.Generator().manual_seed(9)
    x = (torch.randint(0, 256, (4, 3, 64, 64), generator=g).float() / 127.5 - 1.0)
    st = (torch.randint(0, 256, (4, 3, 64, 64), generator=g).float() / 127.5 - 1.0)
    lm = torch.tensor([0.5, 0.6, 0.75, 0.9])
    out = perturb.spatial_transfer_with(x.cuda(), st.cuda(), lm.cuda()).cpu().double()
    l_ = lm.double().reshape(-1, 1, 1, 1)
    matched = ((out - l_ * x.double()) / (1 - l_)).reshape(4, 3, -1)
    assert torch.allclose(matched.sort(-1).values, st.double().reshape(4, 3, -1).sort(-1).values, atol=1e-5)
    xv = x.reshape(4, 3, -1)
    order = torch.sort(xv, dim=-1, stable=True).indices
    assert (matched.gather(-1, order).diff(dim=-1) >= -1e-5).all()


def test_blur_oracle_forms_agree_and_match_scipy():
    """random_blur's third-party arithmetic (torchvision gaussian_blur, absent here): the float64 separable restatement,
    the restatement in torchvision's own operation order and scipy.ndimage's truncated Gaussian with mirror boundaries are
    the same operator."""
    import scipy.ndimage as ndi
    x = param_fill.make_input(2, 40, seed=3)
    a = OP.gaussian_blur5(x.numpy().astype(np.float64))
    b = OP.gaussian_blur5_tv(x.double()).numpy()
    c32 = OP.gaussian_blur5_tv(x).numpy()
    assert np.abs(a - b).max() <= 1e-7 and np.abs(a - c32).max() <= 2e-6      # b: fp32 linspace / exp in the kernel
    k1 = np.exp(-0.5 * (np.arange(-2, 3) / 1.1) ** 2)
    k1 /= k1.sum()
    s_ = ndi.correlate1d(ndi.correlate1d(x.numpy().astype(np.float64), k1, axis=-1, mode="mirror"), k1, axis=-2, mode="mirror")
    assert np.abs(a - s_).max() <= 1e-12
    # constants are kept; the reflect border does not include the edge pixel twice
    assert np.abs(OP.gaussian_blur5(np.full((1, 1, 9, 9), 0.7)) - 0.7).max() <= 1e-15
    ramp = np.tile(np.arange(9.0), (9, 1))[None, None]
    assert abs(OP.gaussian_blur5(ramp)[0, 0, 4, 0] - 2 * (k1[3] * 1 + k1[4] * 2)) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n,size", [(32, 256), (6, 128), (4, 320), (2, 37)])
def test_product_blur_vs_oracle(n, size):
    """perturb.random_blur (csrc/perturb.hip blur5_reflect) against the oracle at the BASELINE sizes and a ragged one."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidefense_amd.model import perturb
    x = param_fill.make_input(n, size, seed=78)
    got = perturb.random_blur(x.cuda()).cpu().numpy()
    _close(got, OP.gaussian_blur5(x.numpy().astype(np.float64)), 1e-6, f"blur {size}")
    _close(got, OP.gaussian_blur5_tv(x).numpy(), 2e-6, f"blur {size} vs torchvision operation order (fp32)")
