"""UniDefenseModelRes50 (ResNet50 backbone, bs 4) at 256x256 and at BASELINE configs[3]'s 320x320 (feature maps
80/40/20/10: the mixed-radix 5*2^k FFT kernels).
CPU: the oracle restatement (oracle/r50.py) vs the vectors recorded from the REFERENCE in float32 and float64
(tests/golden/udr50_n4*.npz).  GPU: the HIP model vs those vectors and vs the oracle in float64."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as OL
from oracle import param_fill, r50
from tests import oracle_util as ou
from tests.margins import within

GRAD_RTOL, GRAD_ATOL = 1e-3, 2e-5
# full loss variant: the L1 reconstruction / frequency tails make the decoder gradients sums of unit-modulus terms — two fp32
# evaluations (the oracle's own included) differ by ~1e-2 of the tensor there, so the bound is led by the 5 x oracle-fp32 yardstick;
# this is its floor (tests/test_c_model_gpu.py uses the same value for UDEB4)
FULL_RTOL = 2e-2


FIXTURES = ["udr50_n4.npz", "udr50_n4_s320.npz"]      # 256x256, and BASELINE configs[3]'s 320x320 (5*2^k FFT sizes)


def make_rng_r50(n, seed, drop_rate=0.5, size=256):
    g = torch.Generator().manual_seed(seed)

    def bern(shape, keep):
        return (torch.rand(shape, generator=g) < keep).float()
    return {"dec_keep": bern((n, 1024, size // 16, size // 16), 0.8),
            "emb_keep": bern((n, 2048, size // 32, size // 32), 1.0 - drop_rate),
            "feat_keep": bern((n, 2048), 1.0 - drop_rate)}


def r50_state(dtype=torch.float32, requires_grad=False):
    sd = param_fill.fill_state_dict(r50.r50_state_shapes(2), 0.0, 0.3, dtype)
    if requires_grad:
        for k, v in sd.items():
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")) and k != "bottleneck.bias":
                v.requires_grad_(True)
    return sd


def _rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _check_outputs(out, g, prefix, tol):
    ld = out["loss_dict"]
    pairs = [("cls_out", out["cls_out"]), ("rec_pool8", F.adaptive_avg_pool2d(out["rec"].detach().cpu(), 8)),
             ("factorization", ld["factorization"][:, :64])]
    pairs += [(k, ld[k]) for k in ("freq_mask", "spat_mask", "spatial", "freq")]
    pairs += [(f"triplet{i}", t) for i, t in enumerate(ld["triplet"])]
    bad = []
    for k, v in pairs:
        e = _rel(v, g[prefix + k])
        print(f"  {k}: rel err {e:.3e}")
        if not within(prefix + k, e, tol):
            bad.append((k, e))
    assert not bad, bad


def _loss(out, tgt, lam, losses_mod=None):
    return OL.pass1_loss(out, tgt, len(tgt) // 2, len(tgt) // 2, lam)["total_loss"]


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("fixture", FIXTURES)
def test_oracle_r50_matches_reference_golden(golden_dir, fixture):
    g = np.load(os.path.join(golden_dir, fixture))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    x = param_fill.make_input(n, size, seed)
    tgt = param_fill.make_labels(n)
    rng = make_rng_r50(n, mseed, size=size)
    # float32 oracle vs float32 reference: forward outputs (the gradients of two fp32 evaluations of this 53-layer
    # ReLU network with batch-4 statistics already differ by ~2e-3 — rounding plus the near-tie flips it causes)
    sd = r50_state()
    with torch.no_grad():
        _check_outputs(r50.forward_r50(sd, x, training=False), g, "eval_", 2e-5)
        _check_outputs(r50.forward_r50(sd, x, training=True, drop_rate=0.5, rng=rng), g, "train_", 1e-3)
    # the tight pin: float64 oracle vs the reference run in float64 — outputs, losses and all 214 gradients
    with torch.no_grad():
        _check_outputs(r50.forward_r50(r50_state(torch.float64), x.double(), training=False), g, "f64_eval_", 1e-9)
    for variant, lam in (("full", ou.LAMBDAS), ("smooth", ou.SMOOTH_LAMBDAS)):
        sd = r50_state(torch.float64, requires_grad=True)
        out = r50.forward_r50(sd, x.double(), training=True, drop_rate=0.5, rng=rng)
        if variant == "full":
            _check_outputs(out, g, "f64_train_", 1e-8)
        total = _loss(out, tgt, lam)
        ref_total = float(g[f"f64_{variant}_loss_total_loss"])
        assert abs(total.item() - ref_total) <= 1e-10 * abs(ref_total)
        total.backward()
        for i, k in enumerate(str(s) for s in g["grad_names"]):
            rn = float(g[f"f64_{variant}_grad_norms"][i])
            gr = sd[k].grad
            head = gr.flatten()[:8].numpy()
            err = max(abs(gr.norm().item() - rn),
                      float(np.abs(head - g[f"f64_{variant}_grad_heads"][i][: head.size]).max()))
            assert err <= 1e-7 * rn + 1e-12, (variant, k, err, rn)


# ------------------------------------------------------------------------------------------------ GPU
def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def to_pix(t):
    return t.permute(0, 2, 3, 1).contiguous()


def to_nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


# the full variant (round 4): the L1 reconstruction / frequency tails' backward at 320 x 320 through the WHOLE model, not only
# through the kernel-level adjoint test of the 320-point transforms
CASES = [("udr50_n4.npz", "smooth"), ("udr50_n4_s320.npz", "smooth"), ("udr50_n4_s320.npz", "full")]


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,variant", CASES)
def test_r50_vs_reference_golden_and_oracle(golden_dir, fixture, variant):
    dev = _dev()
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    g = np.load(os.path.join(golden_dir, fixture))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    x = param_fill.make_input(n, size, seed)
    tgt = param_fill.make_labels(n)
    rng = make_rng_r50(n, mseed, size=size)
    m = load_model("UDR50")(num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
    m = m.to(dev)
    with torch.no_grad():
        _check_outputs(m.eval()(x.to(dev)), g, "eval_", 1e-3)
    lam = ou.SMOOTH_LAMBDAS if variant == "smooth" else ou.LAMBDAS
    m.train()
    m._debug_watch = True
    out = m(x.to(dev), rng=rng)
    _check_outputs(out, g, "train_", 1e-3)
    # the two 3x3 max-pools (stem, emb_block1) have top-2 gaps down to ~1e-6 in every batch: pin the oracle's winners to
    # the HIP path's (the oracle asserts each pinned winner is a maximum within 1e-4 of max|z|) so that the gradients
    # are compared on the same branch; the float32 CPU run keeps its own arg-max and shows the effect of a flip
    sel = {"stem": m._debug_feats["pool_sel_stem"].permute(0, 3, 1, 2).cpu(),
           "emb": m._debug_feats["pool_sel_emb"].permute(0, 3, 1, 2).cpu()}
    # ... and likewise the on/off pattern of every ReLU (oracle/r50.py:relu_site checks that the pattern departs
    # from the oracle's own only on units within 1e-4 of zero): a ReLU network's gradient is piecewise constant in
    # those signs, ~10 of the ~1e7 units of this batch sit within fp32 rounding of 0, and each flip moves some
    # weight gradients by ~1e-2 — two correct fp32 evaluations that round differently land on different pieces
    masks = {k: v.permute(0, 3, 1, 2).cpu() for k, v in m._debug_kinks.items()}
    pinned = dict(rng, pool_sel=sel, relu_masks=masks)
    sd64 = r50_state(torch.float64, requires_grad=True)
    o64 = r50.forward_r50(sd64, x.double(), training=True, drop_rate=0.5, rng=pinned)
    sites = set(masks) - {"_used", "_departures"}
    assert masks["_used"] == sites, sorted(sites - masks["_used"])
    dep = masks["_departures"]
    print(f"  pinned {len(sites)} ReLU sites + the max-pool winners to the HIP path's pattern: "
          f"{dep['relu_flips']} of {dep['relu_units']} ReLU units and {dep['pool_moves']} of {dep['pool_windows']} "
          f"pool windows depart from the float64 evaluation's own choice (each checked to be a near-tie)")
    # the pinning may only ever touch a vanishing share of the decisions: a wrong-sign kernel would flip a large part
    assert dep["relu_flips"] <= 2e-5 * dep["relu_units"] + 20 and dep["pool_moves"] <= 2e-4 * dep["pool_windows"] + 20, dep
    _loss(o64, tgt, lam).backward()
    sd32 = r50_state(requires_grad=True)
    _loss(r50.forward_r50(sd32, x, training=True, drop_rate=0.5, rng=pinned), tgt, lam).backward()
    # Conditioning yardstick, independent of any implementation: how far the EXACT (float64) gradient moves, on the
    # same pinned piece, when the input is perturbed by one fp32 ulp (1e-7 relative).
    sens = {}
    for pseed in (1, 2):
        gp = torch.Generator().manual_seed(pseed)
        xp = x.double() * (1.0 + 1e-7 * torch.randn(x.shape, generator=gp, dtype=torch.float64))
        sdp = r50_state(torch.float64, requires_grad=True)
        _loss(r50.forward_r50(sdp, xp, training=True, drop_rate=0.5, rng=pinned), tgt, lam).backward()
        for k, v in sdp.items():
            if v.grad is not None:
                sens[k] = max(sens.get(k, 0.0), (v.grad - sd64[k].grad).abs().max().item())
    ld, t = out["loss_dict"], tgt.to(dev)
    trip = sum(LOSSES["aw_triplet"](f, t) for f in ld["triplet"])
    total = LOSSES["cross_entropy"](out["cls_out"], t) + lam["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) \
        + lam["lambda_triplet"] * trip
    if variant == "full":
        n_real = n // 2
        total = total + lam["lambda_recons"] * ld["spatial"].narrow(0, 0, n_real).mean() \
            + lam["lambda_freq"] * ld["freq"].narrow(0, 0, n_real).mean()
    ref_total = float(g[f"{variant}_loss_total_loss"])
    e = abs(total.item() - ref_total) / abs(ref_total)
    assert within(f"{variant} total loss", e, 1e-3), e
    total.backward()
    rows = []
    coef_scale = max(sd64[k].grad.abs().max().item() for k, p in m.named_parameters()
                     if p.grad is not None and k.endswith("sf_coef"))
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        ref = sd64[k].grad
        d = (p.grad.detach().double().cpu() - ref).abs().max().item()
        s = ref.abs().max().item()
        d32 = (sd32[k].grad.double() - ref).abs().max().item()
        # on the pinned piece the problem is smooth: 1e-4 of the tensor's scale (observed: <= 2e-5), never looser
        # than 5x what the CPU fp32 run or a one-ulp input perturbation do to the same gradient
        # (the scalar mixing coefficients are global sums with heavy cancellation: 1e-3 of their magnitude)
        rt = 1e-4 if variant == "smooth" else FULL_RTOL
        if k.endswith(("sf_coef", "fuse_coef")):
            # each is ONE sum of ~1e6 products dd * (freq - spat) that cancels to a layer-dependent degree (|ref| from
            # 0.05 to 3 in this model): the error scales with the terms, not with what is left of their sum — measure it
            # against the common scale of these gradients, not only against the tensor's own remainder
            floor = max(1e-3, rt) * max(s, 0.2 * coef_scale) + 2e-6
        else:
            floor = rt * s + 2e-6
        rows.append((d / max(floor, 5.0 * d32, 5.0 * sens.get(k, 0.0)), k, d, s, d32, sens.get(k, 0.0)))
    rows.sort(reverse=True)
    for r in rows[:10]:
        print("  %.3f  %-50s maxerr %.3e  maxref %.3e  cpu-fp32-err %.3e  ulp-sensitivity %.3e" % r)
    within("worst gradient tensor: max err / max(floor, 5 x oracle fp32 err, 5 x one-ulp sensitivity)", rows[0][0], 1.0)
    bad = [r for r in rows if not r[0] < 1.0]
    assert not bad, bad[:10]
