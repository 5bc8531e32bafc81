"""CPU: the two-pass train step restated with the oracle (oracle/eb4.py + oracle/losses.py + torch AdamW) against
the vectors recorded from the REFERENCE's engine (tests/golden/udeb4_step_n4.npz, oracle/make_golden_step.py).

This pins the step-level semantics the HIP engine mirrors: pass 2 encodes the PERTURBED input but keeps the clean
input as the target of the attention residuals and of the reconstruction losses (model/unidefense.py:200,219,
243-248), pass-1 masks / features are the (detached) targets of pass 2, gradients of pass 1 are NOT cleared before
pass 2 (one zero_grad per step, engine/forgery_engine.py:241), AdamW(amsgrad) with timm's no-decay groups.
The golden also carries the same step run by the reference in float64: |fp32 - fp64| of the reference itself is
~1e-6 on the mask means and ~6e-5 on the KL terms, so 1e-3 here is a real check, not a conditioning allowance.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eb4, losses, param_fill
from tests import oracle_util as ou


def oracle_two_pass_step(g, cur_step):
    n, size, in_seed, mask_seed, num_steps = [int(v) for v in g["meta"]]
    x = param_fill.make_input(n, size, in_seed)
    tgt = param_fill.make_labels(n)
    rngs = [ou.make_rng(n, mask_seed, 0.5), ou.make_rng(n, mask_seed + 1, 0.5)]
    sd = ou.oracle_state(0.0, 0.3, requires_grad=True)
    named = [(k, v) for k, v in sd.items() if v.requires_grad]
    before = {k: v.detach().clone() for k, v in named}
    no_decay = [p for k, p in named if p.ndim <= 1 or k.endswith(".bias")]
    decay = [p for k, p in named if not (p.ndim <= 1 or k.endswith(".bias"))]
    opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 5e-6}],
                            lr=1e-4, betas=(0.9, 0.999), amsgrad=True)
    lam = ou.LAMBDAS
    out = eb4.forward_eb4(sd, x, training=True, drop_rate=0.5, rng=rngs[0])
    l1 = losses.pass1_loss(out, tgt, n // 2, n // 2, lam)
    ld = out["loss_dict"]
    fm_gt, sm_gt, fac_gt = ld["freq_mask"].detach(), ld["spat_mask"].detach(), ld["factorization"].detach()
    l1["total_loss"].backward()
    opt.step()
    # pass 2: `downscale` perturbation (model/modules.py:19-21), the choice the golden forces
    xp = F.interpolate(F.interpolate(x, scale_factor=0.75, mode="nearest"), size=x.shape[-2:], mode="nearest")
    out2 = eb4.forward_eb4(sd, x, training=True, drop_rate=0.5, rng=dict(rngs[1], noise_x=xp))
    l2 = losses.pass2_loss(out2, tgt, n // 2, n // 2, lam, fm_gt, sm_gt, fac_gt, cur_step > 0.1 * num_steps)
    l2["total_loss"].backward()          # accumulates onto the pass-1 gradients, like the reference
    opt.step()
    ret = {k: l1[k] for k in ("total_loss", "cls_loss", "triplet_loss", "real_rec_loss", "real_freq_loss")}
    ret.update({k: l2[k] for k in ("freq_mask_loss", "spat_mask_loss", "fac_loss")})
    ret["cls_out"] = out["cls_out"]
    delta = {k: (v.detach() - before[k]) for k, v in named}
    return ret, delta


def check_updates(g, tag, delta, slack=0.02):
    """The first Adam steps are sign-like, so a parameter whose true gradient is 0 (e.g. the bias of a BN that only
    feeds batch-stat BNs) moves by +-lr with the sign of rounding noise in ANY implementation.  The yardstick is the
    reference itself: the fraction of update norms / leading elements on which its own fp32 and fp64 runs agree
    (~97.0 % / ~96.5 %, computed below from the golden); the run under test must reach that fraction - slack."""
    names = [str(s) for s in g[f"{tag}_names"]]
    n32, n64 = g[f"{tag}_delta_norms"], g[f"{tag}64_delta_norms"]
    own_norm = float((np.abs(n32 - n64) <= 2e-3 * n64 + 1e-12).mean())
    own_ok = own_tot = 0
    for i, k in enumerate(names):
        cnt = min(8, delta[k].numel())
        a, b = g[f"{tag}_delta_heads"][i][:cnt], g[f"{tag}64_delta_heads"][i][:cnt]
        own_ok += int((np.abs(a - b) <= 1e-3 * max(np.abs(b).max(), 1e-12) + 1e-9).sum())
        own_tot += cnt
    own_elem = own_ok / own_tot
    norm_ok, elem_ok, elem_tot = 0, 0, 0
    for i, k in enumerate(names):
        d = delta[k].double().cpu()
        rn = float(g[f"{tag}_delta_norms"][i])
        norm_ok += abs(d.norm().item() - rn) <= 2e-3 * rn + 1e-12
        head = d.flatten()[:8].numpy()
        ref = g[f"{tag}_delta_heads"][i][: head.size]
        scale = max(np.abs(ref).max(), 1e-12)
        elem_ok += int((np.abs(head - ref) <= 1e-3 * scale + 1e-9).sum())
        elem_tot += head.size
    print(f"  update norms within 2e-3: {norm_ok}/{len(names)} (reference fp32-vs-fp64: {own_norm:.3f});  "
          f"leading elements within 1e-3: {elem_ok}/{elem_tot} (reference fp32-vs-fp64: {own_elem:.3f})")
    from tests.margins import record
    record(f"{tag}: share of update norms NOT within 2e-3 (bar: the reference's own fp32-vs-fp64 share + slack)",
           1.0 - norm_ok / len(names), 1.0 - (own_norm - slack))
    record(f"{tag}: share of leading update elements NOT within 1e-3 (bar likewise)", 1.0 - elem_ok / elem_tot,
           1.0 - (own_elem - slack))
    assert norm_ok >= (own_norm - slack) * len(names), (norm_ok, len(names), own_norm)
    assert elem_ok >= (own_elem - slack) * elem_tot, (elem_ok, elem_tot, own_elem)


@pytest.mark.parametrize("tag,cur_step", [("kl", 50)])
def test_oracle_two_pass_step_matches_reference_engine(golden_dir, tag, cur_step):
    g = np.load(os.path.join(golden_dir, "udeb4_step_n4.npz"))
    ret, delta = oracle_two_pass_step(g, cur_step)
    bad = []
    for k, v in ret.items():
        key = ("out_" if k == "cls_out" else "loss_") + k
        ref = np.asarray(g[f"{tag}_{key}"], dtype=np.float64)
        err = np.abs(v.detach().double().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        print(f"  {k}: rel err {err:.3e}")
        if not err <= 1e-3:
            bad.append((k, err))
    assert not bad, bad
    check_updates(g, tag, delta)


def test_reference_fp32_vs_fp64_yardstick(golden_dir):
    """The recorded float64 run of the reference bounds how ill-conditioned the returned scalars are."""
    g = np.load(os.path.join(golden_dir, "udeb4_step_n4.npz"))
    for tag in ("early", "kl"):
        for k in g.files:
            if k.startswith(f"{tag}_loss_"):
                a, b = float(g[k]), float(g[k.replace(f"{tag}_", f"{tag}64_", 1)])
                assert abs(a - b) <= 2e-4 * abs(b), (k, a, b)
