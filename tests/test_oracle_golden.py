"""CPU: the oracle (oracle/eb4.py, oracle/losses.py) reproduces the REFERENCE's outputs that
oracle/make_golden.py recorded by importing /root/reference (tests/golden/*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import eb4, param_fill
from tests import oracle_util as ou

GRAD_RTOL, GRAD_ATOL = 1e-3, 2e-5
RTOL = 2e-5   # fp32 CPU restatement vs fp32 CPU reference (different op grouping only)


def _close(a, b, name, rtol=RTOL):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f"{name}: rel err {err:.3e} > {rtol}"


def _check_outputs(out, g):
    ld = out["loss_dict"]
    _close(out["cls_out"].detach(), g["cls_out"], "cls_out")
    _close(torch.nn.functional.adaptive_avg_pool2d(out["rec"].detach(), 8), g["rec_pool8"], "rec")
    _close(ld["factorization"].detach()[:, :64], g["factorization"], "factorization")
    for k in ("freq_mask", "spat_mask", "spatial", "freq"):
        _close(ld[k].detach(), g[k], k)
    for i in range(3):
        _close(ld["triplet"][i].detach(), g[f"triplet{i}"], f"triplet{i}")


@pytest.mark.parametrize("fname,sf,fuse", [("udeb4_eval_n2.npz", 0.0, 0.3),
                                           ("udeb4_eval_n2_init.npz", -10.0, 0.0)])
def test_eval_matches_reference(golden_dir, fname, sf, fuse):
    g = np.load(os.path.join(golden_dir, fname))
    n, size, seed = [int(v) for v in g["meta"]]
    sd = ou.oracle_state(sf, fuse)
    x = param_fill.make_input(n, size, seed)
    with torch.no_grad():
        out = eb4.forward_eb4(sd, x, training=False)
    _check_outputs(out, g)


@pytest.mark.parametrize("variant,fname", [("full", "udeb4_train_n4.npz"), ("smooth", "udeb4_train_n4.npz"),
                                           ("smooth", "udeb4_train_n8.npz")])
def test_train_fwd_bwd_matches_reference(golden_dir, variant, fname):
    g = np.load(os.path.join(golden_dir, fname))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    lam = ou.LAMBDAS if variant == "full" else ou.SMOOTH_LAMBDAS
    sd = ou.oracle_state(0.0, 0.3, requires_grad=True)
    x = param_fill.make_input(n, size, seed)
    tgt = param_fill.make_labels(n)
    rng = ou.make_rng(n, mseed, 0.5)
    out, ls = ou.oracle_train_pass1(sd, x, tgt, rng, drop_rate=0.5, lam=lam)
    _check_outputs(out, g)
    assert out["_max_gap"].item() > 1e-3     # the fixture's batch has no near-tie in torch.max
    for k in ("total_loss", "cls_loss", "triplet_loss", "real_rec_loss", "real_freq_loss"):
        _close(ls[k].item(), g[f"{variant}_loss_" + k], k)
    names = [str(s) for s in g["grad_names"]]
    assert len(names) == 504
    worst = 0.0
    for i, k in enumerate(names):
        gr = sd[k].grad
        assert gr is not None, k
        ref_norm = float(g[f"{variant}_grad_norms"][i])
        got = gr.double().norm().item()
        # allclose-style bound (rtol on the tensor's norm + atol).  The atol matters for the bias of a
        # BN whose output only feeds (via 1x1 convs) other batch-stat BNs: its gradient is
        # mathematically ZERO and both sides hold ~5e-6 of rounding noise there.
        tol = GRAD_RTOL * ref_norm + GRAD_ATOL
        err = abs(got - ref_norm)
        head = gr.flatten()[:8].numpy()
        herr = np.abs(head - g[f"{variant}_grad_heads"][i][: head.size]).max()
        worst = max(worst, err / tol, herr / tol)
        assert err < tol and herr < tol, f"{k}: norm err {err:.2e} head err {herr:.2e} tol {tol:.2e}"
    print("worst grad err / tol", worst)
