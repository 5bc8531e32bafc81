"""GPU parity tests, model level: UniDefenseModelEb4 on the HIP kernels against
  (a) the golden vectors recorded from the REFERENCE (tests/golden/*.npz, oracle/make_golden.py), and
  (b) the oracle run on the CPU on the same seeded inputs.
Tolerance 1e-3 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import eb4, param_fill
from tests import oracle_util as ou

pytestmark = pytest.mark.gpu
RTOL = 1e-3
GRAD_RTOL, GRAD_ATOL = 1e-3, 2e-5


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _model(dev, sf, fuse):
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=sf, fuse_coef=fuse)
    return m.to(dev)


def _close(a, b, name, rtol=RTOL):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    print(f"  {name}: rel err {err:.3e}")
    return name, err


def _check_outputs(out, g):
    ld = out["loss_dict"]
    res = [_close(out["cls_out"], g["cls_out"], "cls_out"),
           _close(torch.nn.functional.adaptive_avg_pool2d(out["rec"].detach().cpu(), 8), g["rec_pool8"], "rec"),
           _close(ld["factorization"][:, :64], g["factorization"], "factorization")]
    for k in ("freq_mask", "spat_mask", "spatial", "freq"):
        res.append(_close(ld[k], g[k], k))
    for i in range(3):
        res.append(_close(ld["triplet"][i], g[f"triplet{i}"], f"triplet{i}"))
    bad = [(n, e) for n, e in res if not e <= RTOL]
    assert not bad, bad


@pytest.mark.parametrize("fname,sf,fuse", [("udeb4_eval_n2.npz", 0.0, 0.3), ("udeb4_eval_n2_init.npz", -10.0, 0.0)])
def test_eval_vs_reference_golden(golden_dir, fname, sf, fuse):
    dev = _dev()
    g = np.load(os.path.join(golden_dir, fname))
    n, size, seed = [int(v) for v in g["meta"]]
    m = _model(dev, sf, fuse).eval()
    x = param_fill.make_input(n, size, seed).to(dev)
    with torch.no_grad():
        out = m(x)
    _check_outputs(out, g)


def test_eval_intermediates_vs_oracle():
    """Stage-by-stage comparison with the oracle (localises a failing kernel)."""
    dev = _dev()
    m = _model(dev, 0.0, 0.3).eval()
    x = param_fill.make_input(2, 256, 5)
    sd = ou.oracle_state(0.0, 0.3)
    with torch.no_grad():
        ref = eb4.forward_eb4(sd, x, training=False)
        got = m._run(x.to(dev), None, None)
    bad = []
    for k in ("x_b4", "x_b5", "dec1", "dec2"):
        n_, e = _close(got["_feats"][k].permute(0, 3, 1, 2), ref["_feats"][k], k)
        if not e <= RTOL:
            bad.append((n_, e))
    n_, e = _close(got["_feats"]["dec3"], ref["_feats"]["dec3"], "dec3")
    if not e <= RTOL:
        bad.append((n_, e))
    assert not bad, bad


def test_train_fwd_bwd_vs_reference_golden(golden_dir):
    dev = _dev()
    from unidefense_amd.loss import LOSSES
    g = np.load(os.path.join(golden_dir, "udeb4_train_n4.npz"))
    n, size, seed, mseed = [int(v) for v in g["meta"]]
    m = _model(dev, 0.0, 0.3).train()
    x = param_fill.make_input(n, size, seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    rng = ou.make_rng(n, mseed, 0.5)
    out = m(x, rng=rng)
    _check_outputs(out, g)
    ld = out["loss_dict"]
    lam = ou.LAMBDAS
    n_real = n // 2
    trip = sum(LOSSES["aw_triplet"](f, tgt) for f in ld["triplet"])
    cls = LOSSES["cross_entropy"](out["cls_out"], tgt)
    real_rec = ld["spatial"].narrow(0, 0, n_real).mean()
    real_freq = ld["freq"].narrow(0, 0, n_real).mean()
    total = cls + lam["lambda_mask"] * ld["freq_mask"].mean() + lam["lambda_mask"] * ld["spat_mask"].mean() + \
        lam["lambda_triplet"] * trip + lam["lambda_recons"] * real_rec + lam["lambda_freq"] * real_freq
    for k, v in (("total_loss", total), ("cls_loss", cls), ("triplet_loss", trip), ("real_rec_loss", real_rec),
                 ("real_freq_loss", real_freq)):
        _, e = _close(v, g["loss_" + k], k)
        assert e <= RTOL, (k, e)
    total.backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    worst, bad = 0.0, []
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        ref_norm = float(g["grad_norms"][i])
        tol = GRAD_RTOL * ref_norm + GRAD_ATOL
        err = abs(gr.double().norm().item() - ref_norm)
        head = gr.flatten()[:8].cpu().numpy()
        herr = float(np.abs(head - g["grad_heads"][i][: head.size]).max())
        worst = max(worst, err / tol, herr / tol)
        if not (err < tol and herr < tol):
            bad.append((k, err, herr, tol))
    print("worst grad err / tol", worst, "bad", len(bad))
    assert not bad, bad[:20]
