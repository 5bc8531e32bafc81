"""TEST INFRASTRUCTURE — CPU restatement of the step's losses (plain torch)."""
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def euclidean_dist(x: Tensor, y: Tensor) -> Tensor:
    """loss/triplet_loss.py:16-30."""
    xx = (x * x).sum(1, keepdim=True)
    yy = (y * y).sum(1, keepdim=True).t()
    d = xx + yy - 2.0 * (x @ y.t())
    return d.clamp(min=1e-12).sqrt()


def aw_triplet(feat: Tensor, labels: Tensor) -> Tensor:
    """AsymmetricalWeightedTripletLoss.forward (loss/triplet_loss.py:33-82): anchors are the
    real samples (label 0), which come first in the batch; SoftMarginLoss(wn - wp, +1)."""
    n = feat.shape[0]
    n_real = int((labels == 0).sum())
    dist = euclidean_dist(feat, feat)[:n_real]
    lab = labels.reshape(1, n)
    same = (labels[:n_real].reshape(-1, 1) == lab)
    not_self = ~torch.eye(n, dtype=torch.bool, device=feat.device)[:n_real]
    is_pos = same & not_self
    is_neg = ~same
    d_ap = dist[is_pos].reshape(n_real, -1)
    d_an = dist[is_neg].reshape(n_real, -1)
    e_ap = torch.exp(d_ap)
    e_an = torch.exp(-d_an)
    wp = e_ap / (e_ap.sum(1, keepdim=True) + 1e-12)
    wn = e_an / (e_an.sum(1, keepdim=True) + 1e-12)
    fwp = (wp * d_ap).sum(1)
    fwn = (wn * d_an).sum(1)
    # nn.SoftMarginLoss: mean(log(1 + exp(-y * x))), y = 1
    return torch.log1p(torch.exp(-(fwn - fwp))).mean()


def factorization(a: Tensor, b: Tensor, off_w: float = 0.005, eps: float = 1e-6) -> Tensor:
    """FactorizationLoss.forward (loss/calib_loss.py:17-28); torch.std is unbiased."""
    an = (a - a.mean(0)) / (a.std(0) + eps)
    bn = (b - b.mean(0)) / (b.std(0) + eps)
    c = an.t() @ bn / a.shape[0]
    d = c.shape[0]
    on = ((torch.diagonal(c) - 1.0) ** 2).mean()
    off = (c * c).sum() - (torch.diagonal(c) ** 2).sum()
    return on + off_w * off / (d * d - d)


def pass1_loss(out: dict, tgt: Tensor, sum_real: int, sum_fake: int, lam: dict) -> dict:
    """Loss assembly of the clean pass (engine/abstract_engine.py:214-278)."""
    ld = out["loss_dict"]
    fm = ld["freq_mask"].mean()
    sm = ld["spat_mask"].mean()
    trip = sum(aw_triplet(f, tgt) for f in ld["triplet"])
    real_rec = ld["spatial"][:sum_real].mean()
    fake_rec = ld["spatial"][sum_real:sum_real + sum_fake].mean()
    real_freq = ld["freq"][:sum_real].mean()
    fake_freq = ld["freq"][sum_real:sum_real + sum_fake].mean()
    cls = F.cross_entropy(out["cls_out"], tgt)
    total = cls + lam.get("lambda_mask", 1.0) * fm + lam.get("lambda_mask", 1.0) * sm + \
        lam.get("lambda_triplet", 1.0) * trip + lam.get("lambda_recons", 1.0) * real_rec + \
        lam.get("lambda_freq", 1.0) * real_freq
    return {"total_loss": total, "cls_loss": cls, "triplet_loss": trip, "real_rec_loss": real_rec,
            "fake_rec_loss": fake_rec, "real_freq_loss": real_freq, "fake_freq_loss": fake_freq,
            "freq_mask_loss": fm, "spat_mask_loss": sm}


def pass2_loss(out: dict, tgt: Tensor, sum_real: int, sum_fake: int, lam: dict,
               freq_mask_gt: Tensor, spat_mask_gt: Tensor, fac_gt: Tensor, kl: bool) -> dict:
    """Loss assembly of the perturbed pass (engine/abstract_engine.py:294-371).
    kl = (cur_step > 0.1 * num_steps)."""
    ld = out["loss_dict"]
    trip = sum(aw_triplet(f, tgt) for f in ld["triplet"])
    real_rec = ld["spatial"][:sum_real].mean()
    real_freq = ld["freq"][:sum_real].mean()
    cls = F.cross_entropy(out["cls_out"], tgt)
    if kl:
        def kld(pred, gt):
            n = pred.shape[0]
            p = torch.log_softmax(pred.reshape(n, -1), -1)
            g = torch.log_softmax(gt.reshape(n, -1), -1)
            # nn.KLDivLoss(reduction="batchmean", log_target=True)(p, g)
            return (torch.exp(g) * (g - p)).sum() / n
        fm = kld(ld["freq_mask"], freq_mask_gt)
        sm = kld(ld["spat_mask"], spat_mask_gt)
    else:
        fm = ld["freq_mask"].mean()
        sm = ld["spat_mask"].mean()
    fac = factorization(ld["factorization"], fac_gt)
    total = 0.1 * cls + lam.get("lambda_mask", 1.0) * fm + lam.get("lambda_mask", 1.0) * sm + \
        lam.get("lambda_triplet", 1.0) * trip + lam.get("lambda_recons", 1.0) * 0.1 * real_rec + \
        lam.get("lambda_freq", 1.0) * 0.1 * real_freq + lam.get("lambda_fac", 1.0) * fac
    return {"total_loss": total, "cls_loss": cls, "triplet_loss": trip, "freq_mask_loss": fm,
            "spat_mask_loss": sm, "fac_loss": fac}
