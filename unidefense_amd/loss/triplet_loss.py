"""Asymmetrical weighted triplet loss on [N, d] feature vectors.

Reference semantics: loss/triplet_loss.py:16-82.  On the GPU, with the batch layout known (`n_real` set by the engine),
the value AND the feature gradient come from two HIP launches (`_FusedAWTriplet` -> csrc/loss.hip, `ud_aw_triplet`) instead of
~90 tiny torch kernels per feature.  Everything else (CPU tensors, unknown layout, normalised features handled before the call)
takes the torch formulation below, written with multiplicative masks instead of the reference's boolean gather + reshape so
that it needs no host synchronisation (graph-capture safe) when ``n_real`` is supplied; the value is identical whenever every
anchor has at least one positive and one negative (the reference additionally requires equal counts per anchor for its reshape).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def pairwise_distance(x: torch.Tensor) -> torch.Tensor:
    """sqrt(clamp(|xi|^2 + |xj|^2 - 2 xi.xj, 1e-12))  (loss/triplet_loss.py:16-30)."""
    sq = (x * x).sum(dim=1, keepdim=True)
    d2 = sq + sq.t() - 2.0 * torch.mm(x, x.t())
    return d2.clamp(min=1e-12).sqrt()


class _FusedAWTriplet(torch.autograd.Function):
    """The whole loss and its feature gradient in two HIP launches (csrc/loss.hip, ud_aw_triplet) instead of ~90
    tiny torch kernels per feature."""

    @staticmethod
    def forward(ctx, feat, n_real):
        from .. import kernels as K
        loss, dfeat = K.aw_triplet(feat.detach().contiguous(), n_real)
        ctx.save_for_backward(dfeat)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dfeat,) = ctx.saved_tensors
        return dfeat * g, None


class AsymmetricalWeightedTripletLoss(nn.Module):
    """Anchors are the real samples (label 0), which come first in the batch (loss/triplet_loss.py:46-53).
    Per anchor: positives weighted by softmax(+d), negatives by softmax(-d); SoftMargin(wn - wp, +1)."""

    def __init__(self):
        super().__init__()
        self.n_real = None      # optional: set by the engine (= sum_real) to avoid a device read-back

    def forward(self, global_feat, labels, normalize_feature=False):
        if normalize_feature:
            global_feat = global_feat / (global_feat.norm(2, dim=-1, keepdim=True) + 1e-12)
        n = global_feat.shape[0]
        if self.n_real is not None and global_feat.is_cuda and global_feat.dtype == torch.float32 \
                and 0 < self.n_real < n:
            return _FusedAWTriplet.apply(global_feat, self.n_real)
        n_real = self.n_real if self.n_real is not None else int((labels == 0).sum().item())
        dist = pairwise_distance(global_feat)[:n_real]                        # [R, N]
        same = labels[:n_real].unsqueeze(1) == labels.unsqueeze(0)              # [R, N]
        eye = torch.eye(n, dtype=torch.bool, device=labels.device)[:n_real]
        pos = (same & ~eye).to(dist.dtype)
        neg = (~same).to(dist.dtype)
        e_ap = torch.exp(dist) * pos
        e_an = torch.exp(-dist) * neg
        wp = e_ap / (e_ap.sum(1, keepdim=True) + 1e-12)
        wn = e_an / (e_an.sum(1, keepdim=True) + 1e-12)
        margin = (wn * dist).sum(1) - (wp * dist).sum(1)
        return F.soft_margin_loss(margin, torch.ones_like(margin))
