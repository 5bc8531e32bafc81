"""Factorization (cross-correlation) loss between two [N, d] embeddings.

Reference semantics: loss/calib_loss.py:5-28 (Barlow-Twins style): every feature standardised over the batch (unbiased
std + eps), C = A^T B / N, then mean((diag C - 1)^2) + w * mean(offdiag(C)^2).  Tiny tensors: device-side torch ops
(SURVEY.md K17).
"""
import torch
import torch.nn as nn


def _standardise(e, eps):
    return (e - e.mean(dim=0, keepdim=True)) / (e.std(dim=0, keepdim=True) + eps)


class FactorizationLoss(nn.Module):
    def __init__(self, off_diag_weight=0.005):
        super().__init__()
        self.off_diag_weight = off_diag_weight

    def forward(self, emb_a, emb_b, eps=1e-6):
        n, d = emb_a.shape
        corr = torch.einsum("nc,nd->cd", _standardise(emb_a, eps), _standardise(emb_b, eps)) / n
        diag = corr.diagonal()
        total_sq, diag_sq = corr.square().sum(), diag.square().sum()
        return (diag - 1.0).square().mean() + self.off_diag_weight * (total_sq - diag_sq) / (d * (d - 1))
