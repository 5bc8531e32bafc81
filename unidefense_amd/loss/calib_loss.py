"""Factorization (cross-correlation) loss between two [N, d] embeddings.

Reference semantics: loss/calib_loss.py:5-28 (Barlow-Twins style): standardise each feature over the batch
(unbiased std + eps), C = A^T B / N, mean((diag - 1)^2) + w * mean(offdiag^2).  Tiny tensors, torch ops
(SURVEY.md K17).
"""
import torch
import torch.nn as nn


class FactorizationLoss(nn.Module):
    def __init__(self, off_diag_weight=0.005):
        super().__init__()
        self.off_diag_weight = off_diag_weight

    def forward(self, emb_a, emb_b, eps=1e-6):
        a = (emb_a - emb_a.mean(0)) / (emb_a.std(0) + eps)
        b = (emb_b - emb_b.mean(0)) / (emb_b.std(0) + eps)
        c = torch.mm(a.t(), b) / a.shape[0]
        d = c.shape[0]
        diag = torch.diagonal(c)
        on_diag = (diag - 1.0).pow(2).mean()
        off_diag = (c.pow(2).sum() - diag.pow(2).sum()) / (d * d - d)
        return on_diag + self.off_diag_weight * off_diag
