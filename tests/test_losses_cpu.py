"""CPU: the product's loss modules (unidefense_amd.loss, torch ops on [N,d] tensors) against the oracle."""
import torch

from oracle import losses as OL
from unidefense_amd.loss import LOSSES, AsymmetricalWeightedTripletLoss, FactorizationLoss


def test_triplet_matches_oracle():
    torch.manual_seed(0)
    for n, d in ((8, 40), (32, 160), (6, 1792)):
        f = torch.randn(n, d, dtype=torch.float64)
        lab = torch.tensor([0] * (n // 2) + [1] * (n // 2))
        a = AsymmetricalWeightedTripletLoss()(f, lab)
        b = OL.aw_triplet(f, lab)
        assert abs(a.item() - b.item()) < 1e-12 * max(1, abs(b.item()))
        m = AsymmetricalWeightedTripletLoss()
        m.n_real = n // 2
        assert abs(m(f, lab).item() - b.item()) < 1e-12 * max(1, abs(b.item()))


def test_factorization_matches_oracle():
    torch.manual_seed(1)
    a, b = torch.randn(16, 64, dtype=torch.float64), torch.randn(16, 64, dtype=torch.float64)
    x = FactorizationLoss()(a, b)
    y = OL.factorization(a, b)
    assert abs(x.item() - y.item()) < 1e-12


def test_loss_table_keys():
    assert set(LOSSES) == {"mse", "bce", "factorization", "cross_entropy", "aw_triplet", "kl_div"}
