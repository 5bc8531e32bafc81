"""Loss registry with the reference's lookup surface (its loss/__init__.py: `get_loss(name, device)` returns the shared
criterion instance moved to `device`; `LOSSES` maps the YAML / engine names to instances)."""
import torch.nn as nn

from .calib_loss import FactorizationLoss
from .triplet_loss import AsymmetricalWeightedTripletLoss


def _instances():
    yield "cross_entropy", nn.CrossEntropyLoss()
    yield "kl_div", nn.KLDivLoss(reduction="batchmean", log_target=True)     # log-space targets (abstract_engine.py:300)
    yield "aw_triplet", AsymmetricalWeightedTripletLoss()                    # HIP value + gradient when the batch layout is known
    yield "factorization", FactorizationLoss()
    yield "bce", nn.BCEWithLogitsLoss()
    yield "mse", nn.MSELoss()


LOSSES = dict(_instances())


def get_loss(name="cross_entropy", device="cuda:0"):
    criterion = LOSSES[name]                     # KeyError for an unknown name, like the reference
    print(f"Using loss: '{criterion}'")
    return criterion.to(device)
