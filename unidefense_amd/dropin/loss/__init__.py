"""`import loss` of the reference (loss/__init__.py) -> unidefense_amd.loss."""
from unidefense_amd.loss import LOSSES, get_loss  # noqa: F401
