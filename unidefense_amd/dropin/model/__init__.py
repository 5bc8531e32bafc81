"""`import model` of the reference (model/__init__.py) -> unidefense_amd.model."""
from unidefense_amd.model import *          # noqa: F401,F403
from unidefense_amd.model import load_model  # noqa: F401
