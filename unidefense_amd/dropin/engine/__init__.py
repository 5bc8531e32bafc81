"""`import engine` of the reference (engine/__init__.py) -> unidefense_amd.engine."""
from unidefense_amd.engine import ENGINE, AbstractEngine, TrainEngine, get_engine  # noqa: F401
