"""Engine lookup with the reference's surface (its engine/__init__.py: `get_engine(name)` -> class, names FE / OCIM /
UE, KeyError otherwise, one "Using engine" line).  The reference's three concrete engines share the train step and differ
in dataset / metric / wandb plumbing (out of scope): all three names resolve to the one concrete engine built here."""
from .abstract_engine import AbstractEngine
from .train_engine import TrainEngine

ENGINE = dict.fromkeys(("FE", "OCIM", "UE"), TrainEngine)


def get_engine(name='UE'):
    cls = ENGINE[name]
    print(f"Using engine: '{name}'")
    return cls
