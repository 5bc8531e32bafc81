"""Mirror of the reference's ``engine`` package surface (engine/__init__.py:1-14)."""
from .abstract_engine import AbstractEngine
from .train_engine import TrainEngine

# the reference's three concrete engines share the train step and differ in dataset / metric plumbing (out of scope):
# all three names resolve to the one concrete engine built here
ENGINE = {
    "FE": TrainEngine,
    "OCIM": TrainEngine,
    "UE": TrainEngine,
}


def get_engine(name='UE'):
    print(f"Using engine: '{name}'")
    return ENGINE[name]
