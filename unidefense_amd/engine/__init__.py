"""Mirror of the reference's ``engine`` package surface (engine/__init__.py:1-14)."""
from .abstract_engine import AbstractEngine

ENGINE = {}


def get_engine(name='UE'):
    print(f"Using engine: '{name}'")
    return ENGINE[name]
