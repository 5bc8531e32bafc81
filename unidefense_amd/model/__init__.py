"""Mirror of the reference's ``model`` package surface (model/__init__.py:1-17)."""
from .unidefense import UniDefenseModelEb4

MODEL = {
    "UDEB4": UniDefenseModelEb4,
}


def load_model(name="UDE"):
    name_upper = name.upper()
    assert name_upper in MODEL, f"Model '{name}' not found."
    print(f"Using model: '{name}'")
    return MODEL[name_upper]
