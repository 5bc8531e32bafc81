"""Mirror of the reference's ``model`` package surface (model/__init__.py:1-17).
UDR50 (model/unidefense.py:439-631) is not built yet (needs the 2^k*5 FFT sizes of the 320-pixel config)."""
from .unidefense import UniDefenseModelEb4
from .unidefense_res import UniDefenseModelRes18

MODEL = {
    "UDEB4": UniDefenseModelEb4,
    "UDR18": UniDefenseModelRes18,
}


def load_model(name="UDE"):
    name_upper = name.upper()
    assert name_upper in MODEL, f"Model '{name}' not found."
    print(f"Using model: '{name}'")
    return MODEL[name_upper]
