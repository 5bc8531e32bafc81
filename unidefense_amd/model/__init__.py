"""Mirror of the reference's ``model`` package surface (model/__init__.py:1-17).
UDR50 runs at power-of-two inputs (256, 512); the 320-pixel config needs the 2^k*5 FFT sizes (not built yet)."""
from .unidefense import UniDefenseModelEb4
from .unidefense_res import UniDefenseModelRes18
from .unidefense_res50 import UniDefenseModelRes50

MODEL = {
    "UDEB4": UniDefenseModelEb4,
    "UDR18": UniDefenseModelRes18,
    "UDR50": UniDefenseModelRes50,
}


def load_model(name="UDE"):
    name_upper = name.upper()
    assert name_upper in MODEL, f"Model '{name}' not found."
    print(f"Using model: '{name}'")
    return MODEL[name_upper]
