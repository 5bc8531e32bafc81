"""Mirror of the reference's ``model`` package surface (model/__init__.py:1-17).
UDR50 runs at 256x256 and at 320x320 (BASELINE configs[3]: 5*2^k FFT sizes)."""
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
