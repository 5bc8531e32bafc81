"""Model registry with the reference's lookup surface (its model/__init__.py:7-17: `load_model(name)` -> class, names
case-insensitive, AssertionError for an unknown name, one "Using model" line).  UDR50 accepts 256x256 and 320x320
inputs (BASELINE configs[3]: FFT sizes 5*2^k)."""
from . import unidefense as _eb4
from . import unidefense_res as _r18
from . import unidefense_res50 as _r50

_REGISTRY = (
    ("UDEB4", _eb4.UniDefenseModelEb4),        # EfficientNet-b4 + SFConv backbone
    ("UDR18", _r18.UniDefenseModelRes18),      # ResNet18 extractor, 128x128 .. 512x512
    ("UDR50", _r50.UniDefenseModelRes50),      # ResNet50 extractor
)
MODEL = dict(_REGISTRY)
UniDefenseModelEb4, UniDefenseModelRes18, UniDefenseModelRes50 = (cls for _, cls in _REGISTRY)


def load_model(name="UDE"):
    key = str(name).upper()
    cls = MODEL.get(key)
    assert cls is not None, f"Model '{name}' not found."
    print(f"Using model: '{name}'")
    return cls
