"""CPU: the C-ABI library loads and exports every symbol include/unidefense_hip.h declares, the ctypes
binding (unidefense_amd/lib.py) covers the same set, and the product fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "unidefense_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(ud_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from unidefense_amd import lib
    names = _declared()
    assert len(names) >= 38
    handle = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert sorted(lib.EXPORTED) == names, (set(lib.EXPORTED) ^ set(names))
    lib.load()


def test_invalid_arguments_are_rejected_without_gpu():
    from unidefense_amd import lib
    # argument validation happens before any HIP call
    assert lib.load().ud_gemm(None, None) == -1000
    assert lib.load().ud_reduce_ws_doubles(0, 16, 8) == -1000      # G < 1
    assert lib.load().ud_rfft2(None, None, 1, 14, 4, 1.0, 1.0, 0, None) == -1000   # no in-register transform for this side (kernels.py falls back to DFT matrices)
    assert lib.load().ud_rfft2(None, None, 1, 20, 4, 1.0, 1.0, 1, None) == -1000   # 5*2^k sizes: fp32 storage only
    with pytest.raises(lib.UDLibraryError):
        lib.call("ud_reduce_ws_doubles", 0, 16, 8)


def test_model_refuses_cpu_tensors():
    from unidefense_amd.model import load_model
    m = load_model("udeb4")(extractor="efficientnet-b4", num_classes=2)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.eval()(torch.zeros(1, 3, 256, 256))


def test_state_dict_keys_match_reference_layout():
    """802 keys / 505 parameters / 128.3 M values with the reference's names."""
    from oracle import eb4
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    sd = m.state_dict()
    want = eb4.eb4_state_shapes(2)
    assert set(sd) == set(want)
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    assert len(list(m.parameters())) == 505
    assert not m.bottleneck.bias.requires_grad


def test_dropin_shims_resolve_the_reference_import_names():
    """`from engine import get_engine`, `from model import load_model`, `from loss import get_loss`
    (main.py:5-6 and the engines of the reference) with unidefense_amd/dropin first on sys.path — in a fresh interpreter."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from engine import get_engine\nfrom model import load_model\nfrom loss import get_loss\n"
            "import unidefense_amd.engine as e, unidefense_amd.model as m\n"
            "assert get_engine is e.get_engine and load_model is m.load_model\n"
            "assert get_engine('UE').__name__ == 'TrainEngine' and load_model('udr50').__name__ == 'UniDefenseModelRes50'\n"
            "print('ok')" % (os.path.join(ROOT, "unidefense_amd", "dropin"), ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]
