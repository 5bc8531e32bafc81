"""Diagnostic: parameter-gradient deviation of the mixed-precision modes from the fp32 step, with and without the common
scale factor (the bottleneck BatchNorm1d over few samples rescales all upstream gradients uniformly)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_e_mixed_precision_gpu import _step
dev = torch.device("cuda:0")
o32, g32 = _step(dev, 0)
for name, kw in (("fp32 storage", {}), ("half storage", {"half_storage": True})):
    o16, g16 = _step(dev, 3, **kw)
    rel, sc, keys = [], [], []
    for k, a in g32.items():
        na = float(a.norm())
        if na < 1e-6 or k.endswith("._bn2.bias") or k.endswith("_coef"):
            continue
        b = g16[k]
        rel.append(float((b - a).norm()) / na)
        sc.append(float((a * b).sum()) / na ** 2)
        keys.append(k)
    r, sc = np.array(rel), np.array(sc)
    s = np.median(sc)
    res = np.array([float((g16[k] / s - g32[k]).norm() / g32[k].norm()) for k in keys])
    print("%s: rel L2 50/90/99/max %.3g %.3g %.3g %.3g | common scale %.4f (10%%..90%%: %.4f..%.4f) | after removing it 50/90/99/max %.3g %.3g %.3g %.3g"
          % (name, *np.percentile(r, [50, 90, 99]), r.max(), s, *np.percentile(sc, [10, 90]), *np.percentile(res, [50, 90, 99]), res.max()), flush=True)
