import sys, torch, numpy as np
sys.path.insert(0, ".")
from tests.test_e_mixed_precision_gpu import _step
from unidefense_amd import tape as T
dev = torch.device("cuda:0")
for n in (16, 64):
    o32, g32 = _step(dev, 0, n=n)
    oy, gy = _step(dev, 0, n=n, round_params=True)
    rms = {k: float((oy[k] - o32[k]).norm() / o32[k].norm().clamp_min(1e-30)) for k in o32}
    rel = [float((gy[k] - a).norm()) / float(a.norm()) for k, a in g32.items()
           if float(a.norm()) >= 1e-6 and not k.endswith("._bn2.bias") and not k.endswith("_coef")]
    r = np.array(rel)
    print(f"n={n} YARDSTICK (fp32 step, parameters + input rounded to fp16 once): out rms", {k: f"{v:.2e}" for k, v in rms.items()},
          "grad relL2 50/90/99/max: %.3g %.3g %.3g %.3g" % (np.percentile(r, 50), np.percentile(r, 90), np.percentile(r, 99), r.max()))
    for storage in ("fp32", "half"):
        for tiled in (False, True):
            T._DW_TILED = tiled
            o16, g16 = _step(dev, 3, n=n, half_storage=storage == "half")
            rms = {k: float((o16[k] - o32[k]).norm() / o32[k].norm().clamp_min(1e-30)) for k in o32}
            rel = []
            for k, a in g32.items():
                na = float(a.norm())
                if na < 1e-6 or k.endswith("._bn2.bias") or k.endswith("_coef"):
                    continue
                rel.append(float((g16[k] - a).norm()) / na)
            r = np.array(rel)
            print(f"n={n} storage={storage} tiled={tiled}: out rms", {k: f"{v:.2e}" for k, v in rms.items()},
                  "grad relL2 50/90/99/max: %.3g %.3g %.3g %.3g" % (np.percentile(r, 50), np.percentile(r, 90), np.percentile(r, 99), r.max()))
    T._DW_TILED = True
