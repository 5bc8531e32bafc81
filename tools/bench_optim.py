"""GPU: time of one AdamW(amsgrad) step on the UDEB4 parameter set: torch foreach vs torch fused."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd.model import load_model
from unidefense_amd.engine.optim import param_groups_weight_decay
dev = torch.device("cuda:0")
m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2).to(dev)
for p in m.parameters():
    p.grad = torch.randn_like(p) * 1e-3
n = sum(p.numel() for p in m.parameters())
for kw in (dict(foreach=True), dict(fused=True)):
    try:
        opt = torch.optim.AdamW(param_groups_weight_decay(m, 5e-6), lr=1e-4, betas=(0.9, 0.999), amsgrad=True, **kw)
        for _ in range(3): opt.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): opt.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(kw, "%.2f ms/step  (%.1f M params, %.0f GB/s of 36 B/param)" % (dt * 1e3, n / 1e6, n * 36 / dt / 1e9))
    except Exception as e:
        print(kw, "failed:", type(e).__name__, str(e)[:200])
