"""GPU: time of one AdamW(amsgrad) step on the UDEB4 parameter set: torch foreach, torch fused, HipAdamW (csrc/optim.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd.model import load_model
from unidefense_amd.engine.optim import HipAdamW, param_groups_weight_decay
dev = torch.device("cuda:0")
m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2).to(dev)
for p in m.parameters():
    p.grad = torch.randn_like(p) * 1e-3
n = sum(p.numel() for p in m.parameters() if p.requires_grad)
scale, finf = torch.tensor(1024.0, device=dev), torch.tensor(0.0, device=dev)
for name, make in (("torch foreach", lambda g: torch.optim.AdamW(g, lr=1e-4, betas=(0.9, 0.999), amsgrad=True, foreach=True)),
                   ("torch fused", lambda g: torch.optim.AdamW(g, lr=1e-4, betas=(0.9, 0.999), amsgrad=True, fused=True)),
                   ("HipAdamW", lambda g: HipAdamW(g, lr=1e-4, betas=(0.9, 0.999), amsgrad=True))):
    try:
        opt = make(param_groups_weight_decay(m, 5e-6))
        if name != "torch foreach":
            opt.grad_scale, opt.found_inf = scale, finf          # what GradScaler.step hands over
        for _ in range(3): opt.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): opt.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("%-14s %.3f ms/step  (%.1f M params, %.0f GB/s of 36 B/param)" % (name, dt * 1e3, n / 1e6, n * 36 / dt / 1e9))
    except Exception as e:
        print(name, "failed:", type(e).__name__, str(e)[:200])
