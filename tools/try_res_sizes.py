"""the ResNet models at input sizes whose feature maps have no in-register FFT (224: 56 / 28 / 14 / 7): eval vs the oracle, and a train step"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r18, r50, param_fill
from unidefense_amd.model import load_model
dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
for name, shapes, fwd, size in (("UDR50", r50.r50_state_shapes, r50.forward_r50, 224), ("UDR18", r18.r18_state_shapes, r18.forward_r18, 224),
                                ("UDR50", r50.r50_state_shapes, r50.forward_r50, 384)):
    try:
        m = load_model(name)(num_classes=2, drop_rate=0.5)
        param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
        m = m.to(dev)
        x = param_fill.make_input(2, size, seed=5)
        sd = param_fill.fill_state_dict(shapes(2), 0.0, 0.3)
        with torch.no_grad():
            ref = fwd(sd, x, training=False)
            got = m.eval()(x.to(dev))
        print(name, size, {k: "%.2e" % rel(got[k], ref[k]) for k in ("cls_out", "rec")}, flush=True)
        m.train()
        out = m(x.to(dev))
        (out["cls_out"].sum() + out["rec"].mean()).backward()
        print("   train step ok,", sum(1 for p in m.parameters() if p.grad is not None), "grads", flush=True)
    except Exception:
        traceback.print_exc()
