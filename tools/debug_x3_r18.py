"""GPU: every plain GEMM of one UDR18 train step (actual operands) on both arithmetic paths against float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import param_fill
from tests import oracle_util as ou
from tests.test_r18 import make_rng_r18
from unidefense_amd import kernels as K, lib
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model
n = 8
dev = torch.device("cuda:0")
x = param_fill.make_input(n, 128, 42); tgt = param_fill.make_labels(n); rng = make_rng_r18(n, 142)
lam = ou.SMOOTH_LAMBDAS
m = load_model("UDR18")(num_classes=2, drop_rate=0.5); param_fill.fill_module_(m, 0.0, 0.3); m = m.to(dev).train()
orig = {k: getattr(K, k) for k in ("gemm_nt", "gemm_nn", "gemm_tn")}
rows = []
def wrap(kind):
    def f(a, b, *args, **kw):
        if args or kw.get("out") is not None or kw.get("accumulate"):
            return orig[kind](a, b, *args, **kw)
        A = a.double().t() if kind == "gemm_tn" else a.double()
        B = b.double().t() if kind == "gemm_nt" else b.double()
        ref = A @ B
        res = []
        for p in (1, 2):
            lib.call("ud_gemm_set_path", p)
            y = orig[kind](a, b)
            res.append(((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item())
        lib.call("ud_gemm_set_path", 0)
        rows.append((res[1] / max(res[0], 1e-12), kind, tuple(a.shape), tuple(b.shape), res[0], res[1],
                     (A.abs() @ B.abs()).max().item() / max(ref.abs().max().item(), 1e-300)))
        return orig[kind](a, b)
    return f
for k in orig:
    setattr(K, k, wrap(k))
out = m(x.to(dev), rng=rng)
ld, t = out["loss_dict"], tgt.to(dev)
trip = sum(LOSSES["aw_triplet"](f, t) for f in ld["triplet"])
(LOSSES["cross_entropy"](out["cls_out"], t) + lam["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) + lam["lambda_triplet"] * trip).backward()
rows.sort(reverse=True)
print("ratio  kind  a.shape b.shape  err(fp32-mfma)  err(split-bf16)  cancellation sum|ab|/|result|   [errors relative to max|result|]")
for r in rows[:25]:
    print("%8.2f %s %s %s  %.3e  %.3e  %.1f" % r)
print(len(rows), "gemms;  worst split-bf16 err %.3e, worst fp32 err %.3e" % (max(r[5] for r in rows), max(r[4] for r in rows)))
