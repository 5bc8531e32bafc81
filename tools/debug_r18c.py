"""In-situ check of the emb_block1 identity branch (conv1x1 -> BN -> maxpool -> add_relu) with HIP activations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import param_fill
from tests import oracle_util as ou
from tests.test_r18 import make_rng_r18, r18_state
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model
n = 8
dev = torch.device("cuda:0")
x = param_fill.make_input(n, 128, 42); tgt = param_fill.make_labels(n); rng = make_rng_r18(n, 142)
lam = ou.SMOOTH_LAMBDAS
m = load_model("UDR18")(num_classes=2, drop_rate=0.5); param_fill.fill_module_(m, 0.0, 0.3); m = m.to(dev).train()
m._debug_watch = True
out = m(x.to(dev), rng=rng)
ld, t = out["loss_dict"], tgt.to(dev)
trip = sum(LOSSES["aw_triplet"](f, t) for f in ld["triplet"])
(LOSSES["cross_entropy"](out["cls_out"], t) + lam["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) + lam["lambda_triplet"] * trip).backward()
cap = m._debug_tape.captured
m._debug_watch = False
with torch.no_grad():
    r = m._run(x.to(dev), None, rng)
ext = r["_feats"]["ext"].permute(0, 3, 1, 2).double().cpu()
emb = r["_feats"]["emb"].permute(0, 3, 1, 2).double().cpu()
d_emb = cap["emb"].permute(0, 3, 1, 2).double().cpu()
sd = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
w = sd["emb_block1.downsample.0.weight"].clone().requires_grad_()
gm = sd["emb_block1.downsample.1.weight"].clone().requires_grad_(); bt = sd["emb_block1.downsample.1.bias"].clone().requires_grad_()
xe = ext.clone().requires_grad_()
idt = F.max_pool2d(F.batch_norm(F.conv2d(xe, w), None, None, gm, bt, True, 0.1, 1e-5), 3, 2, 1)
g_pool = d_emb * (emb > 0).double()
idt.backward(g_pool)
P = dict(m.named_parameters())
def rel(a, b): return ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
print("in-situ d downsample.0.weight", rel(P["emb_block1.downsample.0.weight"].grad, w.grad))
print("in-situ d downsample.1.weight", rel(P["emb_block1.downsample.1.weight"].grad, gm.grad))
# how close are max-pool winners to ties?
with torch.no_grad():
    z = F.batch_norm(F.conv2d(ext, sd["emb_block1.downsample.0.weight"]), None, None, sd["emb_block1.downsample.1.weight"], sd["emb_block1.downsample.1.bias"], True, 0.1, 1e-5)
    u = F.unfold(z, 3, padding=1, stride=2).view(n, 512, 9, -1)
    # padding contributes zeros in unfold; mask them as -inf using an index map
    ones = F.unfold(torch.ones_like(z[:, :1]), 3, padding=1, stride=2).view(n, 1, 9, -1)
    u = torch.where(ones > 0, u, torch.full_like(u, -1e30))
    top2 = u.topk(2, dim=2).values
    gap = (top2[:, :, 0] - top2[:, :, 1])
    print("max-pool top-2 gap: min %.3e  count<1e-6: %d  count==0: %d of %d" % (gap.min().item(), int((gap < 1e-6).sum()), int((gap == 0).sum()), gap.numel()))
