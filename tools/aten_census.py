"""GPU: census of the torch-native (aten) ops one UDEB4 train step still issues, with the unidefense_amd call site —
every one of them is a ~4 us kernel launch in the replayed hipGraph."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model
import bench

dev = torch.device("cuda:0")
bs = 8
model = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5).to(dev).train()
x = (2 * torch.rand(bs, 3, 256, 256) - 1).to(dev)
tgt = torch.tensor([0] * (bs // 2) + [1] * (bs // 2), device=dev)
LOSSES["aw_triplet"].n_real = bs // 2
counts = collections.Counter()

class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ("aten.view", "aten._unsafe_view", "aten.detach", "aten.t.", "aten.permute", "aten.alias",
                                      "aten.slice", "aten.select", "aten.expand", "aten.empty", "aten.as_strided", "aten.unsqueeze",
                                      "aten.squeeze", "aten.transpose", "aten.reshape", "aten.narrow", "aten.is_", "aten.sym_",
                                      "aten.stride", "aten.size", "aten._reshape_alias", "aten.unbind", "aten.split", "aten.lift")):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/unidefense_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            counts[(name, site)] += 1
        return func(*args, **(kwargs or {}))

def step():
    for p in model.parameters():
        p.grad = None
    out = model(x)
    loss = bench.pass1_loss(out, tgt, bs // 2, LOSSES)
    loss.backward()
step()
torch.cuda.synchronize()
with Census():
    step()
torch.cuda.synchronize()
tot = sum(counts.values())
print("aten ops with a kernel launch (approx):", tot)
for (name, site), c in counts.most_common(60):
    print("%5d  %-38s %s" % (c, name, site))
