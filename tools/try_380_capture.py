"""where does hipGraph capture of the 380 x 380 step stop?  (traceback of the first non-capturable call)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unidefense_amd.loss import LOSSES
from unidefense_amd.model import load_model
dev = torch.device("cuda:0")
bs, size = 4, 380
m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5).to(dev).train()
x = (2 * torch.rand(bs, 3, size, size) - 1).to(dev)
tgt = torch.tensor([0] * (bs // 2) + [1] * (bs // 2), device=dev)
LOSSES["aw_triplet"].n_real = bs // 2
def step():
    for p in m.parameters():
        p.grad = None
    out = m(x)
    loss = bench.pass1_loss(out, tgt, bs // 2, LOSSES)
    loss.backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
s = torch.cuda.Stream()
try:
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            step()
    print("captured OK")
except Exception:
    traceback.print_exc()
