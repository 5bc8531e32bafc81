"""what a plain streaming kernel reaches on this part: torch copy / add (2 reads + 1 write) against the fused BatchNorm-backward apply
and the other fused elementwise kernels on the same tensor sizes (GB/s of algorithmic bytes, graph-replayed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
for G, R, Cc in ((32, 4096, 192), (32, 1024, 336), (32, 256, 960), (32, 64, 1632), (32, 16384, 144)):
    x = torch.randn(G, R, Cc, device=dev); dy = torch.randn_like(x); out = torch.empty_like(x)
    nb = x.numel() * 4
    t = {}
    t["copy (1r 1w)"] = (K._time_launches(lambda: out.copy_(x)), 2 * nb)
    t["add (2r 1w)"] = (K._time_launches(lambda: torch.add(x, dy, out=out)), 3 * nb)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    acc = K.zeros64(2 * Cc, x); K.colstats(x.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, G * R, gamma, beta, 1e-3, 1)
    sacc = K.zeros64(2 * Cc, x); K.normbwd_sums(x, dy, None, 1.0, bn, False, G, R, sacc)
    t["normbwd_apply (2r 1w)"] = (K._time_launches(lambda: K.normbwd_apply(x, dy, None, 1.0, bn, False, G, R, sacc)), 3 * nb)
    t["normbwd_sums (2r)"] = (K._time_launches(lambda: K.normbwd_sums(x, dy, None, 1.0, bn, False, G, R, sacc)), 2 * nb)
    s = torch.randn(G, Cc, device=dev)
    t["se_scale_bn (1r 1w)"] = (K._time_launches(lambda: K.se_scale_bn(x, bn, s, G, R)), 2 * nb)
    pool = K.zeros64(G * Cc, x)
    t["colsum_bn (1r)"] = (K._time_launches(lambda: K.colsum_bn(x, bn, G, R, pool)), nb)
    print("%d x %d x %d (%.0f MB): " % (G, R, Cc, nb / 1e6) + "  ".join("%s %.1f us %.2f TB/s" % (k, v * 1e3, b / (v * 1e-3) / 1e12) for k, (v, b) in t.items()), flush=True)
