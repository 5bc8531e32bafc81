"""ud_split_planes_h2t on the step's activation shapes: us per launch, bytes (4 B read + 4 B written per element) / time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K
dev = torch.device("cuda:0")
print("UD_SPLIT_PW =", os.environ.get("UD_SPLIT_PW", "(default)"))
for R, Cc in ((131072, 192), (32768, 336), (17408, 672), (8192, 960), (4608, 1920), (2048, 1632), (1280, 3264), (131072, 32), (8192, 160)):
    x = torch.randn(R, Cc, device=dev)
    pl = K.split_planes(x, prec=2)
    amax = K.empty((256,), x)
    K._call("ud_absmax", K._p(x), R, Cc, x.stride(0), K._p(amax), K._stream())
    t = K._time_launches(lambda: K.split_planes(x, pl, prec=2, absmax=amax))
    print("%7d x %4d  %6.1f us  %.2f TB/s" % (R, Cc, t * 1e3, 8.0 * R * Cc / (t * 1e-3) / 1e12), flush=True)
