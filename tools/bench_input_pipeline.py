"""SURVEY 8(f) rank 3 measured: `worker_loader` + `RealFakePrefetcher` (unidefense_amd/engine/data.py — the reference's
main-process `load_item`, engine/forgery_engine.py:243-266, moved into DataLoader workers + a pinned side-stream H2D one
step ahead) feeding the graph-replayed UDEB4 256 x 256 bs-32 step (16 real + 16 fake per step, as the engine assembles them).

The decode itself is the user's dataset; here a SYNTHETIC decode of comparable shape: per image a uint8 HWC array is drawn
from a per-path seed (the "decoded JPEG"), flipped at random, converted to float, normalised to [-1, 1] and transposed to
CHW — numpy work in the worker process, the batch handed over as one float32 tensor [B, 3, 256, 256] (3.1 MB per image set
of 16: 12.6 MB per source and step, 25 MB per step over PCIe).

  python tools/bench_input_pipeline.py [--steps 60] [--workers 6] > profiles/r06/input_pipeline.txt
"""
import argparse
import contextlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class SyntheticFaces(torch.utils.data.Dataset):
    """(path, label) items + the reference's `load_item(paths, labels) -> {'images': [B,3,H,W]}` batch decode"""

    def __init__(self, n, label, size, uint8=False):
        self.n, self.label, self.size, self.uint8 = n, label, size, uint8

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return f"img_{self.label}_{i:07d}.png", self.label

    def load_item(self, paths, labels, crop=None):
        S = self.size
        out = np.empty((len(paths), 3, S, S), dtype=np.uint8 if self.uint8 else np.float32)
        for k, p in enumerate(paths):
            rng = np.random.default_rng(abs(hash(p)) % (1 << 32))
            img = rng.integers(0, 256, size=(S, S, 3), dtype=np.uint8)          # the "decoded" image
            if rng.random() < 0.5:
                img = img[:, ::-1]
            out[k] = img.transpose(2, 0, 1) if self.uint8 else (img.astype(np.float32) * (2.0 / 255.0) - 1.0).transpose(2, 0, 1)
        return {"images": torch.from_numpy(out), "path": paths}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workers", type=int, default=6, help="decode processes per source (real, fake)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--uint8", action="store_true", help="the dataset hands over uint8 pixels; float conversion + normalisation on "
                                                        "the device (RealFakePrefetcher(device_transform=...))")
    args = ap.parse_args()
    import bench
    from unidefense_amd.engine.data import RealFakePrefetcher, worker_loader
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    dev = torch.device("cuda:0")
    bs, half = args.batch, args.batch // 2
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(sys.stderr):
        model = load_model("UDEB4")(num_classes=2, drop_rate=0.5, extractor="efficientnet-b4").to(dev).train()
    x = torch.zeros(bs, 3, 256, 256, device=dev)
    tgt = torch.tensor([0] * half + [1] * half, device=dev)
    LOSSES["aw_triplet"].n_real = half
    params = [p for p in model.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        loss = bench.pass1_loss(model(x), tgt, half, LOSSES)
        loss.backward()
        return loss
    x.copy_(2 * torch.rand_like(x) - 1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    torch.cuda.synchronize()

    def timed(feed, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            feed(i)
            graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # (a) the step alone, inputs resident
    for _ in range(args.warmup):
        graph.replay()
    t_alone = timed(lambda i: None, args.steps)

    # (b) fed by the pipeline: real + fake sources, decode in worker processes, pinned H2D one step ahead, D2D into the
    # captured step's static input
    n_img = half * (args.steps + args.warmup + 8)
    real = worker_loader(SyntheticFaces(n_img, 0, 256, args.uint8), half, workers=args.workers, keep_dtype=args.uint8)
    fake = worker_loader(SyntheticFaces(n_img, 1, 256, args.uint8), half, workers=args.workers, keep_dtype=args.uint8)
    feeder = RealFakePrefetcher(real, fake, depth=3,
                                device_transform=(lambda u8: u8.float().mul_(2.0 / 255.0).sub_(1.0)) if args.uint8 else None)

    def feed(i):
        xr, yr, xf, yf = feeder(i, bs, 256, dev)
        x[:half].copy_(xr, non_blocking=True)
        x[half:].copy_(xf, non_blocking=True)
    for i in range(args.warmup):
        feed(i)
        graph.replay()
    t_fed = timed(feed, args.steps)

    # (c) the decode alone in ONE process (what the reference's main-process load_item would cost per step)
    ds = SyntheticFaces(64, 0, 256, args.uint8)
    t0 = time.perf_counter()
    for r in range(4):
        ds.load_item([ds[j][0] for j in range(16 * r, 16 * r + 16)], None)
    t_dec = (time.perf_counter() - t0) / 4 * 2          # two sources per step

    gb = bs * 3 * 256 * 256 * (1 if args.uint8 else 4) / 1e9
    print(f"# {'uint8 hand-over, float conversion on the device' if args.uint8 else 'float32 hand-over (the reference load_item contract)'}")
    print(f"# UDEB4 256x256 bs {bs} fwd + pass-1 loss + bwd (hipGraph replay), {args.steps} steps; {args.workers} decode workers per source; "
          f"host CPU quota {bench.host_cores()} cores")
    print(f"step alone (inputs resident in HBM):                 {1e3 * t_alone:7.2f} ms  {bs / t_alone:7.0f} img/s")
    print(f"step fed by worker_loader + RealFakePrefetcher:      {1e3 * t_fed:7.2f} ms  {bs / t_fed:7.0f} img/s   "
          f"H2D {gb / t_fed:5.2f} GB/s of pixels   ({100 * (t_fed / t_alone - 1):+.1f} % vs resident)")
    print(f"synthetic decode of one step's 2 x {half} images in ONE process (the reference's main-process load_item): "
          f"{1e3 * t_dec:7.1f} ms  -> {bs / t_dec:6.0f} img/s if it ran in the training process")
    sys.stdout.flush()
    os._exit(0)          # (daemon feeder thread + worker processes: no orderly teardown needed for a measurement)


if __name__ == "__main__":
    main()
