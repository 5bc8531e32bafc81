"""Batch feeding for TrainEngine (the part next to the hot path): real + fake batch assembly
and host-to-device prefetch.  The reference pulls `(path, label)` lists from two DataLoaders and decodes / augments the
images in the MAIN process (`dataset.load_item`, engine/forgery_engine.py:243-266, with the Python-2 `iter.next()`);
with a 38 ms GPU step that would be the bottleneck.  Here the two sources are any iterables of `(images, labels)` CPU
tensors (a DataLoader with workers doing the decode); a background thread restarts them at epoch ends, pins the
batches, and copies them to the device on a side stream ONE STEP AHEAD of the step that consumes them.
"""
import queue
import threading

import torch
import torch.utils.data


class RealFakePrefetcher:
    """Callable with TrainEngine's iterator signature `(step, batch, size, device) -> (x_real, y_real, x_fake, y_fake)`.

    real, fake: re-iterable sources (e.g. DataLoaders) yielding `(images [B,3,H,W] float, labels [B] int64)`."""

    def __init__(self, real, fake, depth=2, device_transform=None):
        """device_transform: optional callable applied to each image batch ON THE DEVICE (copy stream) after the H2D copy —
        for sources that hand over uint8 pixels (a quarter of the bytes through the worker IPC, the pinned staging copy and
        PCIe: profiles/r06/input_pipeline.txt), e.g. `lambda u8: u8.float().mul_(2 / 255).sub_(1)`; default: `.float()`."""
        self.device_transform = device_transform
        self.sources = (real, fake)
        self.q = queue.Queue(maxsize=depth)
        self.stream = None
        self.thread = None
        self.device = None
        self.error = None

    @staticmethod
    def _cycle(source):
        while True:
            n = 0
            for item in source:                    # a fresh iterator per epoch (the reference's `% len(loader) == 1`)
                n += 1
                yield item
            if n == 0:
                raise RuntimeError("RealFakePrefetcher: a data source yielded nothing")

    def _worker(self):
        try:
            its = [self._cycle(s) for s in self.sources]
            cuda = self.device.type == "cuda"
            while True:
                (xr, yr), (xf, yf) = next(its[0]), next(its[1])
                tf = self.device_transform or (lambda t: t.float())
                if cuda:
                    # images travel in the dtype the source yields (uint8 stays uint8 until it is on the device)
                    batch = [t.pin_memory() for t in (xr, yr.long(), xf, yf.long())]
                    with torch.cuda.stream(self.stream):
                        batch = [t.to(self.device, non_blocking=True) for t in batch]
                        batch[0], batch[2] = tf(batch[0]), tf(batch[2])
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                else:
                    batch = [tf(xr), yr.long(), tf(xf), yf.long()]
                    ev = None
                self.q.put((batch, ev))
        except Exception as e:                      # noqa: BLE001 — surfaced on the consumer side
            self.error = e
            self.q.put((None, None))

    def __call__(self, step, batch, size, device):
        device = torch.device(device)
        if self.thread is None:
            self.device = device
            if device.type == "cuda":
                self.stream = torch.cuda.Stream(device=device)
            self.thread = threading.Thread(target=self._worker, daemon=True)
            self.thread.start()
        tensors, ev = self.q.get()
        if tensors is None:
            raise RuntimeError("data source failed") from self.error
        if ev is not None:
            torch.cuda.current_stream(device).wait_event(ev)          # order the copies before this step's kernels
            for t in tensors:
                t.record_stream(torch.cuda.current_stream(device))
        return tuple(tensors)


class DecodedBatches(torch.utils.data.Dataset):
    """Moves the reference's main-process decode into DataLoader workers.

    The reference's datasets yield `(path, label)` from `__getitem__` and decode / augment whole batches in the MAIN
    process through `dataset.load_item(paths, labels, crop=...)` -> `{'images': [B,3,H,W], 'path': ...}`
    (dataset/abstract_dataset.py:101-160, called at engine/forgery_engine.py:251-266).  This wrapper indexes BATCHES:
    item i = `load_item` of the i-th batch of a (seeded, per-epoch reshuffled) index order, run inside a worker process,
    so that `workers` batches decode in parallel while the GPU steps.  `rank` / `world` shard the order like the
    reference's DistributedSampler (forgery_engine.py:67-86): the epoch's order is padded by wrapping around to a multiple
    of `world` (torch's DistributedSampler with drop_last=False), so EVERY rank sees the same number of samples, hence the
    same number of equally sized batches — unequal step counts would leave the gradient all-reduces and the SyncBN
    exchange of the longer ranks waiting for peers that have finished."""

    def __init__(self, dataset, batch_size, crop=None, shuffle=True, seed=0, rank=0, world=1, drop_last=True, keep_dtype=False):
        """keep_dtype: hand `load_item`'s images over as they are (e.g. uint8, converted on the device by the prefetcher's
        `device_transform`) instead of as float32"""
        self.keep_dtype = bool(keep_dtype)
        self.dataset, self.batch_size, self.crop = dataset, int(batch_size), crop
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, int(seed), int(rank), int(world), drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _order(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        total = self._per_rank() * self.world
        while len(idx) < total:                                      # pad by wrapping (n < world: more than once)
            idx += idx[:total - len(idx)]
        return idx[self.rank:total:self.world]

    def _per_rank(self):
        return -(-len(self.dataset) // self.world)

    def __len__(self):
        n = self._per_rank()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __getitem__(self, i):
        idx = self._order()[i * self.batch_size:(i + 1) * self.batch_size]
        items = [self.dataset[j] for j in idx]                       # (path, label) pairs
        paths = [it[0] for it in items]
        labels = torch.as_tensor([int(it[1]) for it in items], dtype=torch.int64)
        kw = {} if self.crop is None else {"crop": self.crop}
        out = self.dataset.load_item(paths, labels, **kw)
        img = out["images"]
        return (img if self.keep_dtype else img.float()).contiguous(), labels


def worker_loader(dataset, batch_size, workers=4, **kw):
    """A re-iterable source for RealFakePrefetcher: batches decoded by `workers` processes, two batches prefetched per
    worker, re-shuffled every epoch.  workers = 0 decodes in the calling thread (the prefetcher's), still off the main one."""
    ds = DecodedBatches(dataset, batch_size, **kw)

    class _Epochs:
        def __init__(self):
            self.epoch = 0

        def __len__(self):
            return len(ds)

        def __iter__(self):
            ds.set_epoch(self.epoch)
            self.epoch += 1
            loader = torch.utils.data.DataLoader(ds, batch_size=None, shuffle=False, num_workers=workers,
                                                 prefetch_factor=2 if workers > 0 else None,
                                                 persistent_workers=False)
            return iter(loader)
    return _Epochs()
