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


class RealFakePrefetcher:
    """Callable with TrainEngine's iterator signature `(step, batch, size, device) -> (x_real, y_real, x_fake, y_fake)`.

    real, fake: re-iterable sources (e.g. DataLoaders) yielding `(images [B,3,H,W] float, labels [B] int64)`."""

    def __init__(self, real, fake, depth=2):
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
                batch = [xr.float(), yr.long(), xf.float(), yf.long()]
                if cuda:
                    batch = [t.pin_memory() for t in batch]
                    with torch.cuda.stream(self.stream):
                        batch = [t.to(self.device, non_blocking=True) for t in batch]
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                else:
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
