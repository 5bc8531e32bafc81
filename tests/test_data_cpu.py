"""CPU: RealFakePrefetcher — batch assembly order, epoch restarts of both sources, error propagation."""
import pytest
import torch


def _source(n_batches, bs, tag):
    return [(torch.full((bs, 3, 4, 4), float(tag * 100 + i)), torch.full((bs,), tag, dtype=torch.int32))
            for i in range(n_batches)]


def test_prefetcher_order_and_epoch_restart():
    from unidefense_amd.engine.data import RealFakePrefetcher
    pf = RealFakePrefetcher(_source(3, 2, 0), _source(2, 2, 1))       # different epoch lengths
    seen = []
    for step in range(1, 8):
        xr, yr, xf, yf = pf(step, 2, 4, "cpu")
        assert xr.dtype == torch.float32 and yr.dtype == torch.int64 and yr.eq(0).all() and yf.eq(1).all()
        seen.append((int(xr[0, 0, 0, 0]), int(xf[0, 0, 0, 0])))
    assert seen == [(0, 100), (1, 101), (2, 100), (0, 101), (1, 100), (2, 101), (0, 100)]


def test_prefetcher_surfaces_source_errors():
    from unidefense_amd.engine.data import RealFakePrefetcher

    def bad():
        yield torch.zeros(1, 3, 4, 4), torch.zeros(1)
        raise ValueError("decode failed")
    pf = RealFakePrefetcher(bad(), _source(2, 1, 1))
    pf(1, 1, 4, "cpu")
    with pytest.raises(RuntimeError, match="data source failed"):
        pf(2, 1, 4, "cpu")


class _ToyFaces:
    """A dataset with the reference's contract: __getitem__ -> (path, label); load_item(paths, labels, crop) decodes."""

    def __init__(self, n, label):
        self.items = [(f"vid{i // 4}/frame{i}.png", label) for i in range(n)]
        self.decoded = 0

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def load_item(self, paths, labels, crop=None):
        ids = [int(p.rsplit("frame", 1)[1].split(".")[0]) for p in paths]
        imgs = torch.stack([torch.full((3, crop or 8, crop or 8), float(i)) for i in ids])
        return {"images": imgs, "path": paths}


@pytest.mark.parametrize("workers", [0, 2])
def test_worker_loader_decodes_off_the_main_process_and_reshuffles(workers):
    from unidefense_amd.engine.data import RealFakePrefetcher, worker_loader
    real, fake = _ToyFaces(12, 0), _ToyFaces(8, 1)
    src_r = worker_loader(real, 4, workers=workers, crop=6, seed=3)
    src_f = worker_loader(fake, 4, workers=workers, crop=6, seed=5)
    assert len(src_r) == 3 and len(src_f) == 2
    ep0 = [(x[:, 0, 0, 0].tolist(), y.tolist()) for x, y in src_r]
    ep1 = [(x[:, 0, 0, 0].tolist(), y.tolist()) for x, y in src_r]
    for ep in (ep0, ep1):                       # every epoch: each frame once, decoded at the requested crop, labels kept
        assert sorted(int(v) for xs, _ in ep for v in xs) == list(range(12)) and all(y == [0] * 4 for _, y in ep)
    assert ep0 != ep1                           # re-shuffled per epoch (seed + epoch)
    x0, _ = next(iter(src_r))
    assert x0.shape == (4, 3, 6, 6) and x0.dtype == torch.float32
    # rank sharding like DistributedSampler: two ranks split each epoch without overlap
    a = worker_loader(real, 2, workers=0, seed=3, rank=0, world=2)
    b = worker_loader(real, 2, workers=0, seed=3, rank=1, world=2)
    ia = sorted(int(v) for x, _ in a for v in x[:, 0, 0, 0].tolist())
    ib = sorted(int(v) for x, _ in b for v in x[:, 0, 0, 0].tolist())
    assert len(ia) == len(ib) == 6 and sorted(ia + ib) == list(range(12))
    # and through the prefetcher, in TrainEngine's iterator signature
    pf = RealFakePrefetcher(src_r, src_f)
    xr, yr, xf, yf = pf(1, 4, 6, "cpu")
    assert xr.shape == (4, 3, 6, 6) and yr.eq(0).all() and yf.eq(1).all()


@pytest.mark.parametrize("n,world,bs,drop_last", [(127, 2, 32, False), (127, 2, 32, True), (13, 4, 2, False), (3, 4, 2, False),
                                                  (64, 8, 4, True)])
def test_every_rank_gets_the_same_number_of_equal_batches(n, world, bs, drop_last):
    """The reference shards with DistributedSampler (engine/forgery_engine.py:67-86), which pads the epoch to a multiple of
    the world size: all ranks step the same number of times with the same batch sizes (unequal counts would deadlock the
    gradient all-reduce / the SyncBN exchange, unequal rows would break the fused SyncBN's M * world count)."""
    from unidefense_amd.engine.data import DecodedBatches
    ds = _ToyFaces(n, 0)
    per_rank = []
    for rank in range(world):
        db = DecodedBatches(ds, bs, crop=2, seed=7, rank=rank, world=world, drop_last=drop_last)
        per_rank.append([db[i][0][:, 0, 0, 0].int().tolist() for i in range(len(db))])
    counts = {len(b) for b in per_rank}
    assert len(counts) == 1, [len(b) for b in per_rank]
    for i in range(len(per_rank[0])):
        assert len({len(b[i]) for b in per_rank}) == 1
    seen = [v for b in per_rank for batch in b for v in batch]
    if not drop_last:
        assert set(seen) == set(range(n))                   # padding only ever repeats samples, never drops one
        assert len(seen) == -(-n // world) * world


def test_uint8_hand_over_is_converted_by_the_device_transform():
    """sources may yield uint8 pixels (a quarter of the bytes through the worker IPC / staging copy / PCIe): they stay uint8 up to
    the device, where `device_transform` makes the float batch; the default is a plain .float()"""
    from unidefense_amd.engine.data import DecodedBatches, RealFakePrefetcher

    class U8Faces(_ToyFaces):
        def load_item(self, paths, labels, crop=None):
            return {"images": torch.full((len(paths), 3, 4, 4), 255, dtype=torch.uint8), "path": paths}
    db = DecodedBatches(U8Faces(8, 0), 4, shuffle=False, keep_dtype=True)
    img, lab = db[0]
    assert img.dtype == torch.uint8 and lab.dtype == torch.int64
    assert DecodedBatches(U8Faces(8, 0), 4, shuffle=False)[0][0].dtype == torch.float32
    src = [(torch.full((2, 3, 4, 4), 255, dtype=torch.uint8), torch.zeros(2, dtype=torch.int32))]
    pf = RealFakePrefetcher(src, src, device_transform=lambda u8: u8.float().mul_(2.0 / 255.0).sub_(1.0))
    xr, yr, xf, yf = pf(1, 2, 4, "cpu")
    assert xr.dtype == torch.float32 and torch.allclose(xr, torch.ones_like(xr)) and yr.dtype == torch.int64
    xr, _, _, _ = RealFakePrefetcher(src, src)(1, 2, 4, "cpu")
    assert xr.dtype == torch.float32 and float(xr.max()) == 255.0
