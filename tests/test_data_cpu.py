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
