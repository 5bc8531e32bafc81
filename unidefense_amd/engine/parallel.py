"""Data parallelism for the HIP model: one process per GPU, gradients averaged with RCCL over xGMI
(reference: DistributedDataParallel + SyncBatchNorm, engine/forgery_engine.py:142-146).

The whole network is a single autograd node, so the gradient exchange is driven from inside that node's
backward (see model/unidefense.py:_NetFunction): bucketed flat all-reduces (sum / world) of the parameter
gradients.  No data-path collective exists besides this and the SyncBN statistics.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class HipDataParallel(nn.Module):
    """Minimal DDP replacement exposing ``.module`` like torch's wrapper."""

    def __init__(self, module: nn.Module, process_group=None, bucket_bytes: int = 64 << 20):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.bucket_bytes = bucket_bytes
        self.world = dist.get_world_size(process_group)
        # same initial state on every rank (DDP broadcasts rank 0's parameters and buffers)
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, 0, group=process_group)
        module._grad_sync = self._sync_grads

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _sync_grads(self, grads):
        """grads: list of tensors (or None) in parameter order -> averaged over ranks, in place."""
        if self.world == 1:
            return grads
        bucket, size = [], 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat, group=self.process_group)
            flat.mul_(1.0 / self.world)
            off = 0
            for g in bucket:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
            bucket, size = [], 0

        for g in reversed([g for g in grads if g is not None]):     # roughly the order backward produced them
            bucket.append(g)
            size += g.numel() * 4
            if size >= self.bucket_bytes:
                flush()
        flush()
        return grads


def wrap_data_parallel(model: nn.Module, local_rank: int, process_group=None):
    return HipDataParallel(model, process_group)
