"""Data parallelism for the HIP model: one process per GPU, RCCL over xGMI
(reference: SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel, engine/forgery_engine.py:142-146).

The path shards over the batch only.  Exchanges per backward:
  * parameter gradients — bucketed flat all-reduce (sum / world).  The whole network is a single autograd
    node, so the exchange is driven from inside that node's backward (model/unidefense.py:_NetFunction):
    gradients are packed in the order backward produced them (reverse parameter order) into ~64 MB buckets.
  * SyncBatchNorm statistics — one all_gather of (mean, var) per BN forward and one all_reduce of
    (sum dz, sum dz*xhat) per BN backward (tape.batchnorm_act), enabled by ``sync_bn=True``.
No other collective exists on the data path.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import tape as T


class HipDataParallel(nn.Module):
    """Minimal DDP replacement exposing ``.module`` like torch's wrapper."""

    def __init__(self, module: nn.Module, process_group=None, bucket_bytes: int = 64 << 20, sync_bn: bool = True):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.bucket_bytes = bucket_bytes
        self.world = dist.get_world_size(process_group)
        # same initial state on every rank (DDP broadcasts rank 0's parameters and buffers)
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=process_group)
        module._grad_sync = self.sync_grads
        self.force = T.FORCE_COLLECTIVES           # single-GPU exercise of the RCCL calls (tape.py)
        if sync_bn and (self.world > 1 or self.force):
            module._sync_bn_group = process_group if process_group is not None else dist.group.WORLD

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def sync_grads(self, grads):
        """grads: list of tensors (or None) in parameter order -> averaged over ranks (in place)."""
        if self.world == 1 and not self.force:
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

        for g in reversed([g for g in grads if g is not None]):
            bucket.append(g)
            size += g.numel() * g.element_size()
            if size >= self.bucket_bytes:
                flush()
        flush()
        return grads


def wrap_data_parallel(model: nn.Module, local_rank: int = 0, process_group=None, sync_bn: bool = True):
    return HipDataParallel(model, process_group, sync_bn=sync_bn)
