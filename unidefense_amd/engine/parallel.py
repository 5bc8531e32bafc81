"""Data parallelism for the HIP model: one process per GPU, RCCL over xGMI
(reference: SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel, engine/forgery_engine.py:142-146).

The path shards over the batch only.  Exchanges per backward:
  * parameter gradients — bucketed flat all-reduce.  The whole network is a single autograd node, so the exchange
    is driven from inside that node's backward (model/unidefense.py:_NetFunction): the tape hands every parameter
    gradient to the GradReducer the moment it is final (reverse parameter order); large gradients (>= 2 MB: the
    spectral 1x1 weights, 85 % of the bytes) are all-reduced in place, the small ones are packed into ~64 MB buckets;
    every all-reduce is launched ASYNCHRONOUSLY (it runs on the process group's own stream), so the exchange of
    the deep layers' gradients overlaps the backward of the shallow ones; the backward waits for all buckets at its
    end and returns views of the reduced flat buffers (no copy back).  The mean over ranks comes for free: the
    incoming loss gradient is scaled by 1/world before the tape runs (everything downstream is linear in it,
    including the SyncBN sums).
  * SyncBatchNorm statistics, enabled by ``sync_bn=True``.  Fused MBConv path (tape.mbconv_fused): the fp64
    accumulators (sum x, sum x^2) of a BatchNorm are all-reduced IN PLACE between the kernel that fills them and the
    kernels that consume them (tape.DataParallelCtx.reduce), likewise (sum dz, sum dz*xhat) in the backward — one
    all_reduce of 2C doubles each way, ranks need not hold equal row counts.  Operator path (tape.batchnorm_act:
    attention / head / ResNet models): one all_gather of (mean, var) per BN forward and one all_reduce of the two sums
    per BN backward.
No other collective exists on the data path.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import tape as T


class GradReducer:
    """Sums gradients over the ranks of `group`, bucket by bucket, asynchronously.

    begin() -> ready(key, g) in the order gradients become final -> finish() -> {key: reduced g (a view of its
    bucket's flat buffer)}.  Device-agnostic (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, group, bucket_bytes, direct_bytes=2 << 20):
        self.group = group
        self.bucket_bytes = bucket_bytes
        # A gradient of at least this size is reduced IN PLACE as a collective of its own instead of being packed:
        # the spectral 1x1 weights (1344^2 .. 3264^2: 7 - 43 MB each) are 85 % of the 513 MB of gradients, and
        # copying them into a bucket would move them through HBM twice more for nothing.
        self.direct_bytes = direct_bytes
        self.begin()

    def begin(self):
        self.bucket, self.size, self.pending, self.done = [], 0, [], set()

    def ready(self, key, g):
        self.done.add(key)
        nbytes = g.numel() * g.element_size()
        if nbytes >= self.direct_bytes and g.is_contiguous():
            work = dist.all_reduce(g, group=self.group, async_op=True)
            self.pending.append((work, g.view(-1), [(key, g)]))
            return
        self.bucket.append((key, g))
        self.size += nbytes
        if self.size >= self.bucket_bytes:
            self.flush()

    def flush(self):
        if not self.bucket:
            return
        flat = torch.cat([g.reshape(-1) for _, g in self.bucket])
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        self.pending.append((work, flat, self.bucket))
        self.bucket, self.size = [], 0

    def finish(self):
        self.flush()
        out = {}
        for work, flat, bucket in self.pending:
            work.wait()                      # GPU: the current stream waits for the collective's stream
            off = 0
            for key, g in bucket:
                n = g.numel()
                out[key] = flat[off:off + n].view(g.shape)
                off += n
        self.pending = []
        return out


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
        self.force = T.FORCE_COLLECTIVES           # single-GPU exercise of the RCCL calls (tape.py)
        if self.world > 1 or self.force:
            module._grad_reducer = GradReducer(process_group, bucket_bytes)
            module._grad_prescale = 1.0 / self.world
        if sync_bn and (self.world > 1 or self.force):
            module._sync_bn_group = process_group if process_group is not None else dist.group.WORLD

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def sync_grads(self, grads):
        """grads: list of tensors (or None) in parameter order -> averaged over ranks (a new list; the one-shot
        form of what _NetFunction.backward streams)."""
        if self.world == 1 and not self.force:
            return grads
        red = GradReducer(self.process_group, self.bucket_bytes)
        for i in reversed(range(len(grads))):
            if grads[i] is not None:
                red.ready(i, grads[i])
        out = red.finish()
        return [None if g is None else out[i].mul_(1.0 / self.world) for i, g in enumerate(grads)]


def wrap_data_parallel(model: nn.Module, local_rank: int = 0, process_group=None, sync_bn: bool = True):
    return HipDataParallel(model, process_group, sync_bn=sync_bn)
