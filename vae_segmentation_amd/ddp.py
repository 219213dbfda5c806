"""Data parallelism for the train step: one process per GPU, replicated parameters, ONE exchange per step —
an RCCL all-reduce (torch.distributed backend "nccl" is RCCL on ROCm) of the trainable gradients in a flat fp32
bucket.  This replaces the reference's single-process nn.DataParallel (main_source.py:354, main_target.py:436):
its per-forward parameter broadcast and output gather disappear (replicas stay bit-identical because every rank
applies the same averaged gradient), and its gradient reduce-to-GPU0 becomes the all-reduce.

InstanceNorm is per-sample and the Dice/KL losses are per-sample means, so with equal per-rank batch the mean of
the per-rank losses equals the reference's global-batch loss and the averaged gradient equals its gradient."""
import torch
import torch.distributed as dist

from ._lib import check, lib
from .optim import _Tables


class FlatGradSync:
    """gather grads (x 1/world) into a flat bucket -> all_reduce(sum) -> expose flat views as the grads to apply."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._tab = _Tables()

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src, group=self.group)

    def __call__(self, grads=None):
        """grads: tensors aligned with self.params (default: p.grad).  Returns the averaged-gradient views."""
        grads = [p.grad for p in self.params] if grads is None else grads
        if self.flat.is_cuda:
            from . import ops
            ops.join_side()
            (sp, dp, sizes, bm), nb = self._tab.get([grads, self.views], self.flat.device)
            check(lib.vs_copy_scale_multi(sp.data_ptr(), dp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb,
                                          1.0 / self.world, torch.cuda.current_stream().cuda_stream), "copy_scale_multi")
        else:   # gloo / CPU tensors: host-side staging for the multi-process CPU tests of the exchange logic
            for v, g in zip(self.views, grads):
                v.copy_(g)
            self.flat.mul_(1.0 / self.world)
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        return self.views
