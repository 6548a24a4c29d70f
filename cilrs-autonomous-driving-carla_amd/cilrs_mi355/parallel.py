"""Data parallel over the GPUs of one node: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference is single-GPU (notebook/notebook.ipynb:479); BASELINE.json asks for the batch to be
sharded across 8 GPUs with the gradient all-reduce overlapped with backward.  Frames are
independent except through BatchNorm batch statistics, which stay per-replica (each GPU normalises
over its own 128 frames -- the reference's single-GPU arithmetic at B=128, and PyTorch DDP's
default).  The ONLY collective is a sum-all-reduce of the flat gradient arena, issued as three
contiguous buckets as soon as the backward segments that fill them finish:

    bucket 0  layer4 + heads   (14.2 M floats)  after segment 1
    bucket 1  layer3           ( 6.8 M floats)  after segment 2
    bucket 2  stem+layer1+2    ( 1.4 M floats)  after segment 5

xGMI is point-to-point, so few large buckets beat many small ones; the first bucket carries 63 %
of the bytes and is in flight while layers 3..1 (70 % of backward FLOPs) still compute.
The 1/world_size averaging is folded into the Adam kernel's gradient scale.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .engine import segment_ranges


def bucket_plan(seg_ranges):
    """[(last_segment, begin, end)] -- contiguous arena ranges, reverse-layer order."""
    heads, l4, l3, l2, l1, stem = seg_ranges
    assert l4[1] == heads[0] and stem[1] == l1[0] and l1[1] == l2[0]
    return [(1, l4[0], heads[1]), (2, l3[0], l3[1]), (5, stem[0], l2[1])]


class BucketedAllReduce:
    def __init__(self, flat_grads: torch.Tensor, process_group=None, buckets=None, variant=0):
        self.flat = flat_grads
        self.pg = process_group if process_group is not None else dist.group.WORLD
        self.world_size = dist.get_world_size(self.pg)
        self.buckets = buckets if buckets is not None else bucket_plan(segment_ranges(variant))
        self._pending = []

    def reduce_bucket(self, i):
        _, b, e = self.buckets[i]
        # async: the collective is ordered after everything already enqueued on the current
        # stream and runs on the process group's own stream, overlapping later compute
        self._pending.append(dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM, group=self.pg,
                                             async_op=True))

    def wait_all(self):
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def backward_and_reduce(self, eng, plan, dcontrols, dpred_speed, after_bucket=None):
        """Segment-wise backward, one asynchronous all-reduce per finished bucket.  after_bucket
        (i, begin, end), if given, is called for every bucket in issue order once the compute
        stream has been made to wait for that bucket's collective -- the Trainer updates the
        bucket's parameter range there, so the Adam launches of the early (large) buckets run
        while the last (small) bucket's all-reduce, which nothing else can hide, is in flight."""
        seg = 0
        for i, (last_seg, _, _) in enumerate(self.buckets):
            eng.run_backward(plan, dcontrols, dpred_speed, seg, last_seg + 1)
            seg = last_seg + 1
            self.reduce_bucket(i)
        if after_bucket is None:
            self.wait_all()
            return
        for i, w in enumerate(self._pending):
            w.wait()
            _, b, e = self.buckets[i]
            after_bucket(i, b, e)
        self._pending.clear()


def broadcast_parameters(eng, process_group=None, src=0):
    """Make every replica start from rank `src`'s weights and BN buffers."""
    pg = process_group if process_group is not None else dist.group.WORLD
    for t in (eng.params, eng.bn, eng.nbt):
        dist.broadcast(t, src=src, group=pg)
    eng.weights_epoch += 1            # cached eval-mode state (folded BatchNorm, graphs) is stale


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL needs dmabuf IPC here
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"),
                                rank=rank, world_size=world)
    return rank, world, local
