"""Multi-GPU: one process per GPU, utterances sharded across ranks, flat gradient all-reduces per step (the scorer's
buffer after the step, the decoder-side buffer on a side stream, one bucket per decoder layer behind that layer's
attention backward).

The reference uses Lightning's DDPStrategy over NCCL (train_w2v.py:195-196, train_simul.py:197-198): per-rank
replicas, the batch split by the distributed sampler, bucketed gradient all-reduce during backward.  The hot
path has no activation exchange (every utterance is independent through scorer, select, scan, gather and
cross-attention), so here each rank runs the NuggetStep on its own utterances and the step ends with a single
all-reduce (mean) of the flattened parameter gradients over RCCL/xGMI (backend "nccl" on ROCm); on CPU the
same code runs over gloo (tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment. Returns (rank, local_rank, world_size)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # STAR_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("STAR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            # binding the group to its device lets barrier() and the first collective skip the device guess
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard(items: Sequence, rank: int, world: int) -> List:
    """Utterances of a global batch owned by `rank`: a contiguous, near-equal slice (sizes differ by <= 1)."""
    n = len(items)
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return list(items[lo:hi])


def _all_reduce(t: torch.Tensor, op):
    """RCCL all-reduce on the tensor itself; under the gloo rehearsal backend a device tensor is staged through the
    host (gloo's device-tensor path hangs when two ranks share one card)."""
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


class FlatGradAllReduce:
    """All-reduce (mean) of the gradients of `params` as one flat buffer -> one collective per step.

    A ring all-reduce over xGMI is per-link bound, so fewer, larger messages win; the scorer's gradients are
    4.2 MB in fp32 at d = 1024, far below the point where bucketing would help."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=p0.device)
        self.attached = False

    def attach(self):
        """Make every parameter's .grad a view of the flat buffer (fp32 parameters only): backward then accumulates
        straight into it, the step needs no gather / scatter copies around the collective (8 small launches per step
        for the scorer), and the addresses stay fixed under HIP-graph replay.  Call zero() where the step used to
        reset the gradients to None."""
        if any(p.dtype != torch.float32 for p in self.params):
            raise TypeError("FlatGradAllReduce.attach: fp32 parameters only")
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            o += n
        self.attached = True
        return self

    def zero(self):
        self.flat.zero_()

    def _still_attached(self):
        o = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                return False
            o += p.numel()
        return True

    def __call__(self, world: int | None = None):
        world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        if self.attached and self._still_attached():
            if world > 1:
                _all_reduce(self.flat, dist.ReduceOp.SUM)
                self.flat.div_(world)
            return self.flat
        o = 0
        views = []
        for p in self.params:
            n = p.numel()
            v = self.flat[o:o + n].view_as(p)
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            views.append(v)
            o += n
        if world > 1:
            _all_reduce(self.flat, dist.ReduceOp.SUM)
            self.flat.div_(world)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone().to(p.dtype)
            else:
                p.grad.copy_(v)
        return self.flat


class SideStreamAllReduce:
    """All-reduce (mean) of a second flat gradient buffer on a stream of its own, beside the rest of the backward.

    The decoder's cross-attention parameter gradients (nl x (4 d^2 + 6 d) fp32 values, 67 MB at d = 1024, nl = 4) are
    complete as soon as the attention backward kernels are done, long before the gather / scorer backward has
    finished.  start() makes the side stream wait for the work queued so far and enqueues the collective there;
    join() makes the current stream wait for it.  On CPU tensors (gloo tests) both run in order on the host."""

    def __init__(self, flat: torch.Tensor):
        self.flat = flat
        self.stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None

    def start(self, world: int | None = None):
        world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        if world <= 1:
            return
        if self.stream is None:
            _all_reduce(self.flat, dist.ReduceOp.SUM)
            self.flat.div_(world)
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            _all_reduce(self.flat, dist.ReduceOp.SUM)
            self.flat.div_(world)

    def join(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)


class BucketedSideAllReduce:
    """A flat gradient buffer cut into equal buckets, each all-reduced (mean) on ONE shared side stream as soon as the
    caller says its gradients are complete -- DDP's bucketed overlap with the backward (the reference's
    DDPStrategy does the same, train_w2v.py:195-196).  start(i) makes the side stream wait for the work queued on the
    current stream so far and enqueues bucket i behind the buckets already started; join() makes the current stream
    wait for all of them.  Over xGMI a ring all-reduce is per-link bound, so a bucket per decoder layer (17 MB at
    d = 1024) is still far above the size where the launch latency shows."""

    def __init__(self, flat: torch.Tensor, n_buckets: int):
        if flat.numel() % n_buckets:
            raise ValueError("BucketedSideAllReduce: buffer size must be a multiple of the bucket count")
        self.flat = flat
        self.buckets = list(flat.view(n_buckets, -1).unbind(0))
        self.stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None

    def start(self, i: int, world: int | None = None):
        world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        if world <= 1:
            return
        b = self.buckets[i]
        if self.stream is None:
            _all_reduce(b, dist.ReduceOp.SUM)
            b.div_(world)
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            _all_reduce(b, dist.ReduceOp.SUM)
            b.div_(world)

    def join(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
