"""Instance sharding across ranks: contiguous instance ranges, rig replicated,
no data-path collective during evaluation; one all-gather per frame assembles the per-instance
bounding boxes (always) and, on request, the vertex buffers.

torch.distributed is plumbing only: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(instances: int, rank: int, world: int) -> Tuple[int, int]:
    """GPU g of G gets [g*ceil(I/G), min(I,(g+1)*ceil(I/G)))."""
    per = (instances + world - 1) // world
    lo = min(instances, rank * per)
    hi = min(instances, lo + per)
    return lo, hi


def padded_count(instances: int, world: int) -> int:
    """Every rank contributes the same count to the all-gather; the last rank pads."""
    return (instances + world - 1) // world


def all_gather_rows(local, instances: int, world: int, group=None):
    """All-gather per-instance rows (`local`: [n_local, ...] torch tensor on any device) into
    [instances, ...], dropping the padding of the last rank.  world == 1 without a process group:
    returns `local` (with one, the collective still runs — the single-GPU rehearsal of the N>1 path)."""
    import torch
    import torch.distributed as dist

    if world == 1 and not dist.is_initialized():
        return local

    per = padded_count(instances, world)
    if local.shape[0] != per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:instances]


def all_gather_rows_chunked(tensors, instances: int, world: int, budget_bytes: int = 8 << 30, consume=None, group=None):
    """All-gather several per-instance row arrays (`tensors`: [n_local, W] each, e.g. position / normal /
    tangent) one after the other through ONE bounded buffer: row chunks sized so that the gathered chunk
    stays under `budget_bytes`.  After each collective `consume(k, r0, r1, view)` (if given) sees the chunk:
    view[g, j] = row r0 + j of rank g's shard of tensors[k] (the last rank's missing rows are zero).
    Returns the bytes every rank received."""
    import torch
    import torch.distributed as dist

    per = padded_count(instances, world)
    width = int(tensors[0].shape[1])
    row_bytes = width * tensors[0].element_size()
    rows = max(1, min(max(per, 1), budget_bytes // max(1, world * row_bytes)))
    buf = torch.empty((world * rows, width), dtype=tensors[0].dtype, device=tensors[0].device)
    moved = 0
    for k, t in enumerate(tensors):
        for r0 in range(0, per, rows):
            r1 = min(per, r0 + rows)
            part = t[min(r0, t.shape[0]):min(r1, t.shape[0])]
            if part.shape[0] != r1 - r0:  # a short (last) shard: pad
                pad = torch.zeros((r1 - r0 - part.shape[0], width), dtype=t.dtype, device=t.device)
                part = torch.cat([part, pad], dim=0)
            out = buf[: world * (r1 - r0)]
            dist.all_gather_into_tensor(out, part.contiguous(), group=group)
            moved += world * (r1 - r0) * row_bytes
            if consume is not None:
                consume(k, r0, r1, out.view(world, r1 - r0, width))
    return moved


class OverlappedRowGather:
    """The per-frame exchange step: all-gather of one small per-instance array (the bounds, 24 B per
    instance) that runs BESIDE the next frame instead of in front of it.  On a GPU the producer's stream (the library's,
    passed as a raw hipStream_t and wrapped for torch without a copy) records an event when a frame is enqueued; a side
    stream waits for it, copies the shard's rows into a staging buffer and all-gathers that (RCCL).  The next frame only
    has to wait for the COPY (`before_frame`), never for the collective, and the host never blocks.  On the CPU (gloo,
    the tests) there are no streams and the gather simply runs in `after_frame`.

        ex = OverlappedRowGather(local_rows, instances, world, stream_ptr)
        per frame:  ex.before_frame(); <enqueue the frame>; ex.after_frame()
        ex.result()  ->  [instances, W] rows of the LAST frame (synchronises the side stream)"""

    def __init__(self, local, instances: int, world: int, producer_stream: int = 0, group=None):
        import torch

        self.local, self.instances, self.world, self.group = local, instances, world, group
        per = padded_count(instances, world)
        width = tuple(local.shape[1:])
        self.staging = torch.zeros((per,) + width, dtype=local.dtype, device=local.device)  # a short last shard stays zero-padded
        self.gathered = torch.empty((world * per,) + width, dtype=local.dtype, device=local.device)
        self.on_gpu = local.is_cuda
        self.started = False
        if self.on_gpu:
            self.producer = torch.cuda.ExternalStream(producer_stream, device=local.device) if producer_stream else torch.cuda.current_stream(local.device)
            self.side = torch.cuda.Stream(device=local.device)
            self.ev_frame, self.ev_copy = torch.cuda.Event(), torch.cuda.Event()

    def before_frame(self):
        """the previous frame's rows have been copied out before this frame may rewrite them"""
        if self.on_gpu and self.started:
            self.producer.wait_event(self.ev_copy)

    def after_frame(self):
        import torch
        import torch.distributed as dist

        n = self.local.shape[0]
        if not self.on_gpu:
            self.staging[:n].copy_(self.local)
            dist.all_gather_into_tensor(self.gathered, self.staging, group=self.group)
        else:
            self.ev_frame.record(self.producer)
            self.side.wait_event(self.ev_frame)
            with torch.cuda.stream(self.side):
                self.staging[:n].copy_(self.local, non_blocking=True)
                self.ev_copy.record(self.side)
                dist.all_gather_into_tensor(self.gathered, self.staging, group=self.group)
        self.started = True

    def result(self):
        if self.on_gpu:
            self.side.synchronize()
        return self.gathered[: self.instances]


class DeviceArray:
    """Zero-copy view of a libskinrt device buffer for torch (`torch.as_tensor(DeviceArray(...),
    device='cuda')`), via __cuda_array_interface__."""

    def __init__(self, ptr: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape),
            "typestr": typestr,
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }
