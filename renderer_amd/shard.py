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
