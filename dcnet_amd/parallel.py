"""Data-parallel plumbing: one process per GPU, clips sharded across ranks, one gradient
all-reduce per step over RCCL/xGMI (backend "nccl" on ROCm).  Mirrors what the reference gets from
``DistributedSampler`` + ``DistributedDataParallel(find_unused_parameters=True)``
(train_DCNet.py:467-483) with a static graph: parameters that can never receive a gradient are
frozen up front, so no unused-parameter search runs per step.

A clip (its T frames + query) never leaves a GPU — co-attention is intra-clip and BatchNorm
statistics are per GPU in the reference (no SyncBN) — so the forward has no collective at all.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def gradless_parameter_names(model) -> List[str]:
    """Parameters of grounding_model that receive no gradient in the reference either: the dead YOLO
    detection heads and ``feature_map`` (consumed only through arg-max, F8)."""
    live = {op.slot for op in model.visumodel._conv_ops}
    names = []
    for i, seq in enumerate(model.visumodel.module_list):
        if i not in live:
            names += [f"visumodel.module_list.{i}.{n}" for n, _ in seq.named_parameters()]
    names += [f"feature_map.{n}" for n, _ in model.feature_map.named_parameters()]
    return names


def freeze_gradless(model) -> List[str]:
    names = gradless_parameter_names(model)
    table = dict(model.named_parameters())
    for n in names:
        table[n].requires_grad_(False)
    return names


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """DistributedSampler-style partition without shuffling: rank r takes items r, r+W, ... after
    padding the index list by wrap-around to a multiple of W (torch DistributedSampler, drop_last=False)."""
    total = (n_items + world - 1) // world * world
    idx = list(range(n_items))
    pad = total - n_items
    if pad:
        idx += (idx * ((pad + n_items - 1) // n_items))[:pad]
    return idx[rank:total:world]


def wrap_ddp(model, local_rank: int):
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=[local_rank] if torch.cuda.is_available() else None,
        broadcast_buffers=True, gradient_as_bucket_view=True)


class FlatGradAllReduce:
    """One flat buffer, one collective: on fully connected xGMI a single large all-reduce (323 MB of
    fp32 gradients) is cheaper than 13 default 25 MB buckets, and it needs no autograd hooks (our
    backbone returns all its gradients from one backward node anyway).  Call after backward()."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self._flat = None

    def __call__(self) -> None:
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        n = sum(p.numel() for p in self.params)
        if self._flat is None or self._flat.numel() != n or self._flat.device != self.params[0].device:
            self._flat = torch.empty(n, dtype=self.params[0].dtype, device=self.params[0].device)
        off = 0
        views = []
        for p in self.params:
            v = self._flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            views.append(v); off += p.numel()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.div_(world)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def broadcast_parameters(model, src: int = 0, group=None) -> None:
    """Initial parameter/buffer broadcast (what DDP does at wrap time, C3 in SURVEY.md)."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
