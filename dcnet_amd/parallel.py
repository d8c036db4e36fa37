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
    """One flat buffer, one collective: on fully connected xGMI a single large all-reduce (296 MB of fp32 gradients) is
    cheaper than 13 default 25 MB buckets, and it needs no autograd hooks (our backbone returns all its gradients from one
    backward node anyway).  Call after backward().

    ``bind()`` makes every trainable parameter's ``.grad`` a VIEW of the flat buffer: autograd then accumulates straight into it
    (start each step with ``zero()`` instead of ``zero_grad(set_to_none=True)``), the all-reduce runs on the buffer in place and
    the optimiser reads the views — no per-parameter copies (unbound, a step costs ~600 small copy launches).  This is the form
    the graph-replayed data-parallel step uses (dcnet_amd.graph.GraphedTrainStep: the zeroing and the accumulation are part of
    the captured graph, the collective follows the replay).

    A bound parameter ALWAYS has a gradient (zeros when the step produced none), so the optimiser updates it every step —
    weight decay, the running square average, the step counter — where the reference skips a parameter whose ``.grad`` is
    None.  Bind only parameters that receive a gradient every step: run ``freeze_gradless(model)`` first (the dead YOLO
    heads and ``feature_map``); ``check_bound_set(model)`` verifies it (GraphedTrainStep calls it)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, comm_dtype=None):
        """comm_dtype=torch.bfloat16 (bound form): the collective moves a bf16 copy of the buffer — 148 MB instead of 296 MB over
        xGMI (BASELINE configs[2]); gradients stay fp32 on both sides (cast, all-reduce, cast back + average: two passes
        of dcn_cast_rows over the buffer).  The sum over W ranks is then taken in bf16 by RCCL: a reduced-precision choice that
        belongs to the bf16 modes, never to the fp32 run."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.comm_dtype = comm_dtype
        self._comm = None
        self._flat = None
        self.bound = False
        self.always_collective = False   # tests: issue the all-reduce on a one-rank group as well (RCCL behind a graph replay)
        self.collectives = 0             # all-reduces issued so far

    ALIGN = 64           # floats: every view starts on a 256-byte boundary (16-byte vector paths of the kernels that read it)

    def bind(self) -> "FlatGradAllReduce":
        pad = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        n = sum(pad(p.numel()) for p in self.params)
        p0 = self.params[0]
        self._flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        off = 0
        for p in self.params:
            p.grad = self._flat[off:off + p.numel()].view_as(p)
            off += pad(p.numel())
        self.bound = True
        return self

    @staticmethod
    def _cast(src: torch.Tensor, dst: torch.Tensor) -> None:
        if src.is_cuda and src.numel() % 8 == 0:
            from . import ops
            ops.cast_rows(src.view(1, -1), dst.view(1, -1))     # csrc/b16.hip
        else:
            dst.copy_(src)                                       # (CPU tensors: the gloo tests)

    def check_bound_set(self, model) -> None:
        """No parameter that can never receive a gradient (``gradless_parameter_names``: the dead YOLO heads, ``feature_map``) may
        be bound: autograd never writes it, yet its zero view would be decayed by the optimiser every step — the reference skips
        it (``.grad is None``).  (A value test cannot tell: ``loc_text_embedding.0.bias`` sits in front of a BatchNorm and gets
        a gradient of exact zeros from autograd, in the reference too.)"""
        if not self.bound:
            return
        mine = {id(p) for p in self.params}
        table = dict(model.named_parameters())
        bad = [n for n in gradless_parameter_names(model) if id(table[n]) in mine]
        if bad:
            raise RuntimeError(f"FlatGradAllReduce: {len(bad)} bound parameter(s) never receive a gradient ({', '.join(bad[:6])}, ...); "
                               "run parallel.freeze_gradless(model) before bind() — a bound parameter is updated every step")

    def zero(self) -> None:
        """Bound form: zero every gradient with one memset (the views stay attached)."""
        if not self.bound:
            raise RuntimeError("FlatGradAllReduce.zero(): call bind() first")
        self._flat.zero_()

    def __call__(self) -> None:
        world = dist.get_world_size(self.group)
        if self.bound:
            if world > 1 or self.always_collective:
                if self.comm_dtype is not None and self.comm_dtype != self._flat.dtype:
                    if self._comm is None:
                        self._comm = torch.empty(self._flat.numel(), dtype=self.comm_dtype, device=self._flat.device)
                    self._cast(self._flat, self._comm)
                    dist.all_reduce(self._comm, op=dist.ReduceOp.SUM, group=self.group)
                    self._cast(self._comm, self._flat)
                else:
                    dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
                self.collectives += 1
                if world > 1:
                    self._flat.div_(world)
            return
        if world == 1:
            return
        n = sum(p.numel() for p in self.params)
        if self._flat is None or self._flat.numel() != n or self._flat.device != self.params[0].device:
            self._flat = torch.empty(n, dtype=self.params[0].dtype, device=self.params[0].device)
        off = 0
        views = []
        for p in self.params:
            views.append(self._flat[off:off + p.numel()].view_as(p)); off += p.numel()
        have = [(v, p.grad) for v, p in zip(views, self.params) if p.grad is not None]
        for v, p in zip(views, self.params):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.div_(world)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
        if have:
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])


class OverlappedGradReducer:
    """Gradient all-reduce that runs UNDER the backward pass (what the reference gets from DDP's bucket hooks,
    train_DCNet.py:483), for a model whose backbone is one autograd node.

    DDP's hooks fire when autograd hands a gradient to its parameter; our backbone (darknet._DarknetFn) hands over all its
    ~220 gradients at once when its hand-scheduled backward returns, i.e. after the LAST kernel of the step — the all-reduce
    of 160 MB of backbone gradients would start when nothing is left to hide it behind.  Instead the backbone's reverse
    sweep calls ``push`` every time ``bucket_bytes`` of parameter gradients are final (deepest layers first: the 1024-channel
    banks, most of the bytes, are ready while the long early-layer backward is still ahead): the bucket is averaged over the
    ranks on a communication stream, in place, while the data-gradient chain continues.  ``finish`` reduces what autograd
    produced outside the backbone (heads, language branch: one flat bucket) and joins the streams.

    Semantics: every reduced gradient ends up as the mean over ranks, exactly as with DDP / FlatGradAllReduce.  The pushing
    node joins the communication stream before it returns its gradients to autograd, so whatever autograd does with them
    next (adopt, copy) sees averaged values; start each step from ``zero_grad(set_to_none=True)`` — a pushed gradient ADDED
    into an existing ``.grad`` would be averaged while the old content is not (DDP has the same contract under no_sync).
    On CPU tensors (gloo, tests) the same calls run synchronously."""

    uses_parameter_hooks = False      # (reads .grad in finish(): graph.GraphedTrainStep may let the head add its weight gradients directly)

    def __init__(self, model_or_params, bucket_bytes: int = 48 << 20, group=None):
        params = model_or_params.parameters() if hasattr(model_or_params, "parameters") else model_or_params
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_bytes)
        self.group = group
        self._comm = None
        self.always_collective = False   # tests: issue the all-reduce on a one-rank group as well (exercises the RCCL stream ordering)
        self._pushed = set()         # id(param) of the parameters whose gradient was reduced during the backward
        self._bufs = {}
        self.buckets_last_step = 0

    # -- plumbing --------------------------------------------------------------------------------------
    def _world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _stream(self, device):
        if device.type != "cuda":
            return None
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=device)
        return self._comm

    def _buffer(self, key, n, like):
        b = self._bufs.get(key)
        if b is None or b.numel() < n or b.device != like.device:
            b = torch.empty(n, dtype=like.dtype, device=like.device)
            self._bufs[key] = b
        return b[:n]

    def _reduce(self, tensors: Sequence[torch.Tensor], key) -> None:
        """Average ``tensors`` over the ranks in place: pack, one all-reduce, unpack.  On a GPU everything is queued on the
        communication stream, which first waits for the streams that produced the gradients."""
        if not tensors:
            return
        dev = tensors[0].device
        comm = self._stream(dev)
        world = self._world()
        n = sum(t.numel() for t in tensors)
        if comm is not None:
            comm.wait_stream(torch.cuda.current_stream(dev))
            for s in self._producer_streams(dev):
                comm.wait_stream(s)
        ctx = torch.cuda.stream(comm) if comm is not None else _null()
        with ctx:
            flat = self._buffer(key, n, tensors[0])
            views = []
            off = 0
            for t in tensors:
                views.append(flat[off:off + t.numel()].view(t.shape)); off += t.numel()
            torch._foreach_copy_(views, list(tensors))
            if world > 1 or (self.always_collective and dist.is_available() and dist.is_initialized()):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)       # (on a GPU: ordered on `comm` by ProcessGroupNCCL)
                if world > 1:
                    flat.div_(world)
            torch._foreach_copy_(list(tensors), views)
            if comm is not None:
                for t in tensors:
                    t.record_stream(comm)
        self.buckets_last_step += 1

    def _producer_streams(self, device):
        from . import ops
        try:
            return [ops.side_stream(device)] if ops.WGRAD_SIDE else []
        except Exception:
            return []

    # -- the three calls of a step ------------------------------------------------------------------------
    def begin_step(self) -> None:
        self._pushed.clear()
        self.buckets_last_step = 0

    def push(self, pairs: Sequence) -> None:
        """pairs = [(parameter, its final gradient tensor), ...]: called from inside a backward pass as gradients complete."""
        pairs = [(p, g) for p, g in pairs if g is not None and p.requires_grad]
        if not pairs:
            return
        for p, _ in pairs:
            self._pushed.add(id(p))
        self._reduce([g for _, g in pairs], "bucket")

    def join_backward(self, device=None) -> None:
        """The current stream waits for every bucket pushed so far (call before the pushed gradients are read or handed to
        autograd's accumulation)."""
        if self._comm is not None:
            torch.cuda.current_stream(self._comm.device).wait_stream(self._comm)

    def finish(self) -> None:
        """After backward(): reduce the gradients that were not pushed (one flat bucket) and join the streams."""
        rest = [p.grad for p in self.params if id(p) not in self._pushed and p.grad is not None]
        self._reduce(rest, "rest")
        self.join_backward()

    def sync_buffers(self, model, src: int = 0) -> None:
        """BatchNorm running statistics of rank ``src`` to every rank — what DDP(broadcast_buffers=True) leaves behind (rank 0's
        buffers overwrite the others' every forward; train-mode forwards never read them, so once before eval/save is the same)."""
        if self._world() > 1:
            for b in model.buffers():
                dist.broadcast(b.data, src=src, group=self.group)


def attach_overlapped_reducer(model, bucket_bytes: int = 48 << 20, group=None) -> OverlappedGradReducer:
    """Data-parallel setup without the DDP wrapper: parameters of rank 0 to every rank, a reducer over all trainable
    parameters, and the backbone told to push its gradients into it during its backward.  Per step:
    ``red.begin_step(); loss.backward(); red.finish(); optimizer.step()``."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        broadcast_parameters(model, 0, group)
    red = OverlappedGradReducer(model, bucket_bytes, group)
    model.visumodel.__dict__["_grad_reducer"] = red
    return red


class _null:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def broadcast_parameters(model, src: int = 0, group=None) -> None:
    """Initial parameter/buffer broadcast (what DDP does at wrap time, C3 in SURVEY.md)."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
