"""CPU, world_size 2, gloo: the data-parallel path (parameter broadcast, clip sharding, one flat
gradient all-reduce, DDP wrapper) produces on every rank the mean of the per-rank gradients, and the
frozen (gradient-less) parameters stay out of the collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16); self.b = torch.nn.Linear(16, 4); self.dead = torch.nn.Linear(3, 3)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


class _ManyGrads(torch.autograd.Function):
    """The shape of the real backbone (dcnet_amd.darknet._DarknetFn): ONE autograd node that takes every parameter of a
    deep stack and hands back all their gradients at once when its backward returns."""

    @staticmethod
    def forward(ctx, x, *ws):
        acts = [x]
        for w in ws:
            acts.append(torch.tanh(acts[-1] @ w))
        ctx.save_for_backward(*acts, *ws)
        ctx.n = len(ws)
        return acts[-1]

    @staticmethod
    def backward(ctx, g):
        sv = ctx.saved_tensors
        acts, ws = sv[:ctx.n + 1], sv[ctx.n + 1:]
        grads = [None] * ctx.n
        for i in range(ctx.n - 1, -1, -1):
            g = g * (1 - acts[i + 1] ** 2)
            grads[i] = acts[i].t() @ g if ctx.needs_input_grad[1 + i] else None
            g = g @ ws[i].t()
        return (g,) + tuple(grads)


class _Backbone(torch.nn.Module):
    """40 layers behind one autograd node, 6 of them frozen (the dead YOLO heads of the real model), plus a head that
    autograd differentiates the ordinary way."""

    def __init__(self, depth=40, width=12):
        super().__init__()
        self.ws = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(width, width) * 0.3) for _ in range(depth)])
        self.dead = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(5)) for _ in range(6)])
        self.head = torch.nn.Linear(width, 3)

    def forward(self, x):
        return self.head(_ManyGrads.apply(x, *self.ws))


def _worker_backbone(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcnet_amd.parallel import FlatGradAllReduce, broadcast_parameters, shard_indices, wrap_ddp
    torch.manual_seed(5 + rank)
    m = _Backbone()
    for i in (3, 17):                                # frozen layers INSIDE the single node as well
        m.ws[i].requires_grad_(False)
    for p in m.dead:
        p.requires_grad_(False)
    broadcast_parameters(m, src=0)
    g = torch.Generator().manual_seed(11)
    data = torch.randn(16, 12, generator=g); tgt = torch.randn(16, 3, generator=g)
    idx = shard_indices(16, rank, world)
    res = {}
    for it in range(2):                              # two steps: DDP rebuilds its buckets after the first
        m.zero_grad(set_to_none=True)
        ((m(data[idx]) - tgt[idx]) ** 2).mean().backward()
        res[f"local{it}"] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        FlatGradAllReduce(m.parameters())()
        res[f"flat{it}"] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    # bound form (the graph-replayed data-parallel step): gradients are views of ONE flat buffer, zeroed by one memset, reduced in place
    m3 = _Backbone(); m3.load_state_dict(m.state_dict())
    for i in (3, 17):
        m3.ws[i].requires_grad_(False)
    for p in m3.dead:
        p.requires_grad_(False)
    fl = FlatGradAllReduce(m3.parameters()).bind()
    ptrs = {k: p.grad.data_ptr() for k, p in m3.named_parameters() if p.grad is not None}
    for it in range(2):
        fl.zero()
        ((m3(data[idx]) - tgt[idx]) ** 2).mean().backward()
        fl()
        res[f"bound{it}"] = {k: p.grad.clone() for k, p in m3.named_parameters() if p.grad is not None}
    res["bound_views_kept"] = all(p.grad.data_ptr() == ptrs[k] for k, p in m3.named_parameters() if k in ptrs)
    # the bf16 modes move a bf16 copy of the flat buffer (half the xGMI bytes); gradients stay fp32 on both sides
    m4 = _Backbone(); m4.load_state_dict(m.state_dict())
    for i in (3, 17):
        m4.ws[i].requires_grad_(False)
    for p in m4.dead:
        p.requires_grad_(False)
    fb = FlatGradAllReduce(m4.parameters(), comm_dtype=torch.bfloat16).bind()
    fb.zero()
    ((m4(data[idx]) - tgt[idx]) ** 2).mean().backward()
    fb()
    res["bound_bf16"] = {k: p.grad.clone() for k, p in m4.named_parameters() if p.grad is not None}
    res["bound_bf16_dtype"] = (fb._comm.dtype, next(iter(res["bound_bf16"].values())).dtype)
    m2 = _Backbone(); m2.load_state_dict(m.state_dict())
    for i in (3, 17):
        m2.ws[i].requires_grad_(False)
    for p in m2.dead:
        p.requires_grad_(False)
    d = wrap_ddp(m2, 0)                              # static graph: no find_unused_parameters
    for it in range(2):
        d.zero_grad(set_to_none=True)
        ((d(data[idx]) - tgt[idx]) ** 2).mean().backward()
        res[f"ddp{it}"] = {k: p.grad.clone() for k, p in m2.named_parameters() if p.grad is not None}
    res["frozen_grads"] = [m.ws[3].grad, m.dead[0].grad, m2.ws[17].grad]
    torch.save(res, os.path.join(out, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_allreduce_with_a_single_node_backbone(tmp_path):
    """The real model hands 200 gradients to the reducer from ONE autograd node and keeps 20 parameters frozen: both the
    flat all-reduce and the DDP wrapper must give the mean of the rank gradients on that shape, two steps in a row."""
    world, port = 2, _free_port()
    mp.spawn(_worker_backbone, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"b{i}.pt")) for i in range(world)]
    for it in range(2):
        keys = list(r[0][f"local{it}"].keys())
        assert len(keys) == 38 + 2 and not any(".3" == k[-2:] and k.startswith("ws") for k in keys)
        for k in keys:
            mean = (r[0][f"local{it}"][k] + r[1][f"local{it}"][k]) / 2
            for i in range(world):
                assert torch.allclose(r[i][f"flat{it}"][k], mean, atol=1e-6), (it, k)
                assert torch.allclose(r[i][f"ddp{it}"][k], mean, atol=1e-6), (it, k)
                assert torch.allclose(r[i][f"bound{it}"][k], mean, atol=1e-6), (it, k)      # (same parameters: the model state is not stepped)
    assert all(gr is None for gr in r[0]["frozen_grads"])
    assert r[0]["bound_views_kept"] and r[1]["bound_views_kept"], "autograd replaced a gradient view of the flat buffer"
    # bf16 on the wire: each rank's gradient rounded to bf16, summed, rounded again — the mean to ~2^-8 relative, identical on both ranks
    assert r[0]["bound_bf16_dtype"] == (torch.bfloat16, torch.float32)
    for k in r[0]["local0"]:
        mean = (r[0]["local0"][k] + r[1]["local0"][k]) / 2
        tol = 2.0 ** -7 * max(float(r[0]["local0"][k].abs().max()), float(r[1]["local0"][k].abs().max())) + 1e-8
        assert float((r[0]["bound_bf16"][k] - mean).abs().max()) <= tol, k
        assert torch.equal(r[0]["bound_bf16"][k], r[1]["bound_bf16"][k]), k


class _PushingGrads(torch.autograd.Function):
    """_ManyGrads whose backward hands finished gradients to a reducer in groups while it is still running — the shape of
    darknet._DarknetFn with parallel.OverlappedGradReducer attached (sink called per bucket, join before returning)."""

    @staticmethod
    def forward(ctx, red, params, x, *ws):
        acts = [x]
        for w in ws:
            acts.append(torch.tanh(acts[-1] @ w))
        ctx.save_for_backward(*acts, *ws)
        ctx.n, ctx.red, ctx.params = len(ws), red, params
        return acts[-1]

    @staticmethod
    def backward(ctx, g):
        sv = ctx.saved_tensors
        acts, ws = sv[:ctx.n + 1], sv[ctx.n + 1:]
        grads = [None] * ctx.n
        pending = []
        for i in range(ctx.n - 1, -1, -1):
            g = g * (1 - acts[i + 1] ** 2)
            if ctx.needs_input_grad[3 + i]:
                grads[i] = acts[i].t() @ g
                pending.append((ctx.params[i], grads[i]))
            if len(pending) >= 7:                      # a bucket is full: its all-reduce starts now
                ctx.red.push(pending); pending = []
            g = g @ ws[i].t()
        ctx.red.push(pending)
        ctx.red.join_backward()
        return (None, None, g) + tuple(grads)


def _worker_overlap(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcnet_amd.parallel import OverlappedGradReducer, broadcast_parameters, shard_indices
    torch.manual_seed(21 + rank)
    m = _Backbone()
    for i in (3, 17):
        m.ws[i].requires_grad_(False)
    for p in m.dead:
        p.requires_grad_(False)
    broadcast_parameters(m, src=0)
    g = torch.Generator().manual_seed(11)
    data = torch.randn(16, 12, generator=g); tgt = torch.randn(16, 3, generator=g)
    idx = shard_indices(16, rank, world)
    red = OverlappedGradReducer(m)
    res = {}
    for it in range(2):
        # reference: plain local gradients
        m.zero_grad(set_to_none=True)
        ((m(data[idx]) - tgt[idx]) ** 2).mean().backward()
        res[f"local{it}"] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        # the same step with buckets pushed from inside the backbone's backward and the head reduced by finish()
        m.zero_grad(set_to_none=True)
        red.begin_step()
        y = m.head(_PushingGrads.apply(red, list(m.ws), data[idx], *m.ws))
        ((y - tgt[idx]) ** 2).mean().backward()
        res[f"pushed_before_finish{it}"] = len(red._pushed)
        red.finish()
        res[f"red{it}"] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        res[f"buckets{it}"] = red.buckets_last_step
    torch.save(res, os.path.join(out, f"o{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_overlapped_reducer_pushes_buckets_from_inside_the_backbone_backward(tmp_path):
    """parallel.OverlappedGradReducer: the backbone node pushes buckets of finished gradients while its backward runs, the
    head's gradients go in one bucket at finish(); every gradient ends up as the mean over the two ranks, frozen parameters
    stay out, two steps in a row."""
    world, port = 2, _free_port()
    mp.spawn(_worker_overlap, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"o{i}.pt")) for i in range(world)]
    for it in range(2):
        keys = list(r[0][f"local{it}"].keys())
        assert len(keys) == 38 + 2
        assert r[0][f"pushed_before_finish{it}"] == 38            # all live backbone layers were reduced during the backward
        assert r[0][f"buckets{it}"] == 38 // 7 + 1 + 1            # 5 full buckets + the remainder, + the head's flat bucket
        for k in keys:
            mean = (r[0][f"local{it}"][k] + r[1][f"local{it}"][k]) / 2
            for i in range(world):
                assert torch.allclose(r[i][f"red{it}"][k], mean, atol=1e-6), (it, k)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcnet_amd.parallel import FlatGradAllReduce, broadcast_parameters, shard_indices, wrap_ddp
    torch.manual_seed(100 + rank)                  # different init per rank: the broadcast must fix it
    m = _Tiny()
    for p in m.dead.parameters():
        p.requires_grad_(False)
    broadcast_parameters(m, src=0)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(12, 8, generator=g); tgt = torch.randn(12, 4, generator=g)
    idx = shard_indices(12, rank, world)
    # (1) flat all-reduce
    loss = ((m(data[idx]) - tgt[idx]) ** 2).mean(); loss.backward()
    local = [p.grad.clone() for p in m.parameters() if p.requires_grad]
    FlatGradAllReduce(m.parameters())()
    flat = [p.grad.clone() for p in m.parameters() if p.requires_grad]
    # (2) DDP wrapper on the same shard
    m2 = _Tiny(); m2.load_state_dict(m.state_dict())
    for p in m2.dead.parameters():
        p.requires_grad_(False)
    d = wrap_ddp(m2, 0)
    ((d(data[idx]) - tgt[idx]) ** 2).mean().backward()
    ddp = [p.grad.clone() for p in m2.parameters() if p.requires_grad]
    torch.save(dict(local=local, flat=flat, ddp=ddp, w=[p.detach().clone() for p in m.parameters()],
                    dead_grad=m.dead.weight.grad), os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"r{i}.pt")) for i in range(world)]
    for a, b in zip(r[0]["w"], r[1]["w"]):
        assert torch.equal(a, b)                                        # broadcast made the replicas identical
    for k in range(len(r[0]["local"])):
        mean = (r[0]["local"][k] + r[1]["local"][k]) / 2
        for i in range(world):
            assert torch.allclose(r[i]["flat"][k], mean, atol=1e-6)     # flat all-reduce == mean of rank grads
            assert torch.allclose(r[i]["ddp"][k], mean, atol=1e-6)      # and so is DDP's bucketed all-reduce
    assert r[0]["dead_grad"] is None
