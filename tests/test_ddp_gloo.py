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
