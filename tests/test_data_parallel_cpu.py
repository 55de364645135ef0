"""Multi-process data-parallel logic on CPU (gloo, world_size 2): the flat-gradient layout of
FusedAdamW + the bucketed SUM all-reduce of GradReducer (first-step flattened path, second-step
install path and the hook-driven overlap path).  The HIP AdamW kernel itself is not called here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(7, 13)
        self.b = nn.Linear(13, 5)
        self.unused = nn.Parameter(torch.ones(3))      # never receives a gradient (like `reweigh` upstream)
        self.c = nn.Linear(5, 1)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coma_unet_amd.optim import FusedAdamW
    from coma_unet_amd.data_parallel import GradReducer, broadcast_module
    torch.manual_seed(100 + rank)            # different init per rank on purpose
    net = _Net()
    broadcast_module(net, src=0)
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    opt = FusedAdamW(net.parameters(), lr=1e-3)
    red = GradReducer(opt, bucket_bytes=256)  # tiny buckets -> several all-reduces
    ok = True
    for step in range(4):
        g = torch.Generator().manual_seed(step)
        xs = [torch.randn((4, 7), generator=g) for _ in range(world)]     # every rank knows every shard
        opt.zero_grad()
        red.reset()
        # the reference sums per-sample losses (criterions.py:560) -> replica gradients ADD
        net(xs[rank]).sum().backward()
        red.finish()
        got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        # single-process reference on the global batch
        ref = _Net()
        ref.load_state_dict(net.state_dict())
        ref(torch.cat(xs)).sum().backward()
        for n, p in ref.named_parameters():
            if p.grad is None:
                ok &= n not in got
            else:
                ok &= torch.allclose(got[n], p.grad, rtol=1e-5, atol=1e-6)
        if not opt.built:
            opt._build()                      # what step() does first (no HIP kernel on CPU)
        ok &= net.unused.grad is None
        if step >= 1:
            ok &= all(p.grad.data_ptr() >= opt.flat_g.data_ptr() for p in opt._flat_params)
        if step >= 2:
            ok &= red._buckets is not None and len(red._buckets) > 1 and len(red._hooks) == len(opt._flat_params)
    w1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    q.put((rank, bool(ok), w0.tolist(), w1.tolist()))
    dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce_matches_global_batch():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2], "broadcast_module must make the replicas identical"
