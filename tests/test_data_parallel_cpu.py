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


# ---------------------------------------------------------------------------------------------------------------------
# StreamedGradExchange (the sink-driven, capturable exchange of the C-ABI RCCL path) with the torch.distributed stand-in
# communicator, plain and sharded; global-batch RnC through gather_batch
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_adamw(p, g, m, v, lr, b1, b2, eps, wd, step, step_dev=None):
    """torch.optim.AdamW's update on flat views (the HIP kernel's arithmetic, for the CPU-only test)."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt().add_(eps), value=-lr)


def _worker_streamed(rank, world, port, q, sharded):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coma_unet_amd import ops
    from coma_unet_amd.optim import FusedAdamW
    from coma_unet_amd.data_parallel import StreamedGradExchange, TorchComm, broadcast_module
    ops.adamw_ = _cpu_adamw                      # (no HIP kernel on the CPU box)
    torch.manual_seed(7)
    net, ref = _Net(), _Net()
    broadcast_module(net, src=0)
    ref.load_state_dict(net.state_dict())
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    opt = FusedAdamW(net.parameters(), lr=1e-2, write_through=True, pad_to=world)
    ex = StreamedGradExchange(opt, TorchComm(), bucket_bytes=128, sharded=sharded)
    ok = True
    for step in range(5):
        g = torch.Generator().manual_seed(50 + step)
        xs = [torch.randn((4, 7), generator=g) for _ in range(world)]
        opt.zero_grad()
        ex.begin()
        net(xs[rank]).sum().backward()
        if step >= 1:
            # stand-in for the backward kernels' write-through: every flat parameter's slot is announced in reverse
            # order, so buckets complete and go out while "backward" is still running
            for p in reversed(opt._flat_params):
                ops.GradSink.slot(p)
            ops.GradSink.slot(None)
            ok &= ex._buckets is not None and len(ex._buckets) > 1 and any(ex._launched)
        stepped = ex.finish()
        if not stepped:
            opt.step()
        ok &= (stepped is True) == (sharded and step >= 1)
        ropt.zero_grad()
        ref(torch.cat(xs)).sum().backward()
        ropt.step()
        for (n, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
            ok &= bool(torch.allclose(a, b, rtol=1e-4, atol=1e-6))
    if sharded:
        ok &= opt.flat_p.numel() % world == 0
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [False, True])
def test_gloo_world2_streamed_exchange_matches_global_batch_adamw(sharded):
    """5 AdamW steps of 2 replicas (bucketed exchange launched from the gradient sink; sharded: reduce-scatter + AdamW on
    the rank's slices + parameter all-gather) must track torch.optim.AdamW on the global batch."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_streamed, args=(r, world, port, q, sharded)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _worker_flat_step(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coma_unet_amd import ops
    from coma_unet_amd.optim import FusedAdamW
    from coma_unet_amd.data_parallel import GradReducer, broadcast_module
    ops.adamw_ = _cpu_adamw
    torch.manual_seed(11)
    net, ref = _Net(), _Net()
    broadcast_module(net, src=0)
    ref.load_state_dict(net.state_dict())
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    opt = FusedAdamW(net.parameters(), lr=1e-2)
    red = GradReducer(opt, bucket_bytes=96)        # several buckets: the step is taken bucket by bucket
    ok = True
    for step in range(5):
        g = torch.Generator().manual_seed(80 + step)
        xs = [torch.randn((4, 7), generator=g) for _ in range(world)]
        opt.zero_grad()
        red.reset()
        net(xs[rank]).sum().backward()
        if step == 0:
            red.finish()
            opt.step()                             # builds the flat layout
            red.overlap = False
            red.remove_hooks()                     # what train.GraphedTrainStep does: hooks do not fire under replay
        else:
            before = opt._flat_step
            red.reduce_flat_and_step()
            ok &= opt._flat_step == before + 1 and len(range(0, opt.flat_g.numel(), red.bucket_elems)) > 2
        ropt.zero_grad()
        ref(torch.cat(xs)).sum().backward()
        ropt.step()
        for (n, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
            ok &= bool(torch.allclose(a, b, rtol=1e-4, atol=1e-6))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gloo_world2_pipelined_allreduce_and_adamw_matches_global_batch():
    """GradReducer.reduce_flat_and_step (bench.py's default N > 1 mode after the graph-replayed backward): bucket-wise
    all-reduce with each bucket's AdamW slice taken as soon as its exchange is done == torch.optim.AdamW on the global batch."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_flat_step, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _worker_rnc(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coma_unet_amd.criterions import RnCLoss
    from coma_unet_amd.data_parallel import gather_batch
    g = torch.Generator().manual_seed(3)
    feats = torch.randn((6, 16), generator=g)
    labels = torch.rand((6, 6), generator=g)
    w = torch.randn((16, 16), generator=g).requires_grad_(True)       # a "model" shared by the replicas
    mine = slice(rank * 3, rank * 3 + 3)
    f_local = feats[mine] @ w
    loss = RnCLoss()(gather_batch(f_local), gather_batch(labels[mine].contiguous()))
    loss.backward()
    dist.all_reduce(w.grad, op=dist.ReduceOp.SUM)                       # the data-parallel SUM of parameter gradients
    w2 = w.detach().clone().requires_grad_(True)
    ref = RnCLoss()(feats @ w2, labels)
    ref.backward()
    ok = torch.allclose(loss, ref, rtol=1e-5, atol=1e-6) and torch.allclose(w.grad, w2.grad, rtol=1e-4, atol=1e-6)
    local_only = RnCLoss()(f_local.detach(), labels[mine])
    q.put((rank, bool(ok), float(loss), float(local_only)))
    dist.destroy_process_group()


def test_gloo_world2_global_batch_rnc_equals_single_process():
    """forward_loss(..., global_rnc=True): RnC on the all-gathered features equals the single-process loss on the global
    batch, and the SUM-reduced parameter gradient equals the single-process gradient (per-replica RnC does not)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_rnc, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert abs(res[0][2] - res[0][3]) > 1e-3       # the per-replica value is a different number: the gather matters
