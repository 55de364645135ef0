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


# ---------------------------------------------------------------------------------------------------------------------
# multi-slot backward nodes vs bucket boundaries; sharded AdamW state in checkpoints; rank-aware train_dp
# ---------------------------------------------------------------------------------------------------------------------
def test_grad_sink_multi_slot_node_does_not_flush_its_own_bucket_early():
    """A backward node that writes several parameters with ONE kernel (Routing: Wr, br, bias_e; NormAct: gamma, beta,
    slope) asks for its slots together: a bucket boundary between two of them must not put the first bucket on the wire
    before the node's kernel has been enqueued (GradSink.slots flushes once, then marks)."""
    from coma_unet_amd import ops

    class Listener:
        def __init__(self):
            self.pending, self.sent, self.log = [], [], []

        def mark(self, p):
            self.pending.append(p.tag)

        def flush_pending(self):
            self.sent += self.pending
            self.log.append(list(self.pending))
            self.pending = []

    ps = []
    for tag in ("Wr", "br", "be", "next"):
        p = torch.nn.Parameter(torch.zeros(3))
        p.grad = torch.zeros(3)
        p._coma_sink = True
        p.tag = tag
        ps.append(p)
    lis = Listener()
    ops.GradSink.written.clear()
    ops.GradSink.listener = lis
    try:
        got = ops.GradSink.slots(ps[:3])            # the three-slot node asks BEFORE launching its kernel
        assert all(g is p.grad for g, p in zip(got, ps[:3]))
        assert lis.sent == [] and lis.pending == ["Wr", "br", "be"], "nothing of this node may be flushed by its own request"
        ops.GradSink.slot(ps[3])                    # the next node's request proves the kernel above is enqueued
        assert lis.sent == ["Wr", "br", "be"] and lis.pending == ["next"]
        assert ops.GradSink.slots(ps[:3]) == [None, None, None]      # second use in one step: ordinary returned gradients
    finally:
        ops.GradSink.listener = None
        ops.GradSink.written.clear()


def test_graph_bucket_watch_bookkeeping_and_second_use():
    """data_parallel.GraphBucketWatch without a GPU: buckets cut at parameter boundaries, a bucket becomes pending when its
    last write-through gradient has been announced, and a parameter that is used a SECOND time in the step (autograd will
    accumulate into its slot unannounced) takes its bucket out of the early set -- also when it had already been flushed."""
    from coma_unet_amd import ops
    from coma_unet_amd.optim import FusedAdamW
    from coma_unet_amd.data_parallel import GraphBucketWatch
    ops.adamw_ = _cpu_adamw
    torch.manual_seed(1)
    net = _Net()
    opt = FusedAdamW(net.parameters(), lr=1e-2, write_through=True)
    net(torch.randn(4, 7)).sum().backward()
    opt.step()                                              # builds the flat layout
    w = GraphBucketWatch(opt, 16)                           # 16 elements per bucket: several buckets
    assert len(w.bounds) > 2 and w.bounds[0][0] == 0 and w.bounds[-1][1] == opt.flat_g.numel()
    assert all(a[1] == b[0] for a, b in zip(w.bounds, w.bounds[1:]))
    starts = {opt._offsets[id(p)][0] for p in opt._flat_params}
    assert all(s0 in starts for s0, _ in w.bounds), "buckets are cut at parameter boundaries"
    w._left, w._pending, w.groups = list(w._total), [], []
    ops.GradSink.written.clear()
    ops.GradSink.observer = w
    flushed = []
    w.flush_pending = lambda: (flushed.append(list(w._pending)), w.groups.append((list(w._pending), [])), w._pending.clear()) if w._pending else None
    try:
        params = list(reversed(opt._flat_params))
        for p in params:
            assert ops.GradSink.slot(p) is p.grad
        ops.GradSink.slot(None)
        done = [i for g in flushed for i in g]
        assert sorted(done) == [i for i in range(len(w.bounds)) if w._total[i] > 0], "every bucket of announced parameters completes once"
        # a second use of one parameter: its bucket leaves the early set although it was flushed already
        victim = params[0]
        bi = w._p2b[id(victim)]
        assert any(bi in todo for todo, _ in w.groups)
        assert ops.GradSink.slot(victim) is None
        assert not any(bi in todo for todo, _ in w.groups) and bi not in w._pending
    finally:
        ops.GradSink.observer = None
        ops.GradSink.written.clear()


def _worker_sharded_state(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coma_unet_amd import ops
    from coma_unet_amd.optim import FusedAdamW
    from coma_unet_amd.data_parallel import StreamedGradExchange, TorchComm, broadcast_module
    ops.adamw_ = _cpu_adamw
    torch.manual_seed(7)
    net, ref = _Net(), _Net()
    broadcast_module(net, src=0)
    ref.load_state_dict(net.state_dict())
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    opt = FusedAdamW(net.parameters(), lr=1e-2, write_through=True, pad_to=world)
    ex = StreamedGradExchange(opt, TorchComm(), bucket_bytes=128, sharded=True)
    for step in range(4):
        g = torch.Generator().manual_seed(50 + step)
        xs = [torch.randn((4, 7), generator=g) for _ in range(world)]
        opt.zero_grad()
        ex.begin()
        net(xs[rank]).sum().backward()
        if not ex.finish():
            opt.step()
        ropt.zero_grad()
        ref(torch.cat(xs)).sum().backward()
        ropt.step()
    sd, rsd = opt.state_dict(), ropt.state_dict()          # (a collective here: every rank calls it)
    ok = True
    params, rparams = list(net.parameters()), list(ref.parameters())
    for i, p in enumerate(params):
        if p.grad is None:
            ok &= i not in sd["state"]
            continue
        a, b = sd["state"][i], rsd["state"][i]
        ok &= float(a["step"]) == float(b["step"]) == 4.0
        ok &= bool(torch.allclose(a["exp_avg"], b["exp_avg"], rtol=1e-4, atol=1e-7))
        ok &= bool(torch.allclose(a["exp_avg_sq"], b["exp_avg_sq"], rtol=1e-4, atol=1e-9))
    # resume from that state on every rank: the next step still tracks torch.optim.AdamW
    opt2 = FusedAdamW(net.parameters(), lr=1e-2, write_through=True, pad_to=world)
    opt2.load_state_dict(sd)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gloo_world2_sharded_adamw_state_dict_is_gathered():
    """FusedAdamW.state_dict() under StreamedGradExchange(sharded=True): each rank steps only its sub-slices of the
    moments, so the checkpoint (attn_unet_data_parallel.py:946-952) must all-gather them: compared with torch.optim.AdamW
    on the global batch after 4 steps (1 unsharded + 3 sharded)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded_state, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


class _TinyModel(nn.Module):
    """The attribute surface train_dp touches, on a CPU-sized network (the HIP model cannot run here)."""
    static_prompts = True
    embeddings_out = False

    def __init__(self):
        super().__init__()
        self.net = _Net()

    def set_training(self, mode):
        pass


def _worker_train_dp(rank, world, port, q, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coma_unet_amd import ops, train_loop
    from coma_unet_amd.optim import FusedAdamW
    from coma_unet_amd.data_parallel import GradReducer, broadcast_module
    ops.adamw_ = _cpu_adamw
    torch.manual_seed(3)
    model = _TinyModel()
    broadcast_module(model, src=0)
    opt = FusedAdamW(model.parameters(), lr=1.0)          # train_dp must set lr and use THIS optimizer
    red = GradReducer(opt, bucket_bytes=128)

    class Crit:
        class gen_loss:
            batch_reduction = "mean"
            voxel_wise = True
            roi_indices, roi_weights = [], None

    def fake_step(model_, criterion, optimizer, batch, reducer):
        assert optimizer is opt and reducer is red
        optimizer.zero_grad()
        reducer.reset()
        out = model_.net(batch["mri"])
        # rank-dependent loss scale: the LOCAL epoch losses differ by 10x between the ranks
        loss = (out ** 2).sum() * (10.0 if rank == 1 else 1.0)
        loss.backward()
        reducer.finish()
        optimizer.step()
        return (loss.detach(), loss.detach().reshape(1, 1).expand(batch["mri"].shape[0], 1) / batch["mri"].shape[0],
                torch.zeros(()), torch.zeros(())), (out,)

    train_loop.train_step = fake_step
    seen = []

    class Sched(train_loop.ReduceLROnPlateau):
        def step(self, metrics, *a, **k):
            seen.append(float(metrics))
            return super().step(metrics, *a, **k)

    train_loop.ReduceLROnPlateau = Sched
    g = torch.Generator().manual_seed(100 + rank)
    loader = []
    for i in range(3):
        x = torch.randn((4, 7), generator=g)
        item = (x, x, x, (torch.ones(4), torch.zeros(4, 1, 6)), [f"/d/adni/{rank}-{i}-{j}/s" for j in range(4)])
        loader.append((item, item, item))
    lookup = {f"{rank}-{i}-{j}/s": {} for i in range(3) for j in range(4)}
    losses = train_loop.train_dp(model, Crit, loader, None, 3, 1e-2, save_path=tmp, cuda_id="cpu", roi_vecs_dict=lookup,
                                 reducer=red, graph=False)
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    files = sorted(os.listdir(os.path.join(tmp, "checkpoints"))) if os.path.isdir(os.path.join(tmp, "checkpoints")) else []
    q.put((rank, losses, seen, float(opt.param_groups[0]["lr"]), w.tolist(), red._buckets is not None and len(red._buckets) > 1,
           files, opt._flat_step))
    dist.destroy_process_group()


def test_gloo_world2_train_dp_is_rank_aware(tmp_path):
    """train_dp(reducer=...): steps the reducer's optimizer (so the bucketed layout comes into being), feeds the plateau
    scheduler the GLOBAL batch loss (identical on both ranks although the local losses differ 10x), keeps the replicas
    identical, and only rank 0 writes checkpoints."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tmps = [str(tmp_path / f"rank{r}") for r in range(world)]
    for t in tmps:
        os.makedirs(t)
    procs = [ctx.Process(target=_worker_train_dp, args=(r, world, port, q, tmps[r])) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, l0, s0, lr0, w0, b0, f0, n0), (r1, l1, s1, lr1, w1, b1, f1, n1) = res
    assert l0 == l1 and s0 == s1 and len(s0) == 3, "every rank must hand the scheduler the same (global) epoch loss"
    assert lr0 == lr1 == 1e-2 and n0 == n1 == 9
    assert b0 and b1, "the reducer's bucketed layout must exist: train_dp has to step the reducer's own optimizer"
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(a)) for a, b in zip(w0, w1)), "replicas diverged"
    assert f0 == ["checkpoint_epoch_0.pth", "checkpoint_latest_epoch.pth"] and f1 == []
