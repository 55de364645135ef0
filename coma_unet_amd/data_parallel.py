"""Explicit per-GPU replicas + gradient all-reduce over RCCL (replaces the
``torch.nn.DataParallel`` the reference imports but never applies,
attn_unet_data_parallel.py:32,1554; validation.py:268-269 -- SURVEY.md F4).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Every rank
holds a full replica with identical initial weights (broadcast once), runs forward /
backward on its own samples, and the flat gradient buffer of ``FusedAdamW`` is SUM-reduced
in contiguous buckets.  SUM, not mean: the reference reduces the per-sample loss vector with
``torch.sum`` (criterions.py:560), so for the per-sample generative term (RoiMSE) a global
batch's gradient is the sum of the replicas'.  The RnC term (criterions.py:607-644) couples the
samples of a batch and is evaluated PER REPLICA on its local samples: at the
reference's 2 volumes per device it is identically zero (one off-diagonal logit), so under
N-GPU data parallelism at B=2 it contributes nothing, whereas one device holding all 2N samples would
train it (and so would an applied ``nn.DataParallel``, which gathers the outputs and evaluates the loss
on device 0).  ``gather_batch`` below is the opt-in remedy: an autograd-aware all-gather of the (B, 512)
features and (B, 6) labels, so that every rank evaluates RnC on the global batch and backward hands each
rank the gradient of its own rows (``forward_loss(..., global_rnc=True)``; eager steps only).  Default off,
as SURVEY.md section 8(e) prescribes; stated in DESIGN.md section 5.

Overlap: each bucket's all-reduce is issued from a post-accumulate-grad hook as soon as the
last gradient of that bucket has been produced by backward; RCCL runs it on the process
group's own HIP stream while the compute stream continues with the rest of backward, and
``finish()`` makes the compute stream wait before the optimizer reads the buffer.

BatchNorm statistics and running buffers stay per replica (what an unsynchronised
DataParallel would have done; ``SyncBatchNorm`` is imported upstream but never used).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class _GatherBatch(torch.autograd.Function):
    """cat over ranks along dim 0; backward: this rank's rows of the gradient (every rank evaluates the same
    function of the gathered tensor, so the per-rank gradients of it are identical and need no exchange)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        ctx.rank = dist.get_rank(group)
        ctx.rows = x.shape[0]
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows], None


def gather_batch(x, group=None):
    """(B, ...) per rank -> (world*B, ...) on every rank, differentiable.  Every rank then computes the SAME global
    loss term T(F); rank r's backward sees dT/dF[rows of r], and the data-parallel SUM of the parameter gradients
    over ranks assembles sum_r dT/dF_r dF_r/dtheta = dT/dtheta, the single-device gradient (weight 1, no rescale)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    return _GatherBatch.apply(x, group)


def broadcast_module(module, src=0, group=None):
    """Identical initial replicas: parameters and buffers of rank `src` to everyone."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class GradReducer:
    def __init__(self, optimizer, bucket_bytes=64 << 20, group=None, overlap=True):
        self.opt = optimizer
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        if overlap and hasattr(optimizer, "set_write_through"):
            optimizer.set_write_through(False)     # hook-driven overlap needs autograd's AccumulateGrad to run
        self._hooks = []
        self._works = []
        self._buckets = None      # list of [start, end, n_params_pending, n_params_total]
        self._p2b = {}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    # -- first step: no flat layout yet -> one flattened all-reduce of whatever has a gradient
    def _reduce_unbuilt(self, params):
        gs = [p.grad for p in params if p.grad is not None]
        if not gs:
            return
        flat = torch.cat([g.reshape(-1).float() for g in gs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for g in gs:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def _install(self):
        opt = self.opt
        n = opt.flat_g.numel()
        bounds, start = [], 0
        # cut the flat buffer at parameter boundaries into ~bucket_elems pieces
        cur = 0
        for p in opt._flat_params:
            off, k = opt._offsets[id(p)]
            if off + k - start >= self.bucket_elems:
                bounds.append((start, off + k))
                start = off + k
        if start < n:
            bounds.append((start, n))
        self._buckets = [[s, e, 0, 0] for s, e in bounds]
        bi = 0
        for p in opt._flat_params:
            off, k = opt._offsets[id(p)]
            while off >= self._buckets[bi][1]:
                bi += 1
            self._p2b[id(p)] = bi
            self._buckets[bi][3] += 1
            if self.overlap:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.reset()

    def _make_hook(self, bi):
        def hook(_p):
            b = self._buckets[bi]
            b[2] -= 1
            if b[2] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        s, e = self._buckets[bi][0], self._buckets[bi][1]
        w = dist.all_reduce(self.opt.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)
        self._buckets[bi][2] = -1   # launched

    def reset(self):
        """Call before each backward."""
        self._works = []
        if self._buckets is not None:
            for b in self._buckets:
                b[2] = b[3]

    def reduce_flat(self):
        """All buckets of the flat gradient buffer at once (no hooks): used when forward+backward are replayed
        from a hipGraph, where per-parameter hooks do not fire.  Requires the optimizer's flat layout."""
        if self.world == 1:
            return
        opt = self.opt
        assert opt.built, "reduce_flat needs the flat gradient layout (run two eager steps first)"
        n = opt.flat_g.numel()
        works = [dist.all_reduce(opt.flat_g[s:min(s + self.bucket_elems, n)], op=dist.ReduceOp.SUM, group=self.group,
                                 async_op=True) for s in range(0, n, self.bucket_elems)]
        for w in works:
            w.wait()
        for p in opt.param_groups[0]["params"]:
            if id(p) not in opt._flat_ids and p.grad is not None:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)

    def graph_watch(self, device):
        """A GraphBucketWatch for a step about to be captured, or None (one rank, switched off by COMA_DP_GRAPH_OVERLAP=0,
        or the machine's external events do not behave: ``self.watch_probe`` says which)."""
        import os
        self.watch_probe = "off"
        if self.world == 1 and os.environ.get("COMA_DP_GRAPH_OVERLAP") != "force":
            return None
        if os.environ.get("COMA_DP_GRAPH_OVERLAP", "1") == "0" or not self.opt.built:
            return None
        self.watch_probe = GraphBucketWatch.probe(device)
        if self.watch_probe != "node":
            return None
        # finer buckets than the exchange behind the graph uses: a bucket is sendable when its LAST gradient exists, so a
        # 64 MB bucket that reaches into the first encoder block waits for the end of the backward with all its bytes
        mb = float(os.environ.get("COMA_DP_WATCH_BUCKET_MB", "16"))
        return GraphBucketWatch(self.opt, max(1, min(self.bucket_elems, int(mb * (1 << 20)) // 4)))

    def reduce_flat_and_step(self, watch=None):
        """reduce_flat + optimizer.step(), pipelined: every bucket's all-reduce is queued up front (they run one after the
        other on the process group's stream) and each bucket's slice of the AdamW step is launched as soon as ITS all-reduce
        has finished -- the optimizer's 0.9 ms run under the remaining buckets' exchange instead of after it.
        watch (GraphBucketWatch of the graph just replayed): the buckets it holds events for are queued on an auxiliary
        stream behind their event, i.e. while the replayed backward is still running; the rest behind the graph."""
        opt = self.opt
        if self.world == 1:
            opt.step()
            return
        assert opt.built, "reduce_flat_and_step needs the flat gradient layout (run two eager steps first)"
        n = opt.flat_g.numel()
        if watch is not None:
            if getattr(self, "_aux", None) is None:
                self._aux = torch.cuda.Stream()
            spans, works, done = [], [], set()
            for todo, evs in watch.groups:
                watch.wait(evs, self._aux)
                with torch.cuda.stream(self._aux):
                    for i in todo:
                        s0, e0 = watch.bounds[i]
                        spans.append((s0, e0 - s0))
                        works.append(dist.all_reduce(opt.flat_g[s0:e0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                        done.add(i)
            self.early = len(done)
            self.early_mb = round(sum(watch.bounds[i][1] - watch.bounds[i][0] for i in done) * 4 / 2 ** 20, 1)
            for i, (s0, e0) in enumerate(watch.bounds):
                if i not in done:
                    spans.append((s0, e0 - s0))
                    works.append(dist.all_reduce(opt.flat_g[s0:e0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            spans = [(s, min(self.bucket_elems, n - s)) for s in range(0, n, self.bucket_elems)]
            works = [dist.all_reduce(opt.flat_g[s:s + k], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for s, k in spans]
        loose = [p for p in opt.param_groups[0]["params"] if id(p) not in opt._flat_ids and p.grad is not None]
        lworks = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for p in loose]
        for i, ((s, k), w) in enumerate(zip(spans, works)):
            w.wait()
            opt.step_shards([(s, k)], advance=i == 0, loose=False)
        for w in lworks:
            w.wait()
        opt.step_shards([], advance=False, loose=True)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def finish(self):
        """Call after backward, before optimizer.step()."""
        if self.world == 1:
            return
        opt = self.opt
        params = opt.param_groups[0]["params"]
        if not opt.built:
            self._reduce_unbuilt(params)
            return
        if self._buckets is None:
            # layout was built by the previous step(); grads of THIS backward went straight into flat_g
            self._install()
            for bi in range(len(self._buckets)):
                self._launch(bi)
        else:
            for bi, b in enumerate(self._buckets):
                if b[2] != -1:          # a parameter's hook did not fire (or overlap is off)
                    self._launch(bi)
        for w in self._works:
            w.wait()
        self._works = []
        # tensors outside the flat buffer (the two selectable prompts)
        rest = [p for p in params if id(p) not in opt._flat_ids and p.grad is not None]
        for p in rest:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)


class ExternalEvent:
    """A HIP event recorded as an event-record NODE of the graph being captured and waited for by streams outside the graph
    (coma_event_record_external / coma_stream_wait_external of the C ABI; this PyTorch build refuses
    ``torch.cuda.Event(external=True)`` on ROCm)."""
    __slots__ = ("h",)

    def __init__(self):
        import ctypes
        from ._lib import check, lib
        h = ctypes.c_void_p()
        check(lib.coma_event_create(ctypes.byref(h)), "coma_event_create")
        self.h = h.value

    def record(self, stream=None):
        from ._lib import check, lib
        st = stream if stream is not None else torch.cuda.current_stream()
        check(lib.coma_event_record_external(self.h, st.cuda_stream), "coma_event_record_external")

    def wait(self, stream):
        from ._lib import check, lib
        check(lib.coma_stream_wait_external(stream.cuda_stream, self.h), "coma_stream_wait_external")

    def __del__(self):
        try:
            from ._lib import lib
            if self.h:
                lib.coma_event_destroy(self.h)
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


class GraphBucketWatch:
    """Gradient exchange of the graph-replayed step overlapped with its backward WITHOUT a collective inside the graph.
    While forward + backward are being captured this object listens to the gradient sink (``ops.GradSink.observer``): when
    the last write-through gradient of a bucket of the flat buffer has been enqueued, an EXTERNAL event
    (``ExternalEvent``: ``hipGraphAddEventRecordNode``, an event-record node of the graph) is recorded
    behind it.  After every ``graph.replay()`` the host makes an auxiliary stream wait for those events
    and queues the buckets' ``torch.distributed`` all-reduces there: they run beside the rest of the replayed backward, the
    same way hook-driven buckets do in an eager step.  Buckets that hold a gradient autograd accumulates (not written
    through) or that complete with the very last kernels have no event and go out behind the graph.

    The gradient kernels run on several streams of the captured step (``ops.WgradSide``'s side stream, the preparation
    streams the projection heads were queued on): a group of buckets gets ONE EVENT PER STREAM, each recorded in its own
    stream's order with no edge between the streams, and the exchange waits for all of them.  (One node that depends on
    several streams is seen from outside only when the whole graph has finished: the runtime enqueues a graph level by
    level and makes such a node wait for whatever its parents' queues hold by then, which on the shorter side branch is
    everything -- measured, profiles/external_event_probe.py and dp_watch_timeline.py.  A node with a single parent in its
    own chain is seen when that parent is done.)  Nothing here adds a dependency to a kernel or disables the two-stream step.

    ``probe()`` checks on the running machine that a stream wait issued after ``hipGraphLaunch`` really waits for the
    event-record node of that launch (and not for the whole graph, and not for nothing): the overlap is used only where it
    reports "node"."""

    _probe = {}

    def __init__(self, optimizer, bucket_elems):
        opt = self.opt = optimizer
        assert opt.built, "the flat gradient layout exists after the first optimizer step"
        n = opt.flat_g.numel()
        bounds, start = [], 0
        for p in opt._flat_params:
            off, k = opt._offsets[id(p)]
            if off + k - start >= bucket_elems:
                bounds.append((start, off + k))
                start = off + k
        if start < n:
            bounds.append((start, n))
        self.bounds = bounds
        self._p2b, self._total = {}, [0] * len(bounds)
        bi = 0
        for p in opt._flat_params:
            off, k = opt._offsets[id(p)]
            while off >= bounds[bi][1]:
                bi += 1
            self._p2b[id(p)] = bi                    # (cut at parameter boundaries: one bucket per parameter)
            self._total[bi] += 1
        self.groups = []          # [(bucket indices, external event)] in the order they complete during backward
        self._left, self._pending, self._main = None, [], None

    @classmethod
    def probe(cls, device):
        """"node": a wait issued after the launch waits for the event-record node; "graph": it waits for more (correct, no
        overlap); "none": it does not wait (unusable); "error: ...": external events cannot be captured here."""
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        if key in cls._probe:
            return cls._probe[key]
        res = "node"
        try:
            with torch.cuda.device(key):
                x = torch.zeros(1, device="cuda")
                ev = ExternalEvent()
                aux = torch.cuda.Stream()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(aux):             # (first uses load code objects: not inside the timed window)
                    x.fill_(0.0)
                    x.clone()
                    torch.cuda._sleep(1000)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    torch.cuda._sleep(10_000_000)
                    x.fill_(1.0)
                    ev.record()
                    torch.cuda._sleep(40_000_000)
                    x.fill_(2.0)
                for _ in range(2):                   # (the second replay: the event must follow the NEW launch)
                    x.zero_()
                    torch.cuda.synchronize()
                    g.replay()
                    ev.wait(aux)
                    with torch.cuda.stream(aux):
                        y = x.clone()
                    torch.cuda.synchronize()
                    v = float(y)
                    if v == 0.0:
                        res = "none"
                    elif v == 2.0 and res == "node":
                        res = "graph"
        except Exception as e:                       # noqa: BLE001 -- any failure simply means: exchange behind the graph
            res = f"error: {type(e).__name__}: {e}"
            torch.cuda.synchronize()
        cls._probe[key] = res
        return res

    # -- inside the capture ---------------------------------------------------------------------------------------
    def begin(self):
        """Right before ``backward()`` of the step being captured."""
        from . import ops
        self._left = list(self._total)
        self._pending, self.groups = [], []
        self._main = torch.cuda.current_stream()
        ops.GradSink.observer = self

    def mark(self, p):
        i = self._p2b.get(id(p))
        if i is not None:
            self._left[i] -= 1
            if self._left[i] == 0:
                self._pending.append(i)

    def spoil(self, p):
        """A parameter whose slot was written through is used AGAIN in this step (a second pass through the model before the
        backward: the triplet passes): autograd will add that gradient into the slot at a time nobody announces, so the
        parameter's bucket is only complete when the backward is -- it goes behind the graph."""
        i = self._p2b.get(id(p))
        if i is None:
            return
        self._left[i] = 1 << 30
        if i in self._pending:
            self._pending.remove(i)
        for todo, _evs in self.groups:
            if i in todo:
                todo.remove(i)

    def flush_pending(self):
        """Called at the next sink request: every kernel of the earlier requests has been enqueued on its stream."""
        if not self._pending:
            return
        from . import ops
        todo, self._pending = self._pending, []
        evs = []
        for st in {self._main, ops.SidePrep.stream(self._main.device), *ops.PrepAhead.branch_streams}:
            with torch.cuda.stream(st):
                capturing = torch.cuda.is_current_stream_capturing()
            if capturing:                                # (a stream that is not part of the capture holds none of this step's kernels)
                ev = ExternalEvent()
                ev.record(st)
                evs.append(ev)
        self.groups.append((todo, evs))

    def wait(self, group_events, stream):
        for ev in group_events:
            ev.wait(stream)

    def end(self):
        """Right after ``backward()``."""
        from . import ops
        ops.GradSink.observer = None
        self._pending = []


class TorchComm:
    """The communicator interface of rccl_comm.RcclComm on torch.distributed collectives (any backend; gloo on CPU for
    the multi-process tests).  `side` is None: the collectives run synchronously on the caller's stream."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.side = None

    def all_reduce_(self, t, stream=None):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def reduce_scatter(self, send, recv, stream=None):
        tmp = send.clone()                      # (gloo has no reduce_scatter: all-reduce a copy, keep this rank's slice)
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
        n = recv.numel()
        recv.copy_(tmp[self.rank * n:(self.rank + 1) * n])

    def all_gather(self, send, recv, stream=None):
        parts = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(parts, send.contiguous(), group=self.group)
        recv.copy_(torch.cat(parts))

    def broadcast_(self, t, root=0, stream=None):
        dist.broadcast(t, src=root, group=self.group)


class StreamedGradExchange:
    """Bucketed gradient exchange of FusedAdamW's flat gradient buffer on the communicator's side stream, driven by the
    backward kernels themselves: with write-through gradients (ops.GradSink) every parameter gradient is written by a HIP
    kernel straight into its slot of the flat buffer; the sink tells this object which slots have been written, and as
    soon as the last slot of a bucket is on the compute stream the bucket's collective is enqueued on the side stream
    behind an event -- no autograd hooks, so it works unchanged while the step is being CAPTURED into a hipGraph: the
    replayed graph then contains forward, backward, the overlapped exchange, the join and the optimizer step.

    sharded=False: SUM all-reduce of each bucket, then the caller's optimizer.step() on the whole buffer.
    sharded=True:  SUM reduce-scatter of each bucket (rank r keeps sub-slice r of every bucket), `finish()` runs the
                   fused AdamW on this rank's sub-slices only (1/world of the 4 reads + 3 writes per parameter) and
                   all-gathers the updated parameters in place -- same bytes on the wire as the all-reduce.
    Buckets whose parameters are not all written through (autograd-accumulated gradients) go out at finish()."""

    def __init__(self, optimizer, comm, bucket_bytes=64 << 20, sharded=False):
        self.opt, self.comm, self.sharded = optimizer, comm, sharded
        self.world = comm.world
        self.bucket_elems = max(self.world, bucket_bytes // 4)
        self._buckets = None          # [start, end, params_total]
        self._left = None
        self._pending = []
        self._launched = None
        self._forked = False
        self.early = 0                # buckets of the last step that went out before backward had finished
        if sharded:
            optimizer.shard_exchange = self      # FusedAdamW.state_dict() gathers the sharded moments through gather_state()

    # -- layout ---------------------------------------------------------------------------------------------------
    def _build(self):
        opt = self.opt
        assert opt.built, "the flat gradient layout exists after the first optimizer step"
        n = opt.flat_g.numel()
        assert n % self.world == 0 or not self.sharded, "FusedAdamW(pad_to=world) keeps the flat buffer divisible by the world size"
        bounds, start = [], 0
        for p in opt._flat_params:
            off, k = opt._offsets[id(p)]
            end = off + k
            if end - start >= self.bucket_elems:
                if self.sharded:
                    end -= (end - start) % self.world          # shardable buckets; the remainder opens the next bucket
                if end > start:
                    bounds.append((start, end))
                    start = end
        if start < n:
            bounds.append((start, n))
        self._buckets = bounds
        self._p2b = {}
        self._total = [0] * len(bounds)
        for p in opt._flat_params:
            off, k = opt._offsets[id(p)]
            bs = [i for i, (s0, e0) in enumerate(bounds) if off < e0 and off + k > s0]
            self._p2b[id(p)] = bs
            for i in bs:
                self._total[i] += 1

    # -- per step -------------------------------------------------------------------------------------------------
    def begin(self):
        """Call before backward (after optimizer.zero_grad())."""
        from . import ops
        if self._buckets is None and self.opt.built:
            self._build()
        ops.GradSink.listener = self if self._buckets is not None else None
        if self._buckets is not None:
            self._left = list(self._total)
            self._launched = [False] * len(self._buckets)
        self._pending = []
        self._forked = False
        self.early = 0

    def mark(self, p):
        for i in self._p2b.get(id(p), ()):
            self._left[i] -= 1
            if self._left[i] == 0:
                self._pending.append(i)

    def flush_pending(self):
        if self._pending:
            todo, self._pending = self._pending, []
            for i in todo:
                self.early += not self._launched[i]
                self._launch(i)

    def _launch(self, i):
        if self._launched[i]:
            return
        self._launched[i] = True
        s0, e0 = self._buckets[i]
        g = self.opt.flat_g[s0:e0]
        side = self.comm.side
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())       # the bucket's last gradient kernel is on the compute stream
            self._forked = True
        if self.sharded:
            k = (e0 - s0) // self.world
            self.comm.reduce_scatter(g, g[self.comm.rank * k:(self.comm.rank + 1) * k], stream=side)
        else:
            self.comm.all_reduce_(g, stream=side)

    def finish(self, step=True):
        """Call after backward: sends what is left, joins the side stream, and (sharded) steps + re-gathers."""
        from . import ops
        ops.GradSink.listener = None
        opt = self.opt
        if self._buckets is None:          # first step: no flat layout yet -> whole-tensor all-reduces, as GradReducer
            for p in opt.param_groups[0]["params"]:
                if p.grad is not None:
                    self.comm.all_reduce_(p.grad.contiguous().view(-1) if p.grad.is_contiguous() else p.grad)
            return False
        self._pending = []
        for i in range(len(self._buckets)):
            self._launch(i)
        rest = [p for p in opt.param_groups[0]["params"] if id(p) not in opt._flat_ids and p.grad is not None]
        side = self.comm.side
        for p in rest:
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            self.comm.all_reduce_(p.grad.view(-1), stream=side)
        if not self.sharded:
            if side is not None and self._forked:
                torch.cuda.current_stream().wait_stream(side)
            return False
        # sharded: AdamW on this rank's sub-slice of every bucket (on the side stream, right behind its reduce-scatter),
        # then the all-gather of the updated parameters
        ctx = torch.cuda.stream(side) if side is not None else _null()
        with ctx:
            if step:
                opt.step_shards([(s0 + self.comm.rank * ((e0 - s0) // self.world), (e0 - s0) // self.world) for s0, e0 in self._buckets])
            for s0, e0 in self._buckets:
                k = (e0 - s0) // self.world
                pb = opt.flat_p[s0:e0]
                self.comm.all_gather(pb[self.comm.rank * k:(self.comm.rank + 1) * k], pb, stream=side)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return True      # the optimizer step has been taken


    def gather_state(self):
        """Sharded mode: all-gather exp_avg / exp_avg_sq so that every rank holds the full AdamW state (rank r only ever
        updates sub-slice r of every bucket; the reference's checkpoint, attn_unet_data_parallel.py:946-952, stores the
        whole optimizer state).  Collective: called by FusedAdamW.state_dict() on every rank."""
        if not self.sharded or self.world == 1:
            return
        if self._buckets is None:
            if not self.opt.built:
                return                 # nothing has been stepped in shards yet
            self._build()
        opt, side = self.opt, self.comm.side
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        for buf in (opt.flat_m, opt.flat_v):
            for s0, e0 in self._buckets:
                k = (e0 - s0) // self.world
                b = buf[s0:e0]
                self.comm.all_gather(b[self.comm.rank * k:(self.comm.rank + 1) * k], b, stream=side)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
