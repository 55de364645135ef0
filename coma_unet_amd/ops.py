"""torch.autograd.Function wrappers over the C ABI (include/coma_unet.h).

Internal activation layout is channels-last (B, D, H, W, C); a tensor may be a channel
slice of a wider buffer (that is how the reference's torch.cat calls,
attn_unet_data_parallel.py:229,651,654, are made copy-free).  Every Function calls the HIP
library on the current stream; none of them has a CPU or PyTorch fallback.
"""
from __future__ import annotations

import contextlib
import os

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import lib, ct, ptr, check, workspace


class Out:
    """Side-channel holder for a pre-allocated destination view (not an autograd input)."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class KernelTimer:
    """Optional HIP-event bracket around every convolution launch (bench.py's live roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""
    enabled = False
    records = []   # (kind, algo, flops, ev_start, ev_end, layer tag, kernel name)

    @classmethod
    def run(cls, kind, algo, flops, fn, tag=None, name=None):
        """`name`: the kernel(s) of a call that does not go through the convolution dispatch (normalisation, gate)."""
        if not cls.enabled:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        # the kernel variant the library just launched, spelled as rocprofv3 prints it (coma_last_kernel)
        cls.records.append((kind, algo, flops, e0, e1, tag, name if name is not None else lib.coma_last_kernel().decode()))
        return r

    @classmethod
    def by_kernel(cls):
        """{kernel name: (launches, total_ms, total_flops, total_bytes)} -- call after torch.cuda.synchronize().  A bracket
        also covers the small helper launches of a dispatch (split-K merge, replica sum, memset): they are part of the
        layer's cost and are attributed to its main kernel."""
        out = {}
        for kind, algo, fb, e0, e1, _tag, kname in cls.records:
            fl, by = fb if isinstance(fb, tuple) else (fb, 0.0)
            n, ms, f0, b0 = out.get(kname, (0, 0.0, 0.0, 0.0))
            out[kname] = (n + 1, ms + e0.elapsed_time(e1), f0 + fl, b0 + by)
        return out

    @classmethod
    def summary(cls):
        """{(kind, algo): (launches, total_ms, total_flops, total_bytes)} -- call after torch.cuda.synchronize().
        `flops` entries are (flops, algorithmic_bytes) pairs."""
        out = {}
        for kind, algo, fb, e0, e1, _tag, _k in cls.records:
            fl, by = fb if isinstance(fb, tuple) else (fb, 0.0)
            n, ms, f0, b0 = out.get((kind, algo), (0, 0.0, 0.0, 0.0))
            out[(kind, algo)] = (n + 1, ms + e0.elapsed_time(e1), f0 + fl, b0 + by)
        return out

    @classmethod
    def by_layer(cls):
        """{(kind, tag): (launches, total_ms, total_flops)} for the per-layer table (profiles/layer_table.py)."""
        out = {}
        for kind, algo, fb, e0, e1, tag, _k in cls.records:
            fl = fb[0] if isinstance(fb, tuple) else fb
            n, ms, f0 = out.get((kind, tag), (0, 0.0, 0.0))
            out[(kind, tag)] = (n + 1, ms + e0.elapsed_time(e1), f0 + fl)
        return out


def conv_flops(x_shape, y_shape, ksize, stride):
    """(2 * MACs, algorithmic HBM bytes) of one gather launch: every (coarse-grid voxel, tap, c, n) combination
    once; each operand voxel read once and each result voxel written once (2-byte elements)."""
    vx = x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3]
    vy = y_shape[0] * y_shape[1] * y_shape[2] * y_shape[3]
    return (2.0 * min(vx, vy) * (ksize ** 3) * x_shape[4] * y_shape[4], 2.0 * (vx * x_shape[4] + vy * y_shape[4]))


def conv_class(algo_name, cin, cout):
    """MFMA launches are split into 'thick' (>= 32 channels both sides: MFMA-bound) and 'thin' (zero-padded
    tiles on the full-resolution 1..16-channel layers: HBM-bound) so the roofline is quoted per regime."""
    if algo_name != "mfma":
        return algo_name
    return "mfma-thick" if min(cin, cout) >= 32 else "mfma-thin"


_ALGO_NAMES = {1: "direct", 2: "mfma", 3: "mfma-f32"}


def _new(shape, dtype, device):
    """Activation buffer.  (B, D, H, W, C) volumes whose channel count is not a multiple of 8 (the 1..3-channel tensors of
    the full-resolution tail) get a voxel pitch rounded up to 8 channels: every kernel takes the pitch `ld`, and 16-byte
    aligned rows let the convolution staging use (masked) vector loads instead of per-element ones."""
    if len(shape) == 5 and shape[4] % 8 != 0:
        C = shape[4]
        return torch.empty(tuple(shape[:4]) + ((C + 7) // 8 * 8,), dtype=dtype, device=device)[..., :C]
    return torch.empty(shape, dtype=dtype, device=device)


def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


class ZeroArena:
    """ONE zeroed device buffer per step for everything the kernels merge into with atomics: the fp64 statistics records
    of the normalisations (forward {sum, sumsq}, backward partial sums), the kernel-layout weight gradients, their
    replica / split-K scratch and the routing gradients.  ``FusedAdamW.zero_grad()`` arms it, every taker gets a private
    slice that is all zeros, and ``FusedAdamW.step()`` (after backward: every slice is dead by then) clears what the
    step dirtied with ONE memset -- instead of ~60 memset nodes and ~95 finalise launches per step.  Invariant: outside
    [0, cur) the buffer is zero.  Not armed (op tests, inference): ``take`` returns None and callers fall back to
    ``torch.zeros`` / the library's own memsets.  The clearing memset sits at the END of the step so that a step
    captured into a hipGraph clears exactly what it dirtied, whatever ran before the capture."""
    _arenas = {}
    nbytes = int(os.environ.get("COMA_ZERO_ARENA_MB", "1024")) << 20
    enabled = os.environ.get("COMA_ZERO_ARENA", "1") not in ("0", "")

    def __init__(self, device):
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.cur = 0
        self.armed = False
        self.missed = 0          # bytes requests that did not fit (diagnostic)
        self.peak = 0            # most bytes one step has used (diagnostic)

    @classmethod
    def get(cls, device):
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        a = cls._arenas.get(key)
        if a is None:
            a = cls._arenas[key] = ZeroArena(torch.device("cuda", key))
        return a

    @classmethod
    def begin_step(cls, device=None):
        if not cls.enabled or not torch.cuda.is_available():
            return
        a = cls.get(device if device is not None else torch.cuda.current_device())
        a.clean()                # (a step that died half-way, or forward passes outside a step)
        a.armed = True

    @classmethod
    def end_step(cls):
        for a in cls._arenas.values():
            a.clean()
            a.armed = False

    def clean(self):
        if self.cur:
            self.peak = max(self.peak, self.cur)
            self.buf[:self.cur].zero_()
            self.cur = 0

    @classmethod
    def take(cls, nbytes, device, dtype=torch.uint8):
        """A zeroed 1-d tensor of `nbytes` bytes viewed as `dtype`, or None when the arena is not armed / exhausted."""
        if not cls.enabled:
            return None
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        a = cls._arenas.get(key)
        if a is None or not a.armed:
            return None
        n = (int(nbytes) + 255) & ~255
        if a.cur + n > a.buf.numel():
            a.missed += n
            return None
        t = a.buf[a.cur:a.cur + int(nbytes)]
        a.cur += n
        return t if dtype == torch.uint8 else t.view(dtype)


STAT_REPLICAS = 8      # COMA_STAT_REPLICAS of include/coma_unet.h


def stat_record(G, C, k, device):
    """A zeroed fp64 statistics record [STAT_REPLICAS, stride] with stride = G*C*k rounded up to a 64-byte line
    (COMA_NORM_RECORD_DOUBLES): arena slice, else a fresh torch.zeros."""
    rs = (G * C * k + 7) & ~7
    t = ZeroArena.take(8 * STAT_REPLICAS * rs, device, torch.float64)
    if t is None:
        t = torch.zeros(STAT_REPLICAS * rs, dtype=torch.float64, device=device)
    return t.view(STAT_REPLICAS, rs)


def stats_from_sums(sums, G, C, count, eps):
    """(mean, rstd) fp32 [G, C] of a forward statistics record -- what every kernel derives on the fly (csrc/norm.hip,
    NormStat: replicas summed, mean = sum / count, rstd = 1 / sqrt(sumsq / count - mean^2 + eps)); host-side twin for
    tests and diagnostics."""
    t = sums.sum(0)[:G * C * 2].view(G, C, 2)
    m = t[..., 0] / count
    var = (t[..., 1] / count - m * m).clamp_min(0.0)
    return m.float(), (1.0 / torch.sqrt(var + eps)).float()


class GradSink:
    """Write-through parameter gradients.  ``FusedAdamW(write_through=True)`` marks every parameter whose
    ``.grad`` is a slot of its flat gradient buffer; the backward kernels then write that slot directly ("=")
    and hand autograd ``None`` -- no temporary, no AccumulateGrad add launch per parameter.  A second use of the
    same parameter inside one step falls back to the ordinary returned gradient (autograd adds it to the slot).
    Post-accumulate-grad hooks do not fire for sunk gradients, so hook-driven overlap keeps this off."""
    written = set()
    listener = None      # data_parallel.StreamedGradExchange: told which gradient slots have been written
    observer = None      # data_parallel.GraphBucketWatch: the same notifications, passively (the two-stream step stays on)

    @classmethod
    def begin_step(cls):
        cls.written.clear()
        SidePrep.join()
        SidePrep._live = 0
        PrepAhead.live = 0
        ZeroArena.begin_step()

    @classmethod
    def slots(cls, params):
        """Gradient slots of ALL the parameters ONE kernel launch is about to write (None where a parameter is not sunk).
        Flushing and marking are separate steps: a new request proves that the kernels of every EARLIER request have been
        enqueued (python runs the backward nodes one after another on one stream), so buckets those completed may go on the
        wire now -- but the slots marked by THIS request only become sendable at the next request, after this node's kernel
        has been launched.  (Asking slot by slot let the second request of a three-slot node -- Routing, NormAct -- flush a
        bucket completed by the first one before the node's single kernel was enqueued: the collective read zeros.)"""
        if cls.listener is not None:
            cls.listener.flush_pending()
        if cls.observer is not None:
            cls.observer.flush_pending()
        out = []
        for p in params:
            if p is None or not getattr(p, "_coma_sink", False) or p.grad is None or id(p) in cls.written:
                if p is not None and cls.observer is not None and id(p) in cls.written:
                    cls.observer.spoil(p)      # (a second use: autograd accumulates into the slot, unannounced)
                out.append(None)
                continue
            cls.written.add(id(p))
            if cls.listener is not None:
                cls.listener.mark(p)
            if cls.observer is not None:
                cls.observer.mark(p)
            out.append(p.grad)
        return out

    @classmethod
    def slot(cls, p):
        return cls.slots((p,))[0]


class GradFork:
    """Gradient meeting point of an activation with several consumers (a skip connection: the next encoder block, the gate's
    W_x convolution and its final multiply; the up-convolution's output: W_g and the merge convolution).  Autograd would
    give every consumer its own gradient tensor and add them pairwise (an ATen pass of 3 tensor volumes per extra
    consumer: 0.6 ms per 128^3 step).  Here the first consumer's backward kernel WRITES the shared buffer and the later
    ones ACCUMULATE into it in their epilogues (COMA_ACCUMULATE, gate_apply_bwd's accumulate flag); they hand autograd
    ``None``, and ``Fork.backward`` returns the buffer.  A consumer whose kernel cannot accumulate simply returns its
    gradient the ordinary way and ``Fork.backward`` adds it -- correct in any mix and any order."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def offer(self, t):
        """A consumer that already holds its gradient as a tensor (a channel slice of the concat buffer's gradient): it
        becomes the shared buffer if there is none yet.  True: taken (hand autograd None)."""
        if self.buf is None:
            self.buf = t
            return True
        return False


class Fork(Function):
    @staticmethod
    def forward(ctx, x, fork):
        ctx.set_materialize_grads(False)
        ctx.fork = fork
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.fork.buf = ctx.fork.buf, None
        if buf is None:
            return g, None
        if g is None:
            return buf, None
        return buf + g, None


def fork(x):
    """x with a gradient meeting point attached (see GradFork); x itself when no gradient will flow."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x
    f = GradFork()
    y = Fork.apply(x, f)
    y._coma_fork = f
    return y


class SidePrep:
    """Weight preparation off the critical path.  Routing, the expert mix / re-layout of the masters and (backward) the
    scatter of the kernel-layout weight gradient to the experts depend on parameters and covariates only, never on an
    activation -- but in one stream they sit between the convolutions (~110 launches of 4-20 us forward, ~50 backward, plus
    ~1.3 ms of bandwidth-bound mixes on the deep layers).  They run on a second HIP stream here: forked at the start of a
    model forward, every convolution waits for the event of ITS weights only, the backward scatters are joined before the
    optimizer.  Under hipGraph capture the stream becomes a parallel branch of the graph.

    Prepared weights live in persistent per-layer buffers (they are read by the convolution on the main stream after
    the preparing call has returned: an allocator block could be handed out again on the side stream too early).  A second
    forward before the backward of the first (or a layer used twice in one forward) falls back to the one-stream path."""
    # OFF by default: measured on the 128^3 step (hipGraph replay) it is SLOWER, 22.68 ms one stream -> 23.31 ms (forward
    # and backward on the side stream) / 23.78 ms (forward only): the ~200 cross-stream dependencies of the graph cost more
    # than the ~1 ms of small launches and bandwidth-bound mixes they take off the main chain.  COMA_SIDE_PREP=1 enables it.
    enabled = os.environ.get("COMA_SIDE_PREP", "0") not in ("0", "")
    backward = os.environ.get("COMA_SIDE_PREP", "0") != "f"      # ("f": forward preparation only)
    _streams, _bufs, _used = {}, {}, set()
    _live = 0          # ConvLayer nodes whose backward still needs the persistent dgrad weights
    _on = False

    @classmethod
    def stream(cls, dev):
        key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=key)
        return cls._streams[key]

    @classmethod
    def begin(cls, dev):
        """Start of a model forward."""
        cls._on = False
        if not cls.enabled or GradSink.listener is not None or cls._live > 0 or torch.device(dev).type != "cuda":
            return
        cls.stream(dev).wait_stream(torch.cuda.current_stream(dev))
        cls._used.clear()
        cls._on = True

    @classmethod
    def fence(cls, dev):
        """The current stream waits for everything queued on the side stream so far."""
        ev = torch.cuda.Event()
        ev.record(cls.stream(dev))
        torch.cuda.current_stream(dev).wait_event(ev)
        if os.environ.get("COMA_SIDE_SYNC") == "1":      # (debugging aid: no overlap at all)
            torch.cuda.synchronize(dev)

    @classmethod
    def follow(cls, dev):
        """The side stream waits for everything queued on the current stream so far."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        cls.stream(dev).wait_event(ev)

    @classmethod
    def join(cls):
        for key, st in cls._streams.items():
            torch.cuda.current_stream(key).wait_stream(st)
        for st in PrepAhead.branch_streams:              # (model branches queued on an idle preparation stream write gradients through too)
            torch.cuda.current_stream(st.device).wait_stream(st)
        PrepAhead.branch_streams.clear()
        cls._on = False
        WgradSide.joined()

    @classmethod
    def buffers(cls, master, shape_f, fwd_dtype, shape_d, dgrad_dtype):
        """Persistent (wk_f, wk_d) of this layer, or None when it was already prepared in this forward."""
        if id(master) in cls._used:
            return None
        cls._used.add(id(master))
        key = (id(master), tuple(shape_f), fwd_dtype, dgrad_dtype)
        ent = cls._bufs.get(key)
        if ent is None or ent[0] is not master:
            dev = master.device
            # from the SIDE stream's pool: a block of this stream's pool may still be read by a queued kernel of its previous
            # owner, and the side stream (which runs ahead) would overwrite it without waiting
            with torch.cuda.stream(cls.stream(dev)):
                ent = (master, _new(shape_f, fwd_dtype, dev), _new(shape_d, dgrad_dtype, dev) if dgrad_dtype is not None else None)
            cls._bufs[key] = ent
        return ent[1], ent[2]


class PrepAhead:
    """All weight preparation of a forward pass AT ITS START, on a few streams beside each other.  Routing and the expert mix
    / re-layout of the masters depend on parameters and covariates only; in stream order they sit in front of every
    convolution as ~80 launches of 4-20 us (0.74 ms per 128^3 step), most of them far too small to use the chip: the mixes of
    the 20 small layers take 5-10 us each for well under a microsecond of HBM traffic.  The first forward at a given input
    shape RECORDS which layers prepare what (`plan`); later forwards run every entry up front on `K` streams -- the chip sees
    them together, as parallel branches of the step graph under capture -- into persistent per-entry buffers, the main
    stream waits once, and the layers find their weights ready (`take`).  Any deviation from the recorded order, a second
    forward while a backward still needs the buffers, or COMA_PREP_AHEAD=0 falls back to the in-line preparation.
    (Interleaved with the convolutions instead -- ops.SidePrep -- the same work on a second stream made the step SLOWER.)"""
    enabled = os.environ.get("COMA_PREP_AHEAD", "1") not in ("0", "")
    # streams: measured 17.70-17.85 ms per step without, 17.89 with 1, 17.53-17.56 with 2, 17.61 with 3, 17.85 with 4, 18.2 with 8
    K = int(os.environ.get("COMA_PREP_AHEAD_STREAMS", "2"))
    _streams = {}
    _cur = None        # the running forward: {"mode", "plan", "i", "res", "owner", "key"}
    branch_streams = []  # preparation streams a model branch ran on in this step (the projection heads): joined with the side stream
    live = 0           # ConvLayer nodes whose backward still reads the persistent buffers
    used = 0           # layers served from the up-front preparation so far (diagnostic / tests)

    @classmethod
    def streams(cls, dev):
        key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        if key not in cls._streams:
            cls._streams[key] = [torch.cuda.Stream(device=key) for _ in range(max(1, cls.K))]
        return cls._streams[key]

    @classmethod
    def begin(cls, owner, key, dev, covariate, batch):
        """Start of a model forward.  owner: the module that keeps the plans; key: what the plan depends on (input shape,
        dtypes, mode flags); covariate: the (B, 1, n) fp32 covariate tensor the conditional layers slice their rows from."""
        cls._cur = None
        if not cls.enabled or cls.live > 0 or torch.device(dev).type != "cuda" or SidePrep._on:
            return
        plans = owner.__dict__.setdefault("_prep_ahead_plans", {})
        plan = plans.get(key)
        if plan is None:
            cls._cur = {"mode": "record", "plan": [], "i": 0, "res": None, "owner": owner, "key": key}
            return
        if plan == "off":
            return
        from .layers import cov_rows
        main = torch.cuda.current_stream(dev)
        covs = {}
        for e in plan:                      # (the covariate rows are made on the main stream, before the fork)
            if e["ncov"] is not None and e["ncov"] not in covs:
                c = covariate if covariate.shape[-1] == e["ncov"] else covariate[:, :, :e["ncov"]]
                covs[e["ncov"]] = cov_rows(c, batch, dev)
        fork = torch.cuda.Event()
        fork.record(main)
        sts = cls.streams(dev)
        for st in sts:
            st.wait_event(fork)
        # largest masters first, each to the stream with the least work so far (the order is free: everything is done
        # before the first layer runs)
        load = [0] * len(sts)
        res = [None] * len(plan)
        for i in sorted(range(len(plan)), key=lambda j: -plan[j]["master"].numel()):
            e = plan[i]
            k = load.index(min(load))
            load[k] += e["master"].numel() + (1 << 18)          # (+ a launch's worth: the tiny layers are not free)
            st = sts[k]
            with torch.cuda.stream(st):
                master = e["master"]
                r = bm = None
                if e["ncov"] is not None:
                    cov = covs[e["ncov"]]
                    Wr, br, be = e["routing"]
                    if e["bufs"] is None:
                        e["bufs"] = [_f32((batch, Wr.shape[0]), dev), _f32((batch, be.shape[1]), dev), None]
                    r, bm = e["bufs"][0], e["bufs"][1]
                    check(lib.coma_routing_fwd(ptr(cov), batch, cov.shape[1], ptr(Wr), ptr(br), Wr.shape[0], ptr(be), be.shape[1],
                                               ptr(r), ptr(bm), L.stream()), "coma_routing_fwd")
                elif e["bufs"] is None:
                    e["bufs"] = [None, None, None]
                if e["bufs"][2] is None:
                    E_ = master.shape[0] if r is not None else 1
                    A_, B_ = master.shape[-5], master.shape[-4]
                    taps_ = master.shape[-1] * master.shape[-2] * master.shape[-3]
                    cout_, cin_ = (B_, A_) if e["transposed"] else (A_, B_)
                    Bw_ = batch if r is not None else 1
                    e["bufs"][2] = (_new((Bw_, taps_, cout_, cin_), e["fwd_dtype"], dev),
                                    _new((Bw_, taps_, cin_, cout_), e["dgrad_dtype"], dev) if e["dgrad_dtype"] is not None else None)
                wk_f, wk_d, rr, pmeta = _prep_fwd(master, r, e["transposed"], e["fwd_dtype"], e["dgrad_dtype"], e["bufs"][2])
                res[i] = (r, bm, wk_f, wk_d, rr, pmeta)
        for st in sts:
            main.wait_stream(st)
        cls._cur = {"mode": "replay", "plan": plan, "i": 0, "res": res, "owner": owner, "key": key}

    @classmethod
    def note(cls, master, routing, ncov, transposed, fwd_dtype, dgrad_dtype):
        """Recording forward: this layer prepares `master` this way."""
        c = cls._cur
        if c is not None and c["mode"] == "record":
            c["plan"].append({"master": master, "routing": routing, "ncov": ncov, "transposed": bool(transposed),
                              "fwd_dtype": fwd_dtype, "dgrad_dtype": dgrad_dtype, "bufs": None})

    @classmethod
    def take(cls, master, transposed, fwd_dtype, dgrad_dtype):
        """The next entry's prepared (r, bias_mix, wk_f, wk_d, rr, pmeta) if it is this layer's; else None (and the plan is
        dropped: the model took another path than the recorded one)."""
        c = cls._cur
        if c is None or c["mode"] != "replay":
            return None
        i = c["i"]
        if i < len(c["plan"]):
            e = c["plan"][i]
            if e["master"] is master and e["transposed"] == bool(transposed) and e["fwd_dtype"] == fwd_dtype and e["dgrad_dtype"] == dgrad_dtype:
                c["i"] = i + 1
                cls.used += 1
                return c["res"][i]
        c["mode"] = "broken"
        c["owner"].__dict__["_prep_ahead_plans"].pop(c["key"], None)
        return None

    @classmethod
    def end(cls, ok=True):
        c, cls._cur = cls._cur, None
        if c is None:
            return
        plans = c["owner"].__dict__.setdefault("_prep_ahead_plans", {})
        if c["mode"] == "record" and ok and c["plan"]:
            plans[c["key"]] = c["plan"]
        elif c["mode"] == "replay" and ok and c["i"] != len(c["plan"]):
            plans.pop(c["key"], None)          # fewer layers than recorded: record again next time


class WgradSide:
    """Weight gradients beside the data-gradient chain.  A layer's weight gradient (MFMA-bound, 1.7 + 0.9 + 0.9 ms per step
    over the thick / strided / thin layers), the scatter of it to the experts and the routing backward feed nothing but the
    optimizer, while the chain they sit in today -- data gradient, normalisation backward, gate backward -- is mostly
    HBM-bound.  With the step's gradients written through to the flat buffer (``GradSink``) they run on SidePrep's second
    HIP stream: forked where the layer's ``dy`` exists, joined before the optimizer (``SidePrep.join``); under hipGraph
    capture a parallel branch of the graph.  Everything the side stream still reads is kept referenced until the join
    (``keep``): an allocator block of this stream's pool must not be handed out again while a queued side kernel reads it.
    Off for data-parallel overlap (``GradSink.listener``: bucket order is launch order there), for gradients that go
    through autograd instead of a sink slot, and while ``KernelTimer`` brackets launches.  COMA_WGRAD_SIDE=0 disables it."""
    enabled = os.environ.get("COMA_WGRAD_SIDE", "1") not in ("0", "")
    keep = []
    dirty = False        # the side stream holds work of this step that the current stream has not waited for
    launched = 0         # weight gradients sent to the side stream so far (diagnostic / tests)

    @classmethod
    def usable(cls, dev, params):
        """May this backward node launch on the side stream?  `params`: every parameter whose gradient it produces."""
        if not cls.enabled or KernelTimer.enabled or GradSink.listener is not None or torch.device(dev).type != "cuda":
            return False
        for p in params:
            if p is None:
                continue
            if not getattr(p, "_coma_sink", False) or p.grad is None or id(p) in GradSink.written:
                return False
        return True

    @classmethod
    def begin(cls, dev, entry, *tensors):
        """The side stream waits for `entry` (an event of the current stream); `tensors` stay alive until the join, which
        the first use in a backward pass queues for the END of that pass (the caller of backward() finds every gradient
        complete on its stream, as without the side stream).  Returns the side stream."""
        st = SidePrep.stream(dev)
        st.wait_event(entry)
        cls.keep.extend(t for t in tensors if t is not None)
        cls.launched += 1
        if not cls.dirty:
            cls.dirty = True
            torch.autograd.Variable._execution_engine.queue_callback(SidePrep.join)
        return st

    @classmethod
    def joined(cls):
        cls.keep.clear()
        cls.dirty = False


# --------------------------------------------------------------------------------------
# weights: CondConv expert mixing + kernel-layout cast
# --------------------------------------------------------------------------------------
def _prep_fwd(master, r, transposed, fwd_dtype, dgrad_dtype, bufs=None):
    """-> wk_f [Bw, taps, Cout, Cin], wk_d [Bw, taps, Cin, Cout] or None, r (fp32, contiguous) or None, meta.
    bufs: "persistent" takes the layer's SidePrep buffers when they are free (returns side=True in that case)."""
    has_e = r is not None
    m = master if has_e else master.unsqueeze(0)
    assert m.is_contiguous() and m.dtype == torch.float32
    E, A, Bc = m.shape[0], m.shape[1], m.shape[2]
    taps = m.shape[3] * m.shape[4] * m.shape[5]
    cout, cin = (Bc, A) if transposed else (A, Bc)
    se = A * Bc * taps
    if transposed:      # master [ci][co][tap]
        sn_f, sc_f = taps, cout * taps
    else:               # master [co][ci][tap]
        sn_f, sc_f = cin * taps, taps
    Bw = r.shape[0] if has_e else 1
    rr = r.contiguous().float() if has_e else None
    dev = master.device
    if bufs is not None:
        wk_f, wk_d = bufs
    else:
        wk_f = _new((Bw, taps, cout, cin), fwd_dtype, dev)
        wk_d = _new((Bw, taps, cin, cout), dgrad_dtype, dev) if dgrad_dtype is not None else None
    if taps == 27:
        # one pass over the experts writes both layouts: [tap][A][B] and [tap][B][A] of master [E][A][B][27]
        ab, ba = (wk_d, wk_f) if transposed else (wk_f, wk_d)
        check(lib.coma_weight_prep_pair(ptr(m), ptr(rr), E, Bw, A, Bc, ptr(ab), L.dtype_code(ab.dtype) if ab is not None else 0,
                                        ptr(ba), L.dtype_code(ba.dtype) if ba is not None else 0, L.stream()),
              "coma_weight_prep_pair")
    else:
        check(lib.coma_weight_prep(ptr(m), ptr(rr), E, Bw, cout, cin, taps, se, sn_f, sc_f, ptr(wk_f),
                                   L.dtype_code(fwd_dtype), L.stream()), "coma_weight_prep")
        if wk_d is not None:
            check(lib.coma_weight_prep(ptr(m), ptr(rr), E, Bw, cin, cout, taps, se, sc_f, sn_f, ptr(wk_d),
                                       L.dtype_code(dgrad_dtype), L.stream()), "coma_weight_prep")
    return wk_f, wk_d, rr, (has_e, E, Bw, cout, cin, taps, se, sn_f, sc_f)


def _prep_bwd(dwk, master, rr, meta, p_master):
    """fp32 dwk [Bw, taps, Cout, Cin] -> (dmaster or None when written through to p_master.grad, dr or None)."""
    has_e, E, Bw, cout, cin, taps, se, sn_f, sc_f = meta
    sink = GradSink.slot(p_master)
    dmaster = sink if sink is not None else torch.empty_like(master)
    dr, zeroed = None, 0
    if has_e:
        dr = ZeroArena.take(4 * Bw * E, master.device, torch.float32)
        if dr is not None:
            dr, zeroed = dr.view(Bw, E), L.ZEROED_OUT
        else:
            dr = _f32((Bw, E), master.device)
    check(lib.coma_weight_prep_bwd(ptr(dwk), ptr(master), ptr(rr), E, Bw, cout, cin, taps, se, sn_f, sc_f,
                                   ptr(dmaster), ptr(dr), zeroed, L.stream()), "coma_weight_prep_bwd")
    return (None if sink is not None else dmaster), dr


class PrepWeights(Function):
    """master ([E,] A, B, k,k,k) fp32 (+ routing r (Bw, E)) -> (wk_fwd, wk_dgrad).

    wk_fwd [Bw, taps, Cout, Cin] feeds the forward gather, wk_dgrad [Bw, taps, Cin, Cout]
    the data-gradient gather; dtypes are chosen by the conv algo (fp32 direct / bf16 MFMA).
    (Stand-alone form, used by the op tests; the model runs ConvLayer, which keeps the fp32 weight
    gradient inside one autograd node.)
    """

    @staticmethod
    def forward(ctx, master, r, transposed, fwd_dtype, dgrad_dtype):
        ctx.set_materialize_grads(False)
        ctx.p_master = master
        wk_f, wk_d, rr, ctx.meta = _prep_fwd(master, r, transposed, fwd_dtype, dgrad_dtype)
        if wk_d is not None:
            ctx.mark_non_differentiable(wk_d)
        ctx.save_for_backward(master, rr)
        return wk_f, wk_d

    @staticmethod
    def backward(ctx, dwk_f, _dwk_d):
        if dwk_f is None:
            return None, None, None, None, None
        master, rr = ctx.saved_tensors
        dmaster, dr = _prep_bwd(dwk_f.contiguous().float(), master, rr, ctx.meta, ctx.p_master)
        return dmaster, dr, None, None, None


class Routing(Function):
    """CondConv routing + per-sample bias mix in one launch each way:
    r = sigmoid(cov @ Wr^T + br) (B, E);  bias_mix = r @ bias_e (B, Cout)."""

    @staticmethod
    def forward(ctx, cov, Wr, br, bias_e, pre=None):
        """pre: (r, bias_mix) already computed for this forward (PrepAhead) -- the node then only exists for its backward."""
        ctx.set_materialize_grads(False)
        assert cov.dtype == torch.float32 and cov.is_contiguous() and cov.is_cuda
        B, NC = cov.shape
        E, N = Wr.shape[0], bias_e.shape[1]
        if pre is not None:
            r, bm = pre[0].view_as(pre[0]), pre[1].view_as(pre[1])      # (fresh tensor objects: outputs of this node)
        else:
            r, bm = _f32((B, E), cov.device), _f32((B, N), cov.device)
            check(lib.coma_routing_fwd(ptr(cov), B, NC, ptr(Wr), ptr(br), E, ptr(bias_e), N, ptr(r), ptr(bm), L.stream()),
                  "coma_routing_fwd")
        ctx.save_for_backward(cov, r, bias_e)
        ctx.params = (Wr, br, bias_e)
        return r, bm

    @staticmethod
    def backward(ctx, dr, dbm):
        cov, r, bias_e = ctx.saved_tensors
        Wr, br, be = ctx.params
        B, NC = cov.shape
        E, N = r.shape[1], bias_e.shape[1]
        dev = cov.device
        scope = contextlib.nullcontext()
        if WgradSide.dirty:      # dr / dbm of this step were produced on the side stream (ConvLayer.backward)
            if WgradSide.usable(dev, (Wr, br, be)):
                WgradSide.keep.extend(t for t in (cov, r, dr, dbm) if t is not None)
                scope = torch.cuda.stream(SidePrep.stream(dev))
            else:
                SidePrep.fence(dev)
        with scope:
            return Routing._backward(cov, r, bias_e, Wr, br, be, dr, dbm, B, NC, E, N, dev) + (None,)

    @staticmethod
    def _backward(cov, r, bias_e, Wr, br, be, dr, dbm, B, NC, E, N, dev):
        sinks = GradSink.slots((Wr, br, be))      # one kernel writes all three: one flush, then three marks
        dWr = sinks[0] if sinks[0] is not None else _f32((E, NC), dev)
        dbr = sinks[1] if sinks[1] is not None else _f32((E,), dev)
        dbe = sinks[2] if sinks[2] is not None else _f32((E, N), dev)
        dr = dr.contiguous() if dr is not None else None
        dbm = dbm.contiguous() if dbm is not None else None
        check(lib.coma_routing_bwd(ptr(cov), B, NC, ptr(r), E, ptr(bias_e), N, ptr(dr), ptr(dbm), ptr(dWr), ptr(dbr), ptr(dbe),
                                   L.stream()), "coma_routing_bwd")
        return (None, None if sinks[0] is not None else dWr, None if sinks[1] is not None else dbr,
                None if sinks[2] is not None else dbe)


# --------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------
def _desc(ksize, stride, form, per_sample, algo=0):
    return L.ConvDesc(ksize, stride, (ksize - 1) // 2, form, int(per_sample), algo)


def conv_out_grid(shape, ksize, stride, transposed):
    B, D, H, W, _ = shape
    if transposed:
        return B, D * stride, H * stride, W * stride
    p = (ksize - 1) // 2
    f = lambda n: (n + 2 * p - ksize) // stride + 1
    return B, f(D), f(H), f(W)


def pick_algo(x_shape, x_dtype, n_out, ksize, stride, transposed, per_sample, device, algo=0):
    """(algo_fwd, algo_dgrad) the library resolves for this layer."""
    B, Do, Ho, Wo = conv_out_grid(x_shape, ksize, stride, transposed)
    xt = L.Tensor(None, L.dtype_code(x_dtype), x_shape[0], x_shape[1], x_shape[2], x_shape[3], x_shape[4], x_shape[4], 0)
    yt = L.Tensor(None, L.dtype_code(x_dtype), B, Do, Ho, Wo, n_out, n_out, 0)
    form = 1 if transposed else 0
    a_f = lib.coma_conv_pick_algo(_desc(ksize, stride, form, per_sample, algo), xt, yt)
    a_d = lib.coma_conv_pick_algo(_desc(ksize, stride, 1 - form, per_sample, algo), yt, xt)
    return a_f, a_d


def _scratch(nbytes, device, lane=0):
    """(buffer, zeroed flag): the convolution's scratch as a private zeroed arena slice when the step's arena is armed
    (the library then skips its memset), else the shared stream-ordered workspace (of the side stream: lane 1)."""
    if nbytes > 0:
        z = ZeroArena.take(nbytes, device)
        if z is not None:
            return z, L.ZEROED_WS
    return workspace(nbytes, device, lane), 0


def _conv_fwd(x, wk_f, bias, ksize, stride, form, per_sample, algo, out, norm):
    """One forward launch; with `norm` (the mode of the normalisation that follows: L.NORM_BATCH / L.NORM_INSTANCE, or a
    tuple starting with it) its fp64 statistics record sums[G, C, 2] = {sum, sumsq} comes out of the same pass:
    returns (y, sums) then, else (y, None)."""
    B, Do, Ho, Wo = conv_out_grid(x.shape, ksize, stride, form == 1)
    n = wk_f.shape[2]
    y = out.t if out is not None else _new((B, Do, Ho, Wo, n), x.dtype, x.device)
    b = bias.contiguous().float() if bias is not None else None
    d = _desc(ksize, stride, form, per_sample, algo)
    tag = (tuple(x.shape), n, ksize, stride, form)
    cx, cy = ct(x), ct(y)
    kind = conv_class(_ALGO_NAMES[lib.coma_conv_pick_algo(d, cx, cy)] if KernelTimer.enabled else "", x.shape[4], n)
    ws, zf = _scratch(lib.coma_conv_fwd_ws_bytes(d, cx, cy), x.device)      # split-K scratch of the deep layers
    if norm is None:
        KernelTimer.run("conv_fwd", kind, conv_flops(x.shape, y.shape, ksize, stride),
                        lambda: check(lib.coma_conv_fwd_ws(d, cx, ptr(wk_f), L.dtype_code(wk_f.dtype), ptr(b), cy, ptr(ws),
                                                           ws.numel(), zf, L.stream()), "coma_conv_fwd"), tag=tag)
        return y, None
    mode = norm[0] if isinstance(norm, (tuple, list)) else norm
    G = B if mode == L.NORM_INSTANCE else 1
    sums = stat_record(G, n, 2, x.device)
    KernelTimer.run("conv_fwd", kind, conv_flops(x.shape, y.shape, ksize, stride),
                    lambda: check(lib.coma_conv_fwd_norm_stats(d, ct(x), ptr(wk_f), L.dtype_code(wk_f.dtype), ptr(b), cy,
                                                               mode, ptr(sums), ptr(ws), ws.numel(), zf, L.stream()),
                                  "coma_conv_fwd_norm_stats"), tag=tag)
    return y, sums


def _conv_bwd(x, wk_d, dy, ksize, stride, form, per_sample, algo, wshape, need_dx, need_dw, bias_mode, p_bias, fork=None,
              side=False):
    """-> dx, dwk (fp32, kernel layout), dbias (None when absent, written through, or identically zero).
    bias_mode: 0 no bias, 1 reduce dy over the voxels, 2 the bias feeds a mean-removing normalisation (its gradient is
    identically zero: exact zeros are returned instead of a reduction of rounding noise).
    fork: the input's GradFork -- the data gradient is written / accumulated into the shared buffer and dx is None.
    side: the weight gradient is launched on the side stream (WgradSide), forked at the ENTRY of this call -- it does not
    wait for the data gradient; dwk / dbias are then valid on that stream only."""
    dx = dwk = dbias = None
    s = L.stream()
    tag = (tuple(x.shape), dy.shape[4], ksize, stride, form)
    entry = None
    if side and need_dw:
        # forked HERE, not behind the data gradient: measured 18.3-18.6 ms per step against 19.1-19.2 (= no gain at all) when
        # the weight gradient waits for its layer's data gradient
        entry = torch.cuda.Event()
        entry.record(torch.cuda.current_stream(x.device))
    if need_dx:
        assert wk_d is not None, "data gradient requested but dgrad weights were not prepared"
        dd, cdy_ = _desc(ksize, stride, 1 - form, per_sample, algo), ct(dy)
        flags, shared = 0, False
        if fork is not None and fork.buf is None:
            dx = fork.buf = _new(x.shape, x.dtype, x.device)        # first consumer: "="
            shared = True
        elif fork is not None and lib.coma_conv_accumulate_ok(dd, cdy_, ct(fork.buf)):
            dx, flags, shared = fork.buf, L.ACCUMULATE, True         # later consumer: "+=" in the kernel's epilogue
        else:
            dx = _new(x.shape, x.dtype, x.device)
        cdx = ct(dx)
        wsd, zfd = _scratch(lib.coma_conv_fwd_ws_bytes(dd, cdy_, cdx), x.device)
        KernelTimer.run("conv_dgrad", conv_class(_ALGO_NAMES[lib.coma_conv_pick_algo(dd, cdy_, cdx)] if KernelTimer.enabled else "",
                                                   dy.shape[4], x.shape[4]),
                        conv_flops(dy.shape, dx.shape, ksize, stride),
                        lambda: check(lib.coma_conv_fwd_ws(dd, cdy_, ptr(wk_d), L.dtype_code(wk_d.dtype), None, cdx, ptr(wsd),
                                                           wsd.numel(), zfd | flags, s), "coma_conv_fwd(dgrad)"), tag=tag)
        if shared:
            dx = None
    if need_dw:
        lane = 0
        scope = contextlib.nullcontext()
        if entry is not None:
            scope, lane = torch.cuda.stream(WgradSide.begin(x.device, entry, x, dy)), 1
        with scope:
            s = L.stream()
            d = _desc(ksize, stride, form, per_sample, algo)
            cx, cdy = ct(x), ct(dy)
            # scratch: the replicas the kernels merge into must be zero -- a private arena slice of just that size when the
            # step's arena is armed (a bias reduction writes its partial rows into the scratch first: shared workspace then)
            nzs = lib.coma_conv_wgrad_zs_bytes(d, cx, cdy) if bias_mode != 1 else 0
            ws, zf = _scratch(nzs, x.device, lane) if nzs > 0 else (None, 0)
            if not zf:
                ws = workspace(lib.coma_conv_wgrad_ws_bytes(d, cx, cdy), x.device, lane)
            nel = 1
            for k_ in wshape:
                nel *= k_
            dwk = ZeroArena.take(4 * nel, x.device, torch.float32)
            if dwk is not None:
                dwk, zf = dwk.view(wshape), zf | L.ZEROED_OUT
            else:
                dwk = _f32(wshape, x.device)
            bshape = (x.shape[0], wshape[2]) if per_sample else (wshape[2],)
            dbias_k = None
            if bias_mode == 1:
                sink = GradSink.slot(p_bias)
                dbias_k = sink if sink is not None else _f32(bshape, x.device)
                dbias = None if sink is not None else dbias_k
            elif bias_mode == 2 and not per_sample:   # (per-sample: the routing node receives None = zero)
                sink = GradSink.slot(p_bias)          # the flat gradient buffer was zeroed by zero_grad(): nothing to write
                dbias = None if sink is not None else torch.zeros(bshape, dtype=torch.float32, device=x.device)
            walgo = conv_class(_ALGO_NAMES[lib.coma_conv_wgrad_algo(d, cx, cdy)] if KernelTimer.enabled else "", x.shape[4], dy.shape[4])
            KernelTimer.run("conv_wgrad", walgo, conv_flops(x.shape, dy.shape, ksize, stride),
                            lambda: check(lib.coma_conv_wgrad(d, cx, cdy, ptr(dwk), ptr(dbias_k), ptr(ws), ws.numel(), zf, s),
                                          "coma_conv_wgrad"), tag=tag)
    return dx, dwk, dbias


class Conv(Function):
    """y = conv(x) on prepared kernel-layout weights (stand-alone form, used by the op tests).  With `norm` = the mode
    of the BatchNorm(train)/InstanceNorm that follows, its statistics record is produced in the same pass
    (epilogue-fused where the kernel supports it): returns (y, sums[G, C, 2])."""

    @staticmethod
    def forward(ctx, x, wk_f, wk_d, bias, ksize, stride, transposed, per_sample, algo, out, norm=None):
        ctx.set_materialize_grads(False)
        ctx.p_bias = bias if not per_sample else None
        form = 1 if transposed else 0
        y, sums = _conv_fwd(x, wk_f, bias, ksize, stride, form, per_sample, algo, out, norm)
        ctx.save_for_backward(x, wk_d)
        ctx.meta = (ksize, stride, form, per_sample, algo, bias is not None, tuple(wk_f.shape))
        if norm is None:
            return y
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return (None,) * 11
        x, wk_d = ctx.saved_tensors
        ksize, stride, form, per_sample, algo, has_bias, wshape = ctx.meta
        dx, dwk, dbias = _conv_bwd(x, wk_d, dy, ksize, stride, form, per_sample, algo, wshape, ctx.needs_input_grad[0],
                                   ctx.needs_input_grad[1], 1 if has_bias else 0, ctx.p_bias)
        return dx, dwk, None, dbias, None, None, None, None, None, None, None


class ConvLayer(Function):
    """The model's convolution node: expert mix / re-layout of the fp32 master weights + the convolution, forward and
    backward, as ONE autograd node -- the fp32 kernel-layout weight gradient never crosses an autograd edge (where it
    would be cast to the bf16 dtype of the prepared weights and back).
    master ([E,] A, B, k,k,k); r (B, E) routing or None; bias (Cout,) / per-sample (B, Cout) or None."""

    @staticmethod
    def forward(ctx, x, master, r, bias, ksize, stride, transposed, algo, out, norm, fwd_dtype, dgrad_dtype, bias_zero_grad,
                prepped=None):
        """prepped: (wk_f, wk_d, rr, pmeta) from PrepAhead (persistent buffers, prepared at the start of this forward)."""
        ctx.set_materialize_grads(False)
        per_sample = r is not None
        form = 1 if transposed else 0
        side = False
        ctx.ahead = False
        if prepped is not None:
            wk_f, wk_d, rr, pmeta = prepped
            side = None              # (neither the side-stream nor the in-line preparation below)
        elif SidePrep._on:
            E_ = master.shape[0] if per_sample else 1
            A_, B_ = master.shape[-5], master.shape[-4]
            taps_ = ksize ** 3
            cout_, cin_ = (B_, A_) if transposed else (A_, B_)
            Bw_ = r.shape[0] if per_sample else 1
            bufs = SidePrep.buffers(master, (Bw_, taps_, cout_, cin_), fwd_dtype, (Bw_, taps_, cin_, cout_), dgrad_dtype)
            if bufs is not None:
                with torch.cuda.stream(SidePrep.stream(x.device)):
                    wk_f, wk_d, rr, pmeta = _prep_fwd(master, r, transposed, fwd_dtype, dgrad_dtype, bufs)
                SidePrep.fence(x.device)
                side = True
        if side is False:
            if SidePrep._on:
                SidePrep.fence(x.device)      # (routing / bias mix of this layer were queued on the side stream)
            wk_f, wk_d, rr, pmeta = _prep_fwd(master, r, transposed, fwd_dtype, dgrad_dtype)
        y, sums = _conv_fwd(x, wk_f, bias, ksize, stride, form, per_sample, algo, out, norm)
        ctx.save_for_backward(x, wk_d, master, rr)
        ctx.fork = getattr(x, "_coma_fork", None)
        ctx.side = bool(side)
        if side and any(ctx.needs_input_grad):
            SidePrep._live += 1
        if prepped is not None and any(ctx.needs_input_grad):
            ctx.ahead = True
            PrepAhead.live += 1
        ctx.p_master, ctx.p_bias = master, (bias if not per_sample else None)
        # per-sample biases survive a BATCH norm (only their batch mean is removed); an instance norm removes them
        removed = norm is not None and (not per_sample or (norm[0] if isinstance(norm, (tuple, list)) else norm) == L.NORM_INSTANCE)
        bias_mode = 0 if bias is None else (2 if (bias_zero_grad and removed) else 1)
        ctx.meta = (ksize, stride, form, per_sample, algo, bias_mode, tuple(wk_f.shape), pmeta)
        if norm is None:
            return y
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            if ctx.ahead:
                PrepAhead.live = max(0, PrepAhead.live - 1)
            return (None,) * 14
        x, wk_d, master, rr = ctx.saved_tensors
        ksize, stride, form, per_sample, algo, bias_mode, wshape, pmeta = ctx.meta
        need_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        # weight gradient + expert scatter beside the data-gradient chain (WgradSide): when the master's (and a plain
        # bias's) gradient is written through to the flat buffer -- a per-sample bias's goes to the routing node, which
        # follows onto the side stream
        wside = need_dw and WgradSide.usable(x.device, (ctx.p_master, ctx.p_bias if bias_mode else None))
        dx, dwk, dbias = _conv_bwd(x, wk_d, dy, ksize, stride, form, per_sample, algo, wshape, ctx.needs_input_grad[0],
                                   need_dw, bias_mode, ctx.p_bias, ctx.fork, wside)
        dmaster = dr = None
        if ctx.side:
            SidePrep._live = max(0, SidePrep._live - 1)
        if ctx.ahead:
            PrepAhead.live = max(0, PrepAhead.live - 1)
        if wside:
            WgradSide.keep.extend((master, rr))
            with torch.cuda.stream(SidePrep.stream(x.device)):
                dmaster, dr = _prep_bwd(dwk, master, rr, pmeta, ctx.p_master)
            WgradSide.keep.extend(t for t in (dwk, dr, dbias) if t is not None)
        elif need_dw:
            if WgradSide.dirty:       # (a gradient that goes through autograd after all: nothing of it may overtake the side stream)
                SidePrep.fence(x.device)
            # the scatter to the experts runs on the side stream when its result is written through to the flat gradient
            # buffer (nothing on this stream reads it before the optimizer) and the routing node -- the only consumer of
            # dr -- itself lives on the side stream
            sunk = getattr(ctx.p_master, "_coma_sink", False) and ctx.p_master.grad is not None and id(ctx.p_master) not in GradSink.written
            if ctx.side and SidePrep.enabled and SidePrep.backward and GradSink.listener is None and sunk:
                SidePrep.follow(x.device)
                dwk.record_stream(SidePrep.stream(x.device))
                with torch.cuda.stream(SidePrep.stream(x.device)):
                    dmaster, dr = _prep_bwd(dwk, master, rr, pmeta, ctx.p_master)
            else:
                dmaster, dr = _prep_bwd(dwk, master, rr, pmeta, ctx.p_master)
        return (dx, dmaster, dr, dbias) + (None,) * 10


# --------------------------------------------------------------------------------------
# BatchNorm (train) / InstanceNorm + activation
# --------------------------------------------------------------------------------------
class NormAct(Function):
    """BatchNorm3d(train / eval) or InstanceNorm3d + activation.  `pre`: the fp64 statistics record sums[G, C, 2] the
    convolution that wrote `x` already produced; else the record is made here by one statistics pass.  The kernels
    derive mean / rstd from the record themselves; the forward apply also moves BatchNorm's running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, rmean, rvar, mode, act, momentum, eps, training, out, pre=None):
        ctx.params = (gamma, beta, slope)
        dev = x.device
        B, C = x.shape[0], x.shape[4]
        G = B if mode == L.NORM_INSTANCE else 1
        cx = ct(x)
        s = L.stream()
        use_batch_stats = training or mode == L.NORM_INSTANCE
        sums = mean = rstd = None
        if pre is not None:          # statistics already produced by the conv that wrote x
            sums = pre
        elif use_batch_stats:
            sums = stat_record(G, C, 2, dev)
            check(lib.coma_norm_stats(cx, mode, ptr(sums), s), "coma_norm_stats")
        else:
            mean = rmean.reshape(1, C).float().contiguous()
            rstd = torch.rsqrt(rvar.reshape(1, C).float() + eps).contiguous()
        upd = sums is not None and rmean is not None and training and mode == L.NORM_BATCH
        y = out.t if out is not None else _new(x.shape, x.dtype, dev)
        tb = float(x.numel() * x.element_size())          # algorithmic bytes: one read + one write of the tensor
        KernelTimer.run("norm_fwd", "hbm", (0.0, 2.0 * tb),
                        lambda: check(lib.coma_norm_act_fwd(cx, mode, ptr(sums), eps, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act,
                                                            ptr(slope), ptr(rmean) if upd else None, ptr(rvar) if upd else None,
                                                            momentum, ct(y), s), "coma_norm_act_fwd"),
                        name="norm_act_fwd_k<%s, %d>" % ("__bf16" if x.dtype == torch.bfloat16 else "float", 8 if (x.dtype == torch.bfloat16 and C % 8 == 0) else (4 if C % 4 == 0 else 1)))
        ctx.save_for_backward(x, sums, gamma, beta, slope)
        ctx.meta = (mode, act, use_batch_stats, eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sums, gamma, beta, slope = ctx.saved_tensors
        mode, act, use_batch_stats, eps = ctx.meta
        if not use_batch_stats:
            raise RuntimeError("backward through eval-mode BatchNorm is not part of the training path")
        dev = x.device
        B, C = x.shape[0], x.shape[4]
        G = B if mode == L.NORM_INSTANCE else 1
        dx = _new(x.shape, x.dtype, dev)
        sinks = GradSink.slots(ctx.params)      # one kernel chain writes all three: one flush, then three marks
        dgamma = sinks[0] if sinks[0] is not None else (_f32(C, dev) if gamma is not None else None)
        dbeta = sinks[1] if sinks[1] is not None else (_f32(C, dev) if beta is not None else None)
        dslope = sinks[2] if sinks[2] is not None else (_f32(1, dev) if slope is not None else None)
        bsums = stat_record(G, C, 3, dev)
        tb = float(x.numel() * x.element_size())          # algorithmic bytes: (x, dy) read twice + dx written once
        KernelTimer.run("norm_bwd", "hbm", (0.0, 5.0 * tb),
                        lambda: check(lib.coma_norm_act_bwd(ct(x), ct(dy), mode, ptr(sums), eps, ptr(gamma), ptr(beta), act, ptr(slope),
                                                            ct(dx), ptr(dgamma), ptr(dbeta), ptr(dslope), ptr(bsums), L.stream()),
                                      "coma_norm_act_bwd"),
                        name="norm_bwd_partial_k + norm_act_bwd_apply_k<%s>" % ("__bf16" if x.dtype == torch.bfloat16 else "float"))
        return (dx, None if sinks[0] is not None else dgamma, None if sinks[1] is not None else dbeta,
                None if sinks[2] is not None else dslope, None, None, None, None, None, None, None, None, None)


# --------------------------------------------------------------------------------------
# attention gate pieces, strided add, concat-free join
# --------------------------------------------------------------------------------------
class AddRelu(Function):
    @staticmethod
    def forward(ctx, a, b):
        out = _new(a.shape, a.dtype, a.device)
        check(lib.coma_add_relu_fwd(ct(a), ct(b), ct(out), L.stream()), "coma_add_relu_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        da = _new(out.shape, out.dtype, out.device)
        check(lib.coma_add_relu_bwd(ct(out), ct(dout), ct(da), L.stream()), "coma_add_relu_bwd")
        return da, da


class GateMul(Function):
    """out = x * psi (psi broadcast over channels), attn_unet_data_parallel.py:150."""

    @staticmethod
    def forward(ctx, x, psi, out):
        y = out.t if out is not None else _new(x.shape, x.dtype, x.device)
        check(lib.coma_gate_mul_fwd(ct(x), ct(psi), ct(y), L.stream()), "coma_gate_mul_fwd")
        ctx.save_for_backward(x, psi)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, psi = ctx.saved_tensors
        dx = _new(x.shape, x.dtype, x.device)
        dpsi = _new(psi.shape, psi.dtype, psi.device)
        check(lib.coma_gate_mul_bwd(ct(x), ct(psi), ct(dout), ct(dx), 0, ct(dpsi), L.stream()), "coma_gate_mul_bwd")
        return dx, dpsi, None


class GateFused(Function):
    """The attention gate behind its W_g / W_x convolutions (csrc/gate.hip; attn_unet_data_parallel.py:139-150), training mode:
        s = relu(BN_g(g1raw) + BN_x(x1raw));  psi = sigmoid(BN_psi(w_psi . s + b_psi));  att = x * psi
    two launches forward, three backward.  Returns (att, psi); psi is for inspection only (not differentiable)."""

    @staticmethod
    def forward(ctx, x, g1raw, x1raw, sums_g, sums_x, gam_g, bet_g, gam_x, bet_x, w_psi, b_psi, gam_p, bet_p,
                running, momentum, eps, out):
        ctx.set_materialize_grads(False)
        dev, dt = x.device, x.dtype
        B, D, H, W, C = x.shape
        F_ = g1raw.shape[4]
        rm_g, rv_g, rm_x, rv_x, rm_p, rv_p = running if running is not None else (None,) * 6
        eps_g, eps_x, eps_p = eps
        s_t = _new((B, D, H, W, F_), dt, dev)
        psi_raw = _new((B, D, H, W, 1), dt, dev)
        psi = _new((B, D, H, W, 1), dt, dev)
        sums_p = stat_record(1, 1, 2, dev)
        st = L.stream()
        wv = w_psi.reshape(-1)
        assert wv.is_contiguous() and wv.dtype == torch.float32 and wv.numel() == F_
        check(lib.coma_gate_mid_fwd(ct(g1raw), ct(x1raw), ptr(sums_g), eps_g, ptr(gam_g), ptr(bet_g), ptr(sums_x), eps_x,
                                    ptr(gam_x), ptr(bet_x), ptr(wv), ptr(b_psi), ptr(rm_g), ptr(rv_g), ptr(rm_x), ptr(rv_x),
                                    momentum, ct(s_t), ct(psi_raw), ptr(sums_p), st), "coma_gate_mid_fwd")
        att = out.t if out is not None else _new(x.shape, dt, dev)
        check(lib.coma_gate_apply_fwd(ct(x), ct(psi_raw), ptr(sums_p), eps_p, ptr(gam_p), ptr(bet_p), ptr(rm_p), ptr(rv_p),
                                      momentum, ct(psi), ct(att), st), "coma_gate_apply_fwd")
        ctx.save_for_backward(x, g1raw, x1raw, s_t, psi_raw, psi, sums_g, sums_x, sums_p, gam_g, gam_x, gam_p, bet_p, wv)
        ctx.params = (gam_g, bet_g, gam_x, bet_x, w_psi, gam_p, bet_p)
        ctx.b_psi = b_psi
        ctx.eps = eps
        ctx.fork = getattr(x, "_coma_fork", None)
        ctx.mark_non_differentiable(psi)
        return att, psi

    @staticmethod
    def backward(ctx, datt, _dpsi):
        n_in = 17
        if datt is None:
            return (None,) * n_in
        x, g1raw, x1raw, s_t, psi_raw, psi, sums_g, sums_x, sums_p, gam_g, gam_x, gam_p, bet_p, wv = ctx.saved_tensors
        eps_g, eps_x, eps_p = ctx.eps
        dev, dt = x.device, x.dtype
        B, D, H, W, C = x.shape
        F_ = g1raw.shape[4]
        st = L.stream()
        fork = ctx.fork
        if fork is not None and fork.buf is not None:
            dx, acc, shared = fork.buf, 1, True
        else:
            dx, acc, shared = _new(x.shape, dt, dev), 0, fork is not None
            if shared:
                fork.buf = dx
        dz = _new((B, D, H, W, 1), dt, dev)
        bs_p = stat_record(1, 1, 3, dev)
        check(lib.coma_gate_apply_bwd(ct(x), ct(psi), ct(psi_raw), ct(datt), ptr(sums_p), eps_p, ptr(gam_p), ptr(bet_p),
                                      ct(dx), acc, ct(dz), ptr(bs_p), st), "coma_gate_apply_bwd")
        sinks = GradSink.slots(ctx.params)          # one kernel writes all seven: one flush, then the marks
        fresh = lambda i, n: sinks[i] if sinks[i] is not None else _f32(n, dev)
        dgg, dbg, dgx, dbx = fresh(0, F_), fresh(1, F_), fresh(2, F_), fresh(3, F_)
        dw, dgp, dbp = fresh(4, F_), fresh(5, 1), fresh(6, 1)
        rec = stat_record(1, F_, 4, dev)
        dg1 = _new(g1raw.shape, dt, dev)
        dx1 = _new(x1raw.shape, dt, dev)
        check(lib.coma_gate_mid_bwd(ct(dz), ct(psi_raw), ct(s_t), ct(g1raw), ct(x1raw), ptr(sums_p), eps_p, ptr(gam_p), ptr(bs_p),
                                    ptr(sums_g), eps_g, ptr(gam_g), ptr(sums_x), eps_x, ptr(gam_x), ptr(wv), ptr(rec),
                                    ct(dg1), ct(dx1), ptr(dgg), ptr(dbg), ptr(dgx), ptr(dbx), ptr(dw), ptr(dgp), ptr(dbp), st),
              "coma_gate_mid_bwd")
        ret = lambda i, t: None if sinks[i] is not None else t
        # the psi convolution's bias is removed again by BN_psi: its gradient is identically zero (exact zeros, as for every
        # convolution bias under a normalisation: layers.Config.zero_bias_grad_under_norm)
        db_psi = None
        if ctx.b_psi is not None and GradSink.slot(ctx.b_psi) is None:
            db_psi = torch.zeros_like(ctx.b_psi)
        w_shape = ctx.params[4].shape
        return (None if shared else dx, dg1, dx1, None, None, ret(0, dgg), ret(1, dbg), ret(2, dgx), ret(3, dbx),
                ret(4, dw.view(w_shape)), db_psi, ret(5, dgp), ret(6, dbp), None, None, None, None)


class JoinSlices(Function):
    """The parts were written straight into channel slices of `buf`; hand the whole buffer on.
    Backward hands each producer its slice of the buffer's gradient (no copy either way); a part with a GradFork gets its
    slice installed as the fork's shared buffer, so that its other consumers accumulate straight into it."""

    @staticmethod
    def forward(ctx, buf_holder, *parts):
        ctx.set_materialize_grads(False)
        ctx.widths = [p.shape[4] for p in parts]
        ctx.forks = [getattr(p, "_coma_fork", None) for p in parts]
        return buf_holder.t

    @staticmethod
    def backward(ctx, dbuf):
        if dbuf is None:
            return (None,) * (1 + len(ctx.widths))
        outs, c0 = [], 0
        for w, f in zip(ctx.widths, ctx.forks):
            sl = dbuf[..., c0:c0 + w]
            outs.append(None if (f is not None and f.offer(sl)) else sl)
            c0 += w
        return (None, *outs)


class AddBcast(Function):
    """dst = a + b where `a` may have batch 1 (a learned prompt broadcast over the batch,
    attn_unet_data_parallel.py:640,651)."""

    @staticmethod
    def forward(ctx, a, b, out):
        y = out.t if out is not None else _new(b.shape, b.dtype, b.device)
        check(lib.coma_add(ct(a), ct(b), ct(y), L.stream()), "coma_add")
        ctx.a_bcast = a.shape[0] == 1 and b.shape[0] > 1
        ctx.a_shape = a.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        da = dy
        if ctx.a_bcast:
            da = _new(ctx.a_shape, dy.dtype, dy.device)
            check(lib.coma_batch_sum(ct(dy), ct(da), L.stream()), "coma_batch_sum")
        return da, dy, None


class Copy(Function):
    """Strided copy into a channel slice (out[..., k] of a concat buffer)."""

    @staticmethod
    def forward(ctx, a, out):
        check(lib.coma_add(ct(a), None, ct(out.t), L.stream()), "coma_add(copy)")
        return out.t

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class SpatialMean(Function):
    """AdaptiveAvgPool3d(1): (B, D, H, W, C) -> fp32 (B, C)  (attn_unet_data_parallel.py:538)."""

    @staticmethod
    def forward(ctx, x):
        cx = ct(x)
        out = _f32((x.shape[0], x.shape[4]), x.device)
        ws = workspace(lib.coma_norm_ws_bytes(cx), x.device)
        check(lib.coma_spatial_mean(cx, ptr(out), ptr(ws), ws.numel(), L.stream()), "coma_spatial_mean")
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        # only reached with a zero-weighted loss term upstream (criterions.py:562); tiny broadcast
        B, D, H, W, C = ctx.shape
        g = (dout / float(D * H * W)).to(ctx.dtype).view(B, 1, 1, 1, C).expand(ctx.shape)
        return g.contiguous()


class RoiPaint(Function):
    """cat((selected prompt, saliency volume, suvr volume)) of attn_unet_data_parallel.py:632-651."""

    @staticmethod
    def forward(ctx, pos_prompt, neg_prompt, roi, x, prior, roi_ids, abeta, out_dtype, need_pos=True, need_neg=True):
        B, D, H, W, _ = x.shape
        ctx.need = (need_pos, need_neg)
        out3 = _new((B, D, H, W, 3), out_dtype, x.device)
        check(lib.coma_roi_paint_fwd(ct(roi), ct(x), ptr(prior), ptr(roi_ids), roi_ids.numel(), ptr(abeta),
                                     ptr(pos_prompt), ptr(neg_prompt), ct(out3), L.stream()), "coma_roi_paint_fwd")
        ctx.save_for_backward(abeta)
        ctx.pshape = pos_prompt.shape
        return out3

    @staticmethod
    def backward(ctx, dout3):
        (abeta,) = ctx.saved_tensors
        dpos = torch.zeros(ctx.pshape, dtype=torch.float32, device=dout3.device)
        dneg = torch.zeros(ctx.pshape, dtype=torch.float32, device=dout3.device)
        check(lib.coma_roi_paint_bwd(ct(dout3), ptr(abeta), ptr(dpos), ptr(dneg), L.stream()), "coma_roi_paint_bwd")
        # a prompt no sample selected gets grad None, as in the reference (:639) -> AdamW skips it
        need_pos, need_neg = ctx.need
        return (dpos if need_pos else None), (dneg if need_neg else None), None, None, None, None, None, None, None, None


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
class RoiMSELoss(Function):
    """criterions.py:181-211 (voxel_wise=False): per-sample loss vector (B, 1)."""

    @staticmethod
    def forward(ctx, pred, gt, roi, roi_ids, roi_w):
        B = pred.shape[0]
        cp = ct(pred)
        loss, mm = _f32(B, pred.device), _f32(B, pred.device)
        ws = workspace(lib.coma_loss_ws_bytes(cp), pred.device)
        check(lib.coma_roi_mse_fwd(cp, ct(gt), ct(roi), ptr(roi_ids), ptr(roi_w), roi_ids.numel(), ptr(loss), ptr(mm),
                                   ptr(ws), ws.numel(), L.stream()), "coma_roi_mse_fwd")
        ctx.save_for_backward(pred, gt, mm)
        return loss.view(B, 1)

    @staticmethod
    def backward(ctx, gout):
        pred, gt, mm = ctx.saved_tensors
        g = gout.contiguous().float().view(-1)
        dpred = torch.empty_like(pred)
        check(lib.coma_roi_mse_bwd(ct(pred), ct(gt), ptr(g), ptr(mm), ct(dpred), L.stream()), "coma_roi_mse_bwd")
        return dpred, None, None, None, None


class L1Loss(Function):
    """Per-sample voxel MAE (the reference's evaluation metric, attn_unet_data_parallel.py:1215)."""

    @staticmethod
    def forward(ctx, pred, gt):
        B = pred.shape[0]
        cp = ct(pred)
        loss = _f32(B, pred.device)
        ws = workspace(lib.coma_loss_ws_bytes(cp), pred.device)
        check(lib.coma_l1_fwd(cp, ct(gt), ptr(loss), ptr(ws), ws.numel(), L.stream()), "coma_l1_fwd")
        ctx.save_for_backward(pred, gt)
        return loss.view(B, 1)

    @staticmethod
    def backward(ctx, gout):
        pred, gt = ctx.saved_tensors
        g = gout.contiguous().float().view(-1)
        dpred = torch.empty_like(pred)
        check(lib.coma_l1_bwd(ct(pred), ct(gt), ptr(g), ct(dpred), L.stream()), "coma_l1_bwd")
        return dpred, None


def adamw_(p, g, m, v, lr, beta1, beta2, eps, wd, step, step_dev=None):
    """In-place fused AdamW on flat fp32 buffers (step_dev: device int32 step counter, for graph replay)."""
    assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32
    check(lib.coma_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, wd, step, ptr(step_dev),
                         L.stream()), "coma_adamw")
