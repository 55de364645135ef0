"""Fused AdamW on flat fp32 buffers (torch.optim.AdamW semantics; the reference builds
``torch.optim.AdamW(model.parameters(), lr)``, attn_unet_data_parallel.py:736).

Parameters that receive a gradient in the first step are packed into ONE flat parameter
buffer and ONE flat gradient buffer (``p.data`` / ``p.grad`` become views), so a step is a
single HBM-bound kernel launch and the data-parallel gradient exchange can all-reduce
contiguous buckets of the same buffer.  Parameters whose gradient is ``None`` are skipped
exactly as torch does (no decay, no moment update, no step count): the reference model has
~75 such tensors ("aux model heads never used in forward").
"""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, dynamic=(), write_through=False,
                 pad_to=1):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        assert len(self.param_groups) == 1, "one parameter group (the reference uses one)"
        self._dynamic = {id(p) for p in dynamic}   # may or may not get a gradient, step to step
        self.flat_p = self.flat_g = self.flat_m = self.flat_v = None
        self._flat_params = []
        self._flat_step = 0
        # write_through: backward kernels store parameter gradients straight into the flat buffer's slots
        # (ops.GradSink) -- no per-parameter AccumulateGrad launch.  Off when a GradReducer drives its
        # all-reduce from post-accumulate-grad hooks (those do not fire for sunk gradients).
        self.write_through = write_through
        self.pad_to = max(1, int(pad_to))   # flat buffers padded to a multiple of this (the world size, for sharded steps)
        # set by data_parallel.StreamedGradExchange(sharded=True): each rank then steps only its own sub-slices of the moment
        # buffers, and state_dict() must re-assemble them first (a COLLECTIVE: every rank calls state_dict())
        self.shard_exchange = None
        # sparse zero_grad (write-through only): the backward kernels overwrite ("=") the gradient slots of the large
        # tensors every step, so zero_grad() only has to clear what lies between them -- one small launch instead of a
        # memset of the whole buffer (647 MB, ~90 us).  Built from what the previous step's kernels actually wrote and
        # checked again at every step (a tensor that was NOT written after all is zeroed before the update reads it).
        self._zero_tab = None          # (device int64 [n, 2] table, n, longest range)
        self._zero_big = ()            # ids of the parameters zero_grad() skips
        self.sparse_zero = True

    # -- layout -------------------------------------------------------------------------
    def _build(self):
        ps = [p for p in self.param_groups[0]["params"] if p.grad is not None and id(p) not in self._dynamic]
        assert ps, "no parameter has a gradient"
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        n = (n + self.pad_to - 1) // self.pad_to * self.pad_to
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self._offsets = {}
        for p in ps:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            self.flat_g[off:off + k].copy_(p.grad.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            self._offsets[id(p)] = (off, k)
            p._coma_sink = bool(self.write_through)
            off += k
        self._flat_params = ps
        self._flat_ids = {id(p) for p in ps}
        # a state_dict loaded BEFORE the layout existed (the reference's resume order: fresh optimizer ->
        # load_state_dict -> train, validation.py:276-281, attn_unet_data_parallel.py:729-733) parked its moments in
        # self.state[p]: move them into the flat buffers and carry the step count over
        for p in ps:
            st = self.state.pop(p, None)
            if st:
                off, k = self._offsets[id(p)]
                self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self._flat_step = max(self._flat_step, int(st["step"]))
        self._step_dev = torch.full((1,), self._flat_step, dtype=torch.int32, device=dev)

    @property
    def built(self):
        return self.flat_p is not None

    def set_write_through(self, on: bool):
        self.write_through = bool(on)
        for p in self._flat_params:
            p._coma_sink = self.write_through

    def zero_grad(self, set_to_none=True):
        ops.GradSink.begin_step()
        for p in self.param_groups[0]["params"]:
            if self.built and id(p) in self._flat_ids:
                continue
            p.grad = None
        if self.built:
            if self._zero_tab is not None and self.write_through and self.sparse_zero:
                tab, n, longest = self._zero_tab
                from ._lib import lib, check, stream
                check(lib.coma_zero_ranges(self.flat_g.data_ptr(), tab.data_ptr(), n, longest, stream()), "coma_zero_ranges")
            else:
                self.flat_g.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self.step_shards(None)

    @torch.no_grad()
    def step_shards(self, shards, advance=True, loose=True):
        """One AdamW step.  shards=None: the whole flat buffer (one launch).  shards=[(offset, length)]: only those slices
        of it -- the data-parallel step with a reduce-scattered gradient updates this rank's slices and all-gathers the
        parameters afterwards (data_parallel.StreamedGradExchange); the moments of the other slices are never touched here
        (they live, up to date, on the ranks that own them).  A step may also be taken slice by slice as the slices'
        gradients arrive (GradReducer.reduce_flat_and_step): advance=True on the first call only (the step count moves
        once), loose=True on one call only (parameters outside the flat buffer)."""
        ops.SidePrep.join()       # the side stream's expert-gradient scatters land in the flat gradient buffer
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = float(g["lr"]), g["betas"], g["eps"], g["weight_decay"]
        if not self.built:
            self._build()
        if advance:
            self._flat_step += 1
            self._step_dev += 1            # device-side copy: a captured step keeps counting under graph replay
            self._sparse_zero_bookkeeping()
        for off, k in ([(0, self.flat_p.numel())] if shards is None else shards):
            if k > 0:
                ops.adamw_(self.flat_p[off:off + k], self.flat_g[off:off + k], self.flat_m[off:off + k], self.flat_v[off:off + k],
                           lr, b1, b2, eps, wd, self._flat_step, self._step_dev)
        if not loose:
            return
        # backward is over: every slice of the step's zeroed arena is dead -- ONE memset clears them.  Behind the AdamW launch,
        # not in front of it: with ~170 MB of freshly dirtied lines ahead of it the 2.7 GB AdamW sweep measured 1.10 ms
        # instead of 0.85 ms.
        ops.ZeroArena.end_step()
        for p in g["params"]:
            if id(p) in self._flat_ids or p.grad is None:
                continue
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
            st["step"] += 1
            gr = p.grad.float().contiguous()
            ops.adamw_(p.data.view(-1), gr.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), lr, b1, b2, eps,
                       wd, st["step"])

    def _sparse_zero_bookkeeping(self):
        """After a backward: (1) any large tensor zero_grad() skipped that the kernels did NOT overwrite this step still
        holds last step's gradient -- clear it now, before the update reads it, and go back to full clears; (2) otherwise
        (re)build the table of what zero_grad() has to clear from what this step's kernels wrote."""
        if not (self.write_through and self.sparse_zero):
            self._zero_tab, self._zero_big = None, ()
            return
        written = ops.GradSink.written
        if self._zero_tab is not None:
            missing = [p for p in self._flat_params if id(p) in self._zero_big and id(p) not in written]
            if missing:
                for p in missing:
                    p.grad.zero_()
                self._zero_tab, self._zero_big, self.sparse_zero = None, (), False
            return
        big = [p for p in self._flat_params if id(p) in written and p.numel() >= (1 << 16)]
        if not big:
            return
        ranges, pos = [], 0
        for p in big:                                   # (flat order == _flat_params order)
            off, k = self._offsets[id(p)]
            if off > pos:
                ranges.append((pos, off - pos))
            pos = off + k
        n = self.flat_g.numel()
        if n > pos:
            ranges.append((pos, n - pos))
        if sum(r[1] for r in ranges) * 4 > n:           # not worth a table: most of the buffer needs clearing anyway
            return
        tab = torch.tensor(ranges if ranges else [(0, 0)], dtype=torch.int64, device=self.flat_g.device)
        self._zero_tab = (tab, len(ranges), max([r[1] for r in ranges], default=0))
        self._zero_big = {id(p) for p in big}

    # -- torch.optim.AdamW-compatible checkpoint format (attn_unet_data_parallel.py:946-952) ----
    def state_dict(self):
        """torch.optim.AdamW's format.  Under a sharded exchange the moments of this rank's sub-slices are current and the
        others stopped at the last unsharded step: they are all-gathered first (every rank must call state_dict(), like any
        collective; afterwards every rank holds the full, identical state)."""
        if self.shard_exchange is not None and self.built:
            self.shard_exchange.gather_state()
        params = self.param_groups[0]["params"]
        state = {}
        for i, p in enumerate(params):
            if self.built and id(p) in self._flat_ids:
                off, k = self._offsets[id(p)]
                state[i] = {"step": torch.tensor(float(self._flat_step)),
                            "exp_avg": self.flat_m[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.flat_v[off:off + k].view(p.shape).clone()}
            elif p in self.state and self.state[p]:
                st = self.state[p]
                state[i] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"].clone(),
                            "exp_avg_sq": st["exp_avg_sq"].clone()}
        pg = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        pg["params"] = list(range(len(params)))
        return {"state": state, "param_groups": [pg]}

    def load_state_dict(self, sd):
        params = self.param_groups[0]["params"]
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        # the flat layout exists only after the first backward: until then the moments wait in self.state[p]
        # (per-parameter AdamW entries) and _build() moves them into the flat buffers
        for i, st in sd["state"].items():
            p = params[int(i)]
            if self.built and id(p) in self._flat_ids:
                off, k = self._offsets[id(p)]
                self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self._flat_step = int(st["step"])
                self._step_dev.fill_(self._flat_step)
            else:
                self.state[p] = {"step": int(st["step"]), "exp_avg": st["exp_avg"].to(p.device).float().clone(),
                                 "exp_avg_sq": st["exp_avg_sq"].to(p.device).float().clone()}
        if self.built:      # parameters inside the flat buffer share one step count
            self._step_dev.fill_(self._flat_step)
