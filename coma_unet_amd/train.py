"""The training step of the reference's ``train_dp`` (attn_unet_data_parallel.py:779-912):
zero_grad -> forward -> RnC pseudo-batch (:842-845) -> criterion (:878) -> backward (:884)
-> [gradient all-reduce] -> AdamW step (:885).  Host-side string work (extract_id, holdout
filter), logging and the per-sample ``.item()`` bookkeeping of :892-910 are the caller's.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .optim import FusedAdamW
from .data_parallel import GradReducer, StreamedGradExchange


def make_optimizer(model, lr=1e-3, write_through=True, pad_to=1):
    """torch.optim.AdamW(model.parameters(), lr) of :736, fused.  ``write_through``: parameter gradients are
    written by the backward kernels straight into the optimizer's flat gradient buffer (see ops.GradSink).
    ``pad_to``: the world size when the step is sharded over ranks (StreamedGradExchange(sharded=True))."""
    dyn = [model.pos_dynamic_prompt, model.neg_dynamic_prompt] if not getattr(model, "static_prompts", False) else []
    return FusedAdamW(model.parameters(), lr=lr, dynamic=dyn, write_through=write_through, pad_to=pad_to)


def forward_loss(model, criterion, batch, global_rnc=False):
    """Returns (total, gen_loss_vec, weighted_pred_contra, weighted_ds_contra), model_outputs.
    global_rnc: under data parallelism evaluate the RnC term on the features / labels of ALL ranks
    (data_parallel.gather_batch) instead of per replica -- see data_parallel's module docstring."""
    mri, tau, roi, covars = batch["mri"], batch["tau"], batch["roi"], batch["covars"]
    outs = model(mri, covars.to(device=mri.device), roi_pred_dicts=batch["roi_pred_dicts"], sample_roi_mask=roi)
    pred, projected, final_repr = outs[0], outs[1], outs[2]
    feats = torch.vstack([projected[-1]])                                   # :842
    labels = torch.vstack([covars[:, -1].to(device=mri.device)])            # :843
    if global_rnc:
        from .data_parallel import gather_batch
        feats, labels = gather_batch(feats), gather_batch(labels.float())
    pos = torch.zeros_like(final_repr)                                      # :855 (fp16 zeros upstream)
    neg = torch.zeros_like(final_repr)                                      # :856
    losses = criterion(pred, tau, roi, (final_repr, pos, neg), (feats, labels))
    return losses, outs


def train_step(model, criterion, optimizer, batch, reducer: GradReducer = None, global_rnc=False):
    if reducer is not None and reducer.world > 1:
        # every rank must reduce the same tensors in the same order: with rank-local prompt selection the two
        # selectable prompts get grad=None on some ranks only (attn_unet_data_parallel.py:638-639) and the collective
        # sequences diverge (hang, or pos paired with neg)
        assert getattr(model, "static_prompts", False), "data parallel steps need a model built with static_prompts=True"
    optimizer.zero_grad()                                                   # :806
    streamed = isinstance(reducer, StreamedGradExchange)
    if streamed:
        reducer.begin()
    elif reducer is not None:
        reducer.reset()
    losses, outs = forward_loss(model, criterion, batch, global_rnc)
    losses[0].backward()                                                    # :884
    ops.SidePrep.join()          # expert-gradient scatters of the weight-preparation stream (ops.SidePrep)
    stepped = False
    if reducer is not None:
        stepped = bool(reducer.finish())       # (a sharded exchange also takes the optimizer step, on its shard)
    if not stepped:
        optimizer.step()                                                    # :885
    return losses, outs


class GraphedTrainStep:
    """The step captured in a hipGraph and replayed: ~450 kernel launches collapse into one graph launch, which
    removes the host launch gaps (the reference's step is launch-bound in the same way on its cuDNN path).
    One GPU: zero_grad + forward + loss + backward + fused AdamW are ONE graph.  Data parallel (`reducer`
    given): the graph holds zero_grad + forward + loss + backward; the bucketed RCCL SUM all-reduce of the
    flat gradient buffer and the one-kernel AdamW step follow it eagerly (no collective inside the graph).
    Requirements: a model built with ``static_prompts=True`` (no host read of the covariates), ROI priors
    passed as a (B,36,2) tensor, fixed shapes.  New data is copied into the static input buffers.
    """

    def __init__(self, model, criterion, optimizer, batch, warmup=3, reducer=None, prewarmed=False):
        """prewarmed: the caller has already taken >= 2 eager steps at these shapes with this optimizer (the flat layout
        exists, the workspaces are sized): capture without any warm-up step, so that no optimizer step is taken here
        (train_loop.train_dp, where every step must be a real one on real data)."""
        assert getattr(model, "static_prompts", False), "graph capture needs static_prompts=True"
        self.model, self.criterion, self.optimizer, self.reducer = model, criterion, optimizer, reducer
        self.batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        assert torch.is_tensor(self.batch["roi_pred_dicts"]), "pass ROI priors as a (B, 36, 2) device tensor"
        # a StreamedGradExchange enqueues its collectives through the C ABI on a forked side stream: the WHOLE step
        # (forward, backward, overlapped exchange, join, optimizer) is captured as one graph, as on one GPU
        self.in_graph = isinstance(reducer, StreamedGradExchange)
        if reducer is not None and not self.in_graph:
            reducer.overlap = False          # hooks do not fire under replay
            reducer.remove_hooks()
            optimizer.set_write_through(True)
        if prewarmed:
            assert optimizer.built, "prewarmed capture needs the optimizer's flat layout (two eager steps first)"
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(warmup, 2)):      # builds the flat optimizer layout, sizes workspaces
                    train_step(model, criterion, optimizer, self.batch, reducer)
            torch.cuda.current_stream().wait_stream(side)
        from ._lib import pin_workspace
        self._ws = pin_workspace(self.batch["mri"].device)     # the graph bakes this buffer's address in: keep it alive and in place
        self._capture()

    def _hyper(self):
        g = self.optimizer.param_groups[0]
        return (float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"]))

    def _capture(self):
        """(Re)capture.  The optimizer's hyper-parameters are kernel arguments frozen into the graph, so a change of
        learning rate (ReduceLROnPlateau, attn_unet_data_parallel.py:737,921) triggers one re-capture on the next call."""
        model, criterion, optimizer, reducer = self.model, self.criterion, self.optimizer, self.reducer
        torch.cuda.synchronize()
        self._captured_hyper = self._hyper()
        self.watch = None
        if reducer is not None and not self.in_graph and hasattr(reducer, "graph_watch"):
            self.watch = reducer.graph_watch(self.batch["mri"].device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may query events while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            if reducer is None or self.in_graph:
                self.losses, self.outputs = train_step(model, criterion, optimizer, self.batch, reducer)
            else:
                optimizer.zero_grad()
                self.losses, self.outputs = forward_loss(model, criterion, self.batch)
                if self.watch is not None:       # external events behind every completed gradient bucket (GraphBucketWatch)
                    self.watch.begin()
                try:
                    self.losses[0].backward()
                finally:
                    if self.watch is not None:
                        self.watch.end()
                ops.SidePrep.join()
                ops.ZeroArena.end_step()      # the clearing memset of the step's zeroed arena belongs INSIDE the replayed part
        if reducer is None or self.in_graph:
            # capture executes nothing: undo the host-side step count train_step's optimizer.step() just added
            optimizer._flat_step -= 1

    def load(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.batch[k].copy_(v, non_blocking=True)

    def __call__(self, batch=None):
        if batch is not None:
            self.load(batch)
        whole = self.reducer is None or self.in_graph
        if whole and self._hyper() != self._captured_hyper:
            self._capture()
        self.graph.replay()
        if whole:
            self.optimizer._flat_step += 1
        else:
            self.reducer.reduce_flat_and_step(watch=self.watch)
        return self.losses, self.outputs


class InputStager:
    """Host -> device input staging beside the replayed step (SURVEY.md section 8 f-4 meets the step): the NEXT batch's
    volumes are copied from pinned host buffers into a device staging set on a side HIP stream while the current step's
    graph replays; `run()` then moves them into the graph's static input buffers with device-to-device copies (3 x 17 MB
    at 128^3: ~30 us) on the compute stream and replays.  The reference's loader hands `train_dp` pageable CPU tensors
    (VolumeDataset_ADNI_A4_combined.py:91) and `.cuda()`s them synchronously in front of every step
    (attn_unet_data_parallel.py:789-804); this takes that copy off the critical path."""

    def __init__(self, step: GraphedTrainStep):
        self.step = step
        dev = step.batch["mri"].device
        self.stream = torch.cuda.Stream(device=dev)
        self.dev = {k: torch.empty_like(v) for k, v in step.batch.items() if torch.is_tensor(v)}
        self.pinned = {}
        self.ready = None
        self.consumed = None

    def submit(self, host_batch):
        """Queue the copies of the next batch (CPU tensors, pinned or pageable -- pageable ones go through a pinned
        buffer of this object; device tensors are taken as they are)."""
        if self.ready is not None:
            raise RuntimeError("InputStager.submit: the previous batch has not been run yet")
        with torch.cuda.stream(self.stream):
            if self.consumed is not None:
                self.stream.wait_event(self.consumed)          # the previous batch has left the staging set
            for k, dst in self.dev.items():
                src = host_batch[k]
                if not torch.is_tensor(src):
                    continue
                if src.device.type == "cpu" and not src.is_pinned():
                    buf = self.pinned.get(k)
                    if buf is None or buf.shape != src.shape or buf.dtype != src.dtype:
                        buf = self.pinned[k] = torch.empty(src.shape, dtype=src.dtype).pin_memory()
                    self.stream.synchronize()                  # (the previous copy out of this pinned buffer is done)
                    buf.copy_(src)
                    src = buf
                dst.copy_(src, non_blocking=True)
            self.ready = torch.cuda.Event()
            self.ready.record(self.stream)

    def run(self):
        """Replay the step on the submitted batch."""
        assert self.ready is not None, "InputStager.run: submit() a batch first"
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready)
        self.step.load(self.dev)
        self.consumed = torch.cuda.Event()
        self.consumed.record(cur)
        self.ready = None
        return self.step()
