"""Host loops of the reference with the reference's signatures: ``train_dp`` (attn_unet_data_parallel.py:696-1034) and
``contrastive_test`` (:1129-1359), so that ``validation.py:158`` / ``:348`` work after the import swap of INTEGRATION.md.

Both are thin loops over this package's device path: ``train.train_step`` (zero_grad -> forward -> criterion -> backward
-> fused AdamW, :806-885) and ``metrics`` (one fused statistics kernel per batch instead of 36 x ~7 masked passes).
What the reference does around them on the host is kept where it decides NUMBERS (loss bookkeeping, the plateau
scheduler input, checkpoint contents, metric accumulation order -- including its quirks, each marked "as upstream") and
left out where it only draws pictures (``visualization_util`` is absent upstream, :34).

Differences that are interface, not arithmetic:
 * ROI-prior lookups: the reference reads two JSON files at hard-coded paths under os.getcwd() (:708-710, :1135);
   here ``roi_vecs_dict=`` (the kwarg ``contrastive_test`` already has upstream, :1132) is honoured by both loops and the
   JSON paths are only tried when it is absent.
 * ``data_util.filter_for_holdout`` / ``extract_id`` are host string work from a module outside the path
   (data_util.py:701-745): ``holdout_filter=`` / ``extract_id=`` kwargs replace them (defaults: keep everything /
   the same token rules restated below).
"""
from __future__ import annotations

import json
import logging
import os

import numpy as np
import torch
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import metrics as M
from .checkpoint import save_checkpoint
from .train import make_optimizer, train_step, GraphedTrainStep


def extract_id(path: str) -> str:
    """Sample id of a scan path (data_util.py:716-745: the token after a dataset marker, two tokens for scan-dated layouts)."""
    tok = path.split("/")
    rules = (("A4_processing", 2, 1), ("a4", 1, 1), ("ucsf", 1, 2), ("scan", 1, 2), ("processed", 1, 1), ("outputs", 1, 1),
             ("adni", 1, 2))
    for marker, skip, n in rules:
        if marker in tok:
            i = tok.index(marker) + skip
            return "/".join(tok[i:i + n])
    raise ValueError(f"no dataset marker in {path!r}")


def _load_json(path):
    with open(path) as f:
        return json.load(f)


def _roi_lookup(kwargs, candidates):
    if kwargs.get("roi_vecs_dict") is not None:
        return kwargs["roi_vecs_dict"]
    out = {}
    for c in candidates:
        if os.path.exists(c):
            out.update(_load_json(c))
    if not out:
        raise FileNotFoundError("no ROI-prior lookup: pass roi_vecs_dict={sample_id: {roi_name: {'loc', 'std'}}} or provide "
                                + " / ".join(candidates))
    return out


def _unpack(values):
    mri, tau, roi, abeta, tau_path = values
    abeta, covars = abeta
    return mri, tau, roi, abeta, covars, list(tau_path)


def _dev(t, device):
    return t.to(device) if torch.is_tensor(t) else t


def _dist_info(reducer):
    """(world, rank) of the data-parallel job this loop is part of (1, 0 without a reducer / process group)."""
    import torch.distributed as dist
    world = int(getattr(reducer, "world", 1)) if reducer is not None else 1
    rank = dist.get_rank() if (world > 1 and dist.is_initialized()) else 0
    return world, rank


def _same_shapes(a, b):
    return all((not torch.is_tensor(v)) or (k in b and torch.is_tensor(b[k]) and tuple(b[k].shape) == tuple(v.shape)
                                             and b[k].dtype == v.dtype) for k, v in a.items())


def train_dp(model, criterion, train_loader, validation_loader, epochs, lr, save_path="", cuda_id=0, pred_sample_file="",
             from_checkpoint=False, **kwargs):
    """attn_unet_data_parallel.py:696.  Returns the list of per-epoch average losses (the reference returns None; the
    list is additional).

    Build-side kwargs (none exists upstream; all optional):
     * ``graph`` ("auto" | True | False): replay the step from a captured hipGraph (train.GraphedTrainStep -- the path
       bench.py measures) when the model was built with ``static_prompts=True``: the first two batches run eagerly (they
       build the optimizer's flat layout and size the workspaces), the step is captured on the third and every later
       batch of the same shape is copied into the graph's static input buffers and replayed; batches of another shape
       (a short last batch) run eagerly.  "auto" = True when the model allows it.
     * ``reducer`` (data_parallel.GradReducer | StreamedGradExchange): one process per GPU.  The optimizer is the
       reducer's (its flat layout is what the reducer buckets); the plateau scheduler sees the loss of the GLOBAL
       batch (all-reduced sums), so every rank keeps the same learning rate; checkpoints, CSVs and validation are rank 0's
       (``optimizer.state_dict()`` is still called by every rank: it is a collective under a sharded exchange).
     * ``optimizer``: use this FusedAdamW instead of building one (``from_checkpoint=True`` requires it, as upstream)."""
    import torch.distributed as dist
    cwd = os.getcwd()
    fold = kwargs.get("fold_id")
    base = f"{cwd}/training_folds/adni_a4_first_scan_combined_folds/tau_prediction_lookups"
    lookup = _roi_lookup(kwargs, [f"{base}/formatted_fold_{fold}_predictions_for_train.json",
                                  f"{base}/formatted_fold_{fold}_predictions_for_test.json"])          # :708-710
    get_id = kwargs.get("extract_id", extract_id)
    keep = kwargs.get("holdout_filter", lambda *a: a)
    device = torch.device("cuda", cuda_id) if isinstance(cuda_id, int) else torch.device(cuda_id)
    criterion.gen_loss.batch_reduction = None                                                             # :717
    val_iter = kwargs.get("val_iter", 5)
    overfit_val_iter = 10
    checkpoint_iter = val_iter
    start_epoch = 0
    reducer = kwargs.get("reducer")
    world, rank = _dist_info(reducer)
    if from_checkpoint:                                                                                   # :729-733
        optimizer = kwargs["optimizer"]
        start_epoch = kwargs["start_epoch"]
        scheduler = kwargs["scheduler"] if kwargs.get("scheduler") is not None else \
            ReduceLROnPlateau(optimizer, "min", patience=5, factor=0.2)
    else:                                                                                                 # :736-737
        if kwargs.get("optimizer") is not None:
            optimizer = kwargs["optimizer"]
        elif reducer is not None:
            optimizer = reducer.opt           # the reducer buckets THIS optimizer's flat gradient buffer
        else:
            optimizer = make_optimizer(model, lr)
        for g in optimizer.param_groups:
            g["lr"] = lr
        scheduler = ReduceLROnPlateau(optimizer, "min", patience=5)
    if reducer is not None:
        assert reducer.opt is optimizer, "train_dp(reducer=...) must step the optimizer the reducer was built on"
    use_graph = kwargs.get("graph", "auto")
    if use_graph == "auto":
        use_graph = bool(getattr(model, "static_prompts", False))
    if use_graph:
        assert getattr(model, "static_prompts", False), "train_dp(graph=True) needs a model built with static_prompts=True"
    graphed, eager_done = None, 0
    # graph mode: the whole loop runs on a side stream, like GraphedTrainStep's own warm-up -- the eager steps that precede
    # the capture must not have run on the (legacy) default stream (torch.cuda.graph's documented requirement; a capture
    # right behind default-stream steps crashed in capture_end on ROCm 7.2)
    import contextlib
    scope = contextlib.ExitStack()
    if use_graph and device.type == "cuda":
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        scope.enter_context(torch.cuda.stream(side))
        scope.callback(lambda: torch.cuda.current_stream(device).wait_stream(side))
    with scope:
        return _train_epochs(model, criterion, train_loader, validation_loader, epochs, save_path, cuda_id, pred_sample_file,
                             kwargs, lookup, get_id, keep, device, optimizer, scheduler, reducer, world, rank, use_graph,
                             start_epoch, val_iter, overfit_val_iter, checkpoint_iter)


def _train_epochs(model, criterion, train_loader, validation_loader, epochs, save_path, cuda_id, pred_sample_file, kwargs, lookup,
                  get_id, keep, device, optimizer, scheduler, reducer, world, rank, use_graph, start_epoch, val_iter,
                  overfit_val_iter, checkpoint_iter):
    import torch.distributed as dist
    graphed, eager_done = None, 0
    epoch_avg_losses = []
    hist = {k: [] for k in ("mae", "rse", "rrmse", "ssim", "mape", "avg_corr", "roi_maes", "roi_mapes", "roi_wrrmses",
                            "roi_corrs", "roi_rses")}
    best_mape, best_avg_corr = float("inf"), -float("inf")
    loss = None
    for epoch in range(start_epoch, epochs):
        model.train(True)
        epoch_loss, num_samples = 0.0, 0
        epoch_gen = epoch_pred = epoch_ds = 0.0
        epoch_pos = epoch_neg = 0.0
        n_pos = n_neg = 0
        for batch_idx, batch_data in enumerate(train_loader):                                             # :779
            anchor = keep(*batch_data[0])
            if isinstance(anchor, int):
                continue
            mri, tau, roi, abeta, covars, tau_path = _unpack(anchor)
            priors = [lookup[get_id(p)] for p in tau_path]                                                # :809-810
            batch = dict(mri=_dev(mri, device), tau=_dev(tau, device), roi=_dev(roi, device), covars=_dev(covars, device),
                         roi_pred_dicts=priors)
            if use_graph:
                batch["roi_pred_dicts"] = model._priors(priors, batch["mri"].shape[0], device)            # (B, 36, 2) table
            if use_graph and graphed is None and eager_done >= 2 and optimizer.built:
                graphed = GraphedTrainStep(model, criterion, optimizer, batch, reducer=reducer, prewarmed=True)
            if graphed is not None and _same_shapes(graphed.batch, batch):
                (loss, gen_loss, pred_contra, ds_contra), outs = graphed(batch)                           # load + replay
            else:
                (loss, gen_loss, pred_contra, ds_contra), outs = train_step(model, criterion, optimizer, batch, reducer)   # :806-885
                eager_done += 1
            epoch_loss += loss.item()                                                                     # :892
            gl = gen_loss.detach().reshape(-1).tolist()                                                   # :901-910 (one host read, not B)
            ds = float(ds_contra)
            epoch_gen += sum(gl)
            epoch_pred += float(pred_contra)
            epoch_ds += ds
            num_samples += outs[0].size(0)
            for b, a in enumerate(torch.as_tensor(abeta).reshape(-1).tolist()):
                if a == 1:
                    epoch_pos += gl[b] + ds
                    n_pos += 1
                elif a == 0:
                    epoch_neg += gl[b] + ds
                    n_neg += 1
        if world > 1:
            # the plateau scheduler must see the same number on every rank, or the replicas' learning rates drift apart:
            # the loss of the GLOBAL batch (what the reference's single process would have accumulated)
            tot = torch.tensor([epoch_loss, float(num_samples)], dtype=torch.float64,
                               device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            epoch_loss, num_samples = float(tot[0]), int(tot[1])
        if num_samples == 0:
            continue
        scheduler.step(epoch_loss / num_samples)                                                          # :921
        epoch_avg_losses.append(epoch_loss / num_samples)
        if rank == 0:
            logging.info(f"epoch {epoch}: avg loss {epoch_loss / num_samples:.6f} (gen {epoch_gen / num_samples:.6f}, "
                         f"pred-contra {epoch_pred / num_samples:.6f}, ds-contra {epoch_ds / num_samples:.6f}; "
                         f"pos {epoch_pos / max(n_pos, 1):.6f}, neg {epoch_neg / max(n_neg, 1):.6f})")
        if save_path:                                                                                     # :943-955
            # every rank assembles the dict (FusedAdamW.state_dict() all-gathers sharded moments), rank 0 writes it
            save_checkpoint(save_path, epoch, model, optimizer, loss, scheduler, checkpoint_iter=checkpoint_iter, write=rank == 0)
        if epoch % val_iter == 0 and validation_loader is not None and rank == 0:                         # :957-1017
            with torch.no_grad():
                val_save = os.path.join(save_path, f"{epoch}_output_samples") if save_path else ""
                if val_save:
                    os.makedirs(val_save, exist_ok=True)
                res = contrastive_test(model, validation_loader, criterion.gen_loss.roi_indices, criterion.gen_loss.roi_weights,
                                       save_path=val_save, cuda_id=cuda_id, pred_sample_file=pred_sample_file, **kwargs)
                general = res[0]
                mae, mape, rse, rrmse, ssim_error, roi_maes, roi_mapes, roi_rses, roi_wrrmses, roi_corr = general[:10]
                record_results(criterion, save_path, hist, epoch, mae, mape, rse, rrmse, ssim_error, roi_maes, roi_mapes,
                               roi_rses, roi_wrrmses, roi_corr)
                if not criterion.gen_loss.voxel_wise:                                                     # :985-991
                    new_w = criterion.gen_loss.calculate_new_weights(roi_mapes / 100, with_update=True)
                    logging.info(f"Updated weights: {new_w}")
            if float(mape) < best_mape:
                best_mape = float(mape)
                logging.info(f"Lowest MAPE so far: Epoch {epoch}")
            if float(np.nanmean(roi_corr)) > best_avg_corr:
                best_avg_corr = float(np.nanmean(roi_corr))
                logging.info(f"Highest ROI Averaged Correlations so far: Epoch {epoch}")
        if epoch != 0 and epoch > 29 and epoch % overfit_val_iter == 0 and rank == 0:                     # :1019-1034
            with torch.no_grad():
                res = contrastive_test(model, train_loader, criterion.gen_loss.roi_indices, criterion.gen_loss.roi_weights,
                                       save_path="", cuda_id=cuda_id, pred_sample_file=pred_sample_file,
                                       with_train_loader=not kwargs.get("with_test_loader", False), in_sample_test=True, **kwargs)
                print_metrics(criterion, *res[0][:10])
        model.train(True)
        model.set_training(True)
        if world > 1:
            dist.barrier()          # ranks 1.. wait for rank 0's validation before the next epoch's first collective
    return epoch_avg_losses


def print_metrics(criterion, mae, mape, rse, rrmse, ssim_error, roi_maes, roi_mapes, roi_rses, roi_wrrmses, roi_correlations,
                  metric_types=""):
    """attn_unet_data_parallel.py:1109-1127."""
    t = metric_types
    logging.info(f"{t}Validation results: MAE {mae} MAPE {mape} RSE {rse} RRMSE {rrmse} SSIM {ssim_error}")
    logging.info(f"{t}ROI MAEs:{roi_maes}\n{t}ROI MAPEs:{roi_mapes}\n{t}ROI RSEs:{roi_rses}\n{t}ROI Weighted RRMSEs: {roi_wrrmses}")
    logging.info(f"{t}ROI correlations: {roi_correlations}")
    lo, hi = int(np.argmin(roi_correlations)), int(np.argmax(roi_correlations))
    ids = criterion.gen_loss.roi_indices
    logging.info(f"\tLowest correlation (ROI {ids[lo]}): {roi_correlations[lo]}; highest (ROI {ids[hi]}): {roi_correlations[hi]}; "
                 f"{t}average: {np.mean(np.nan_to_num(roi_correlations, 0)).item()}")


def record_results(criterion, save_path, hist, epoch, mae, mape, rse, rrmse, ssim_error, roi_maes, roi_mapes, roi_rses,
                   roi_wrrmses, roi_correlations, metric_types=""):
    """attn_unet_data_parallel.py:1036-1107 without the plots: running lists + one `epoch_<n>` column appended to each
    CSV under <save_path>/validation_metric_results/."""
    print_metrics(criterion, mae, mape, rse, rrmse, ssim_error, roi_maes, roi_mapes, roi_rses, roi_wrrmses, roi_correlations,
                  metric_types)
    npf = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    corr0 = np.nan_to_num(roi_correlations, 0)
    for k, v in (("mae", mae), ("mape", mape), ("rse", rse), ("rrmse", rrmse), ("ssim", ssim_error),
                 ("roi_maes", npf(roi_maes)), ("roi_mapes", npf(roi_mapes)), ("roi_rses", npf(roi_rses)),
                 ("roi_wrrmses", npf(roi_wrrmses)), ("roi_corrs", corr0), ("avg_corr", float(np.mean(corr0)))):
        hist[k].append(v)
    if not save_path:
        return
    import pandas as pd
    d = os.path.join(save_path, "validation_metric_results")
    os.makedirs(d, exist_ok=True)
    cols = {"roi_corr": roi_correlations, "roi_mapes": npf(roi_mapes), "roi_maes": npf(roi_maes), "avg_corr": float(np.mean(corr0)),
            "roi_rse": npf(roi_rses), "roi_rrmses": npf(roi_wrrmses), "mape": float(mape), "mae": float(mae)}
    for name, val in cols.items():
        f = os.path.join(d, f"{name}.csv")
        df = pd.read_csv(f) if os.path.exists(f) and os.path.getsize(f) > 1 else pd.DataFrame()
        df[f"epoch_{epoch}"] = np.atleast_1d(np.asarray(val, dtype=np.float64))
        df.to_csv(f, index=False)


class _Acc:
    """One of the three accumulator sets of contrastive_test (all / abeta == 1 / abeta == 0)."""

    def __init__(self, n_roi, device, roi_indices):
        z = lambda: torch.zeros(n_roi, device=device)
        self.n = 0
        self.mae = self.mape = self.rse = self.rrmse = 0.0
        self.mape_count = 0
        self.roi_maes, self.roi_mapes, self.roi_rses, self.roi_wrrmses, self.roi_nonnan = z(), z(), z(), z(), z()
        self.ssim = M.SSIMMetric(spatial_dims=3, data_range=1.0)
        self.corr = M.RoiCorrMetric(roi_indices)

    def add_batch_terms(self, g, temps, count_mape):
        self.mae = self.mae + g["mae"]
        self.mape = self.mape + g["mape_sum"]
        if count_mape:
            self.mape_count += g["mape_count"]
        self.rse = self.rse + g["rse"]
        self.rrmse = self.rrmse + g["rrmse"]
        t_maes, t_mapes, t_rses, t_wrr, t_nonnan = temps
        self.roi_maes += t_maes
        self.roi_mapes += t_mapes
        self.roi_rses += t_rses
        self.roi_wrrmses += t_wrr
        self.roi_nonnan += t_nonnan

    def finish(self, n, save_path, in_sample, prefix=""):
        div = lambda a, b: (a / b) if b != 0 else (a * float("inf") if torch.is_tensor(a) else (float("inf") if a else float("nan")))
        ssim = self.ssim.aggregate().item() if self.ssim._vals else float("nan")
        self.ssim.reset()
        corr = self.corr.calc_roi_corr() if self.corr.pred_means[0] else np.full(len(self.corr.roi_indices), np.nan)
        if save_path and not in_sample:
            _save_corr_matrices(self.corr, save_path, prefix)
        return (div(self.mae, n), div(self.mape, self.mape_count), div(self.rse, n), div(self.rrmse, n), ssim,
                self.roi_maes / n if n else self.roi_maes * float("nan"), (self.roi_mapes * 100) / self.roi_nonnan,
                self.roi_rses / n if n else self.roi_rses * float("nan"), self.roi_wrrmses / n if n else self.roi_wrrmses * float("nan"),
                corr)


def _save_corr_matrices(corr, save_path, prefix):
    import pandas as pd
    if not corr.pred_means[0]:
        return
    ids = corr.sample_ids if len(corr.sample_ids) == len(corr.pred_means[0]) else None
    pd.DataFrame(np.stack(corr.pred_means)).to_csv(os.path.join(save_path, f"{prefix}pred_means.csv"), header=ids or True, index=False)
    pd.DataFrame(np.stack(corr.gt_means)).to_csv(os.path.join(save_path, f"{prefix}gt_means.csv"), header=ids or True, index=False)


def contrastive_test(model, test_loader, roi_indices, roi_weights, save_path="", cuda_id=0, pred_sample_file="",
                     with_train_loader=False, **kwargs):
    """attn_unet_data_parallel.py:1129.  Returns (general, pos, neg) [+ embeddings when model.embeddings_out], each a
    tuple (mae, mape, rse, rrmse, ssim, roi_maes, roi_mapes, roi_rses, roi_wrrmses, roi_correlations); `general` carries an
    11th entry, the (never updated upstream, :1184-1185,1358) voxel-MAPE volume.

    Accumulation follows the reference term by term, including where it is surprising ("as upstream"):
     * the overall MAPE is divided by a counter nothing increments (:1302 `mape / mape_smp_count`, count stays 0) -> inf;
     * the per-class (abeta 1 / 0) sums add the WHOLE batch's term once per sample of that class (:1270-1296);
     * the per-class ROI correlations branch on sample 0's abeta for every sample of the batch (:1246 `abeta[0]`).
    """
    cwd = os.getcwd()
    lookup = _roi_lookup(kwargs, [f"{cwd}/scripts/CatBoostUQ_longitudinal_ADNI_predictions/CatBoostUQ_longitudinal_predictions/"
                                  "CatBoostUQ_predictions_for_unseen_longitudinal_ADNI.json"])           # :1132-1137
    get_id = kwargs.get("extract_id", extract_id)
    device = torch.device("cuda", cuda_id) if isinstance(cuda_id, int) else torch.device(cuda_id)
    in_sample = bool(kwargs.get("in_sample_test", False))
    model.eval()
    model.set_training(False)                                                                             # :1144
    n_roi = len(roi_indices)
    acc = _Acc(n_roi, device, roi_indices)
    pos = _Acc(n_roi, device, roi_indices)
    neg = _Acc(n_roi, device, roi_indices)
    emb = None
    if model.embeddings_out:
        depth = model.get_depth()
        emb = ([[] for _ in range(depth)], [[] for _ in range(depth)], [])
    for values in test_loader:
        if with_train_loader:
            values = values[0]
        mri, tau, roi, abeta, covars, tau_path = _unpack(values)
        mri, tau, roi, covars = (_dev(t, device) for t in (mri, tau, roi, covars))
        priors = [lookup[get_id(p)] for p in tau_path]
        with torch.no_grad():
            pred = model(mri, covars, roi_pred_dicts=priors, sample_roi_mask=roi)                         # :1209
        if model.embeddings_out:
            pred, projected, _final, inter = pred
            for i in range(len(projected)):
                emb[1][i].append(projected[i].detach().float().cpu())
                emb[0][i].append(inter[i].detach().float().cpu())
            emb[2].extend(tau_path)
        pred = pred.float()
        B = mri.size(0)
        st = M.eval_stats(pred, tau, roi, roi_indices)              # ONE pass over (pred, tau, roi) for everything below
        g = M.batch_global_metrics(pred, tau, stats=st)                                                   # :1214-1231
        temps = M.calc_roi_metrics(roi_indices, roi_weights, None, None, None, None, None, tau, roi, pred, stats=st)   # :1254
        acc.add_batch_terms(g, temps, count_mape=False)             # (as upstream: mape_smp_count is never incremented)
        acc.n += B
        acc.ssim(y_pred=pred, y=tau)                                                                      # :1234
        acc.corr.acc_roi_corr(pred, tau, roi, stats=st)                                                   # :1242
        acc.corr.acc_sample_ids(tau_path)
        ab = torch.as_tensor(abeta).reshape(-1).tolist()
        for b in range(B):
            cls = pos if ab[b] == 1 else (neg if ab[b] == 0 else None)
            if cls is not None:
                cls.ssim(y_pred=pred[b][None], y=tau[b][None])                                            # :1235-1239
                cls.add_batch_terms(g, temps, count_mape=True)                                            # :1270-1296 (as upstream)
                cls.n += 1
            c2 = pos if ab[0] == 1 else neg                                                               # :1246 (as upstream)
            c2.corr.acc_roi_corr(pred[b][None], tau[b][None], roi[b][None])
            c2.corr.acc_sample_ids([tau_path[b]])
    general = acc.finish(acc.n, save_path, in_sample)
    pos_res = pos.finish(pos.n, save_path, in_sample, "pos_")
    neg_res = neg.finish(neg.n, save_path, in_sample, "neg_")
    if model.embeddings_out:
        return general, pos_res, neg_res, emb
    vol = torch.zeros(tuple(getattr(model, "volume_shape", (128, 128, 128))), device=device)
    return general + (100 * vol / max(acc.n, 1),), pos_res, neg_res
