"""The training step of the reference's ``train_dp`` (attn_unet_data_parallel.py:779-912):
zero_grad -> forward -> RnC pseudo-batch (:842-845) -> criterion (:878) -> backward (:884)
-> [gradient all-reduce] -> AdamW step (:885).  Host-side string work (extract_id, holdout
filter), logging and the per-sample ``.item()`` bookkeeping of :892-910 are the caller's.
"""
from __future__ import annotations

import torch

from .optim import FusedAdamW
from .data_parallel import GradReducer


def make_optimizer(model, lr=1e-3):
    """torch.optim.AdamW(model.parameters(), lr) of :736, fused."""
    dyn = [model.pos_dynamic_prompt, model.neg_dynamic_prompt] if not getattr(model, "static_prompts", False) else []
    return FusedAdamW(model.parameters(), lr=lr, dynamic=dyn)


def forward_loss(model, criterion, batch):
    """Returns (total, gen_loss_vec, weighted_pred_contra, weighted_ds_contra), model_outputs."""
    mri, tau, roi, covars = batch["mri"], batch["tau"], batch["roi"], batch["covars"]
    outs = model(mri, covars.to(device=mri.device), roi_pred_dicts=batch["roi_pred_dicts"], sample_roi_mask=roi)
    pred, projected, final_repr = outs[0], outs[1], outs[2]
    feats = torch.vstack([projected[-1]])                                   # :842
    labels = torch.vstack([covars[:, -1].to(device=mri.device)])            # :843
    pos = torch.zeros_like(final_repr)                                      # :855 (fp16 zeros upstream)
    neg = torch.zeros_like(final_repr)                                      # :856
    losses = criterion(pred, tau, roi, (final_repr, pos, neg), (feats, labels))
    return losses, outs


def train_step(model, criterion, optimizer, batch, reducer: GradReducer = None):
    optimizer.zero_grad()                                                   # :806
    if reducer is not None:
        reducer.reset()
    losses, outs = forward_loss(model, criterion, batch)
    losses[0].backward()                                                    # :884
    if reducer is not None:
        reducer.finish()
    optimizer.step()                                                        # :885
    return losses, outs
