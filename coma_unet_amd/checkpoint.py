"""Checkpoint files in the reference's format (attn_unet_data_parallel.py:943-955, resumed by validation.py:221-281):

    {'epoch', 'model_state_dict', 'optimizer_state_dict', 'loss', 'scheduler_state_dict'}

``model_state_dict`` carries the reference's key names (this build's module tree mirrors the MONAI / CondConv one),
``optimizer_state_dict`` is ``torch.optim.AdamW``-shaped (``FusedAdamW.state_dict``), so a file written here loads into
the reference stack and vice versa.  Files are read with ``weights_only=True`` (nothing in them needs unpickling).
"""
from __future__ import annotations

import os

import torch


def checkpoint_dict(epoch, model, optimizer, loss, scheduler=None):
    ckpt = {"epoch": int(epoch), "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
            "loss": loss.detach().cpu() if torch.is_tensor(loss) else loss}
    if scheduler is not None:
        ckpt["scheduler_state_dict"] = scheduler.state_dict()
    return ckpt


def save_checkpoint(save_path, epoch, model, optimizer, loss, scheduler=None, checkpoint_iter=None, write=True):
    """<save_path>/checkpoints/checkpoint_latest_epoch.pth every call and checkpoint_epoch_<n>.pth every
    ``checkpoint_iter`` epochs (:953-955).  Returns the list of files written.  ``write=False`` (data-parallel ranks other
    than 0): the state is still assembled -- ``FusedAdamW.state_dict()`` is a collective under a sharded gradient exchange
    -- but nothing is written."""
    ckpt = checkpoint_dict(epoch, model, optimizer, loss, scheduler)
    if not write:
        return []
    d = os.path.join(save_path, "checkpoints")
    os.makedirs(d, exist_ok=True)
    files = [os.path.join(d, "checkpoint_latest_epoch.pth")]
    if checkpoint_iter and epoch % checkpoint_iter == 0:
        files.append(os.path.join(d, f"checkpoint_epoch_{epoch}.pth"))
    for f in files:
        torch.save(ckpt, f)
    return files


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location="cpu"):
    """validation.py:221-281: returns the epoch to resume at (saved epoch + 1)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return int(ckpt["epoch"]) + 1
