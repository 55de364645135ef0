"""Evaluation metrics of the reference's ``contrastive_test`` / ``calc_roi_metrics`` / ``RoiCorrMetric``
(attn_unet_data_parallel.py:1129-1359, 1361-1397, 36-96) on the MI355X.

The reference walks 36 ROIs in Python and runs ~7 masked full-volume reductions per ROI and batch.  Here
ONE HIP kernel (``coma_eval_stats``) reads (pred, tau, roi) once and emits, per sample and per bin
(36 ROIs + whole volume), the eight fp64 sums every one of those metrics is a closed-form function of; the
remaining arithmetic runs on a (B, 37, 8) table.  Function names and return contracts follow the reference.
SSIM: ``SSIMMetric`` below restates MONAI's (third party, unpinned upstream) on a fused tile kernel.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import lib, ct, ptr, check

CNT, SAD, SSD, SG, SGG, SP, SMAPE, NMAPE = range(8)


def _vol(t, dtype=None):
    v = t.reshape(t.shape[0], 1, *t.shape[-3:]).permute(0, 2, 3, 4, 1)
    return (v if dtype is None else v.to(dtype)).contiguous()


def eval_stats(pred, tau, roi, roi_indices):
    """(B, n_roi + 1, 8) fp64 table; last bin = whole volume (see include/coma_unet.h)."""
    dev = pred.device
    ids = torch.as_tensor(list(roi_indices), dtype=torch.int32, device=dev)
    p = _vol(pred.detach())
    out = torch.empty((p.shape[0], len(roi_indices) + 1, 8), dtype=torch.float64, device=dev)
    check(lib.coma_eval_stats(ct(p), ct(_vol(tau, p.dtype)), ct(_vol(roi, torch.float32)), ptr(ids), len(roi_indices),
                              ptr(out), L.stream()), "coma_eval_stats")
    return out


def calc_roi_metrics(roi_indices, roi_weights, roi_maes, roi_mapes, roi_rses, roi_wrrmses, roi_nonnan_voxels,
                     tau_volume, roi, pred, diff=None, raw_mape=None, stats=None):
    """Same signature and return tuple as attn_unet_data_parallel.py:1361 (diff / raw_mape are derived in-kernel)."""
    st = eval_stats(pred, tau_volume, roi, roi_indices) if stats is None else stats
    r = st[:, :-1, :]                                          # (B, R, 8)
    n = r[..., CNT]
    maes = (r[..., SAD] / n).sum(0)                            # 0/0 -> nan when a sample lacks the ROI, as upstream
    mapes = r[..., SMAPE].sum(0)
    nonnan = r[..., NMAPE].sum(0)
    wrrmses = torch.sqrt(r[..., SSD] / r[..., SGG]).sum(0)
    rses = (r[..., SSD] / (r[..., SGG] - r[..., SG] ** 2 / n)).sum(0)
    f = lambda t: t.to(torch.float32)
    return f(maes), f(mapes), f(rses), f(wrrmses), f(nonnan)


def batch_global_metrics(pred, tau_volume, stats=None, roi=None, roi_indices=None):
    """Per-batch terms of contrastive_test (:1214-1231): mae, mape_sum (x100), mape_count, rse, rrmse."""
    if stats is None:
        from .roi_tables import ROI_INDICES
        roi_indices = roi_indices or ROI_INDICES
        roi = roi if roi is not None else torch.zeros_like(tau_volume, dtype=torch.float32)
        stats = eval_stats(pred, tau_volume, roi, roi_indices)
    g = stats[:, -1, :]
    nvox = g[:, CNT]
    rse = (g[:, SSD] / (g[:, SGG] - g[:, SG] ** 2 / nvox)).mean()
    rr = torch.sqrt(g[:, SSD] / g[:, SGG])
    return dict(mae=(g[:, SAD].sum() / nvox.sum()).float(), mape_sum=(100.0 * g[:, SMAPE].sum()).float(),
                mape_count=int(g[:, NMAPE].sum()), rse=rse.float(), rrmse=torch.nanmean(rr).float())


class RoiCorrMetric:
    """attn_unet_data_parallel.py:36-96: per-ROI means of prediction and target, Pearson r over the samples."""

    def __init__(self, roi_indices, spatial_dims=3, win_size=7, reduction="mean"):
        self.roi_indices = roi_indices
        self.pred_means = [[] for _ in roi_indices]
        self.gt_means = [[] for _ in roi_indices]
        self.sample_ids = []

    def acc_roi_corr(self, pred, gt, roi, stats=None):
        st = eval_stats(pred, gt, roi, self.roi_indices) if stats is None else stats
        r = st[:, :-1, :]
        pm = (r[..., SP] / r[..., CNT]).float().cpu().numpy()       # (B, R)
        gm = (r[..., SG] / r[..., CNT]).float().cpu().numpy()
        for i in range(len(self.roi_indices)):
            self.pred_means[i].extend(pm[:, i])
            self.gt_means[i].extend(gm[:, i])

    def acc_sample_ids(self, sample_ids):
        self.sample_ids.extend(sample_ids)

    def calc_roi_corr(self):
        return np.array([np.corrcoef(self.pred_means[i], self.gt_means[i])[0, 1] for i in range(len(self.roi_indices))])


# ---------------------------------------------------------------------------------------------------------------------
# SSIM (attn_unet_data_parallel.py:23,1176-1178,1234-1239,1300: monai.metrics.regression.SSIMMetric)
# ---------------------------------------------------------------------------------------------------------------------
def ssim_window(kernel_type="gaussian", win_size=11, kernel_sigma=1.5):
    """1-D separable window; MONAI builds the n-D kernel as the outer product of these."""
    if kernel_type == "gaussian":
        dist = np.arange((1 - win_size) / 2, (1 + win_size) / 2, 1.0)
        g = np.exp(-np.square(dist / kernel_sigma) / 2.0)
        return (g / g.sum()).astype(np.float32)
    if kernel_type == "uniform":
        return np.full((win_size,), 1.0 / win_size, dtype=np.float32)
    raise ValueError(kernel_type)


def ssim3d(y_pred, y, data_range=1.0, kernel_type="gaussian", win_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03):
    """Per-sample SSIM (B,) fp64 of (B, 1, D, H, W) volumes: windowed statistics over the un-padded ("valid") region, the
    formulas of MONAI >= 1.2 `compute_ssim_and_cs`.  (MONAI <= 1.1, whose constructor signature the reference's call
    matches as well, used a uniform 7^3 window with an unbiased-covariance factor; the dependency is unpinned upstream.)"""
    assert y_pred.shape == y.shape and y_pred.dim() == 5 and y_pred.shape[1] == 1
    w = ssim_window(kernel_type, win_size, kernel_sigma)
    p = _vol(y_pred.detach())
    t = _vol(y, p.dtype)
    cp, ctt = ct(p), ct(t)
    nbytes = lib.coma_ssim_ws_bytes(cp, win_size)
    assert nbytes > 0, "volume smaller than the SSIM window"
    part = torch.empty(nbytes // 8, dtype=torch.float64, device=p.device)
    nt = ctypes.c_int32(0)
    dr = float(data_range)
    check(lib.coma_ssim_partial(cp, ctt, w.ctypes.data_as(ctypes.c_void_p), win_size, (k1 * dr) ** 2, (k2 * dr) ** 2, ptr(part),
                                nbytes, ctypes.addressof(nt), L.stream()), "coma_ssim_partial")
    B, D, H, W = p.shape[0], p.shape[1], p.shape[2], p.shape[3]
    nvalid = (D - win_size + 1) * (H - win_size + 1) * (W - win_size + 1)
    return part.view(B, nt.value).sum(1) / nvalid


class SSIMMetric:
    """MONAI's cumulative metric surface as the reference uses it: `m(y_pred=..., y=...)` per batch, `aggregate()`,
    `reset()` (:1234,1300-1301)."""

    def __init__(self, spatial_dims=3, data_range=1.0, kernel_type="gaussian", win_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03,
                 reduction="mean"):
        assert spatial_dims == 3 and reduction == "mean"
        self.data_range = float(data_range.reshape(-1)[0]) if torch.is_tensor(data_range) else float(data_range)
        self.kw = dict(kernel_type=kernel_type, win_size=win_size, kernel_sigma=kernel_sigma, k1=k1, k2=k2)
        self._vals = []

    def __call__(self, y_pred, y):
        v = ssim3d(y_pred, y, self.data_range, **self.kw)
        self._vals.append(v)
        return v.unsqueeze(1).float()

    def aggregate(self):
        return torch.cat(self._vals).mean().float()      # mean over every sample seen since reset()

    def reset(self):
        self._vals = []
