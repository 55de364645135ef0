"""CPU ORACLE (test infrastructure, NOT product code) -- evaluation metrics.

Literal CPU restatement of /root/reference/attn_unet_data_parallel.py:
  calc_roi_metrics ............ :1361-1397  (only change: masks are created on `roi.device`;
                                             upstream passes roi.get_device() which is -1 on CPU)
  per-batch global metrics .... :1214-1231  (MAE, nan-aware MAPE, RSE, RRMSE)
  RoiCorrMetric.acc_roi_corr .. :49-60
**parity unpinned**: the reference ships no fixtures for these and its module cannot be imported (MONAI);
pinned by source text only.  Imported by tests/ only.
"""
import numpy as np
import torch


def calc_roi_metrics(roi_indices, roi_weights, roi_maes, roi_mapes, roi_rses, roi_wrrmses, roi_nonnan_voxels,
                     tau_volume, roi, pred, diff, raw_mape):
    dev = roi.device
    roi_mask = torch.zeros(roi.size(), device=dev, dtype=pred.dtype)
    roi_bool_mask = torch.zeros(roi.size(), device=dev, dtype=torch.bool)
    temp_roi_maes = torch.zeros(len(roi_indices), device=dev, dtype=pred.dtype)
    temp_roi_mapes = torch.zeros(len(roi_indices), device=dev, dtype=pred.dtype)
    temp_roi_rses = torch.zeros(len(roi_indices), device=dev, dtype=pred.dtype)
    temp_roi_wrrmses = torch.zeros(len(roi_indices), device=dev, dtype=pred.dtype)
    temp_roi_nonnan_voxels = torch.zeros(len(roi_indices), device=dev, dtype=pred.dtype)
    for i, idx in enumerate(roi_indices):
        roi_mask[:] = 0
        roi_bool_mask[:] = False
        roi_mask[roi == idx] = 1
        roi_bool_mask[roi == idx] = True
        roi_mask_size = torch.count_nonzero(roi_mask, dim=(-3, -2, -1))
        nr_roi_mae = torch.sum(torch.abs(diff) * roi_mask, dim=(-3, -2, -1)) / roi_mask_size
        temp_roi_maes[i] += torch.sum(nr_roi_mae)
        roi_raw_mape = raw_mape[roi_bool_mask]
        temp_roi_mapes[i] = torch.nansum(roi_raw_mape)
        temp_roi_nonnan_voxels[i] = (torch.count_nonzero(roi_bool_mask) - torch.count_nonzero(torch.isnan(roi_raw_mape))).item()
        num = torch.sum(roi_mask * torch.square(diff), dim=(-3, -2, -1))
        den = torch.sum(roi_mask * torch.square(tau_volume), dim=(-3, -2, -1))
        temp_roi_wrrmses[i] = torch.sum(torch.sqrt(num / den))
        gt_mean = torch.sum(roi_mask * tau_volume, dim=(-3, -2, -1)) / roi_mask_size
        roi_se_num = torch.sum(roi_mask * torch.square(tau_volume - pred), dim=(-3, -2, -1))
        roi_se_den = torch.sum(roi_mask * torch.square(tau_volume - gt_mean.view(-1, 1, 1, 1, 1)), dim=(-3, -2, -1))
        temp_roi_rses[i] += torch.sum(roi_se_num / roi_se_den)
    return temp_roi_maes, temp_roi_mapes, temp_roi_rses, temp_roi_wrrmses, temp_roi_nonnan_voxels


def batch_global_metrics(pred, tau_volume):
    """The per-batch terms contrastive_test accumulates (:1214-1231)."""
    diff = pred - tau_volume
    mae = torch.mean(torch.abs(diff))
    nr_mape = torch.where(torch.abs(tau_volume) > 1e-08, torch.abs((tau_volume - pred) / tau_volume),
                          torch.full_like(pred, float("nan")))
    mape_sum = torch.nansum(nr_mape * 100, dim=(-3, -2, -1)).sum()
    mape_count = nr_mape.numel() - int(torch.isnan(nr_mape).sum())
    gt_mean = torch.mean(tau_volume, dim=(-3, -2, -1))
    num = torch.sum(torch.square(tau_volume - pred), dim=(-3, -2, -1))
    den = torch.sum(torch.square(tau_volume - gt_mean.view(-1, 1, 1, 1, 1)), dim=(-3, -2, -1))
    rse = torch.mean(num / den)
    rrmse = torch.nanmean(torch.sqrt(num / torch.sum(torch.square(tau_volume), dim=(-3, -2, -1))))
    return dict(mae=mae, mape_sum=mape_sum, mape_count=mape_count, rse=rse, rrmse=rrmse)


class RoiCorrMetric:   # :36-96 (accumulation + correlation only; CSV dumps are out of scope)
    def __init__(self, roi_indices):
        self.roi_indices = roi_indices
        self.pred_means = [[] for _ in roi_indices]
        self.gt_means = [[] for _ in roi_indices]

    def acc_roi_corr(self, pred, gt, roi):
        roi_mask = torch.zeros(roi.size(), device=roi.device, dtype=pred.dtype)
        for i, idx in enumerate(self.roi_indices):
            roi_mask[:] = 0
            roi_mask[roi == idx] = 1
            cnt = torch.count_nonzero(roi_mask, dim=(-3, -2, -1))
            self.pred_means[i].extend((torch.sum(roi_mask * pred, dim=(-3, -2, -1)) / cnt).detach().cpu().numpy().flatten())
            self.gt_means[i].extend((torch.sum(roi_mask * gt, dim=(-3, -2, -1)) / cnt).detach().cpu().numpy().flatten())

    def calc_roi_corr(self):
        return np.array([np.corrcoef(self.pred_means[i], self.gt_means[i])[0, 1] for i in range(len(self.roi_indices))])



def ssim3d(y_pred, y, data_range=1.0, kernel_type="gaussian", win_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03):
    """MONAI >= 1.2 `compute_ssim_and_cs` + per-sample mean (monai/metrics/regression.py), restated from its published
    source with stock torch ops (the dependency itself is absent and unpinned upstream: PARITY UNPINNED).  fp64."""
    import torch
    import torch.nn.functional as F
    x, t = y_pred.double(), y.double()
    if kernel_type == "gaussian":
        dist = torch.arange((1 - win_size) / 2, (1 + win_size) / 2, 1.0, dtype=torch.float64)
        g = torch.exp(-torch.pow(dist / kernel_sigma, 2) / 2)
        w1 = g / g.sum()
    else:
        w1 = torch.full((win_size,), 1.0 / win_size, dtype=torch.float64)
    k = (w1[:, None, None] * w1[None, :, None] * w1[None, None, :])[None, None]
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    mu_x, mu_y = F.conv3d(x, k), F.conv3d(t, k)
    mu_xx, mu_yy, mu_xy = F.conv3d(x * x, k), F.conv3d(t * t, k), F.conv3d(x * t, k)
    sx, sy, sxy = mu_xx - mu_x * mu_x, mu_yy - mu_y * mu_y, mu_xy - mu_x * mu_y
    cs = (2 * sxy + c2) / (sx + sy + c2)
    full = ((2 * mu_x * mu_y + c1) / (mu_x ** 2 + mu_y ** 2 + c1)) * cs
    return full.reshape(full.shape[0], -1).mean(1)


def contrastive_test_accumulate(batches, roi_indices, ssim_fn=None):
    """Literal CPU restatement of the accumulation of contrastive_test (attn_unet_data_parallel.py:1147-1347) over
    `batches` = [(pred, tau_volume, roi, abeta, tau_path)], i.e. everything after the model call of :1209, quirks
    included: `mape / mape_smp_count` with a counter that is never incremented (:1302), whole-batch terms added once per
    sample of a class (:1270-1296), `abeta[0]` deciding the class of every sample's ROI means (:1246).
    Returns (general, pos, neg) 10-tuples like the reference (SSIM through `ssim_fn`, default the restatement above)."""
    ssim_fn = ssim_fn or ssim3d
    n_roi = len(roi_indices)

    def fresh():
        return dict(n=0, mae=0, mape=0, rse=0, rrmse=0, cnt=0, maes=torch.zeros(n_roi), mapes=torch.zeros(n_roi),
                    rses=torch.zeros(n_roi), wrr=torch.zeros(n_roi), nonnan=torch.zeros(n_roi), ssim=[],
                    corr=RoiCorrMetric(roi_indices))
    A, P, N = fresh(), fresh(), fresh()
    for pred, tau_volume, roi, abeta, tau_path in batches:
        diff = pred - tau_volume
        g = batch_global_metrics(pred, tau_volume)
        raw_mape = torch.abs(diff / tau_volume)
        A["mae"] += g["mae"]; A["mape"] += g["mape_sum"]; A["rse"] += g["rse"]; A["rrmse"] += g["rrmse"]
        A["ssim"].append(ssim_fn(pred, tau_volume))
        for b in range(pred.size(0)):
            if abeta[b] == 1:
                P["ssim"].append(ssim_fn(pred[b][None], tau_volume[b][None]))
            elif abeta[b] == 0:
                N["ssim"].append(ssim_fn(pred[b][None], tau_volume[b][None]))
        A["corr"].acc_roi_corr(pred, tau_volume, roi)
        for b in range(pred.size(0)):
            (P if abeta[0] == 1 else N)["corr"].acc_roi_corr(pred[b], tau_volume[b], roi[b])
        z = torch.zeros(n_roi)
        t = calc_roi_metrics(roi_indices, None, z, z, z, z, z, tau_volume, roi, pred, diff, raw_mape)
        for k, v in zip(("maes", "mapes", "rses", "wrr", "nonnan"), t):
            A[k] = A[k] + v
        A["n"] += pred.size(0)
        for b in range(pred.size(0)):
            C = P if abeta[b] == 1 else (N if abeta[b] == 0 else None)
            if C is None:
                continue
            C["mae"] += g["mae"]; C["mape"] += g["mape_sum"]; C["cnt"] += g["mape_count"]
            C["rse"] += g["rse"]; C["rrmse"] += g["rrmse"]
            for k, v in zip(("maes", "mapes", "rses", "wrr", "nonnan"), t):
                C[k] = C[k] + v
            C["n"] += 1

    def fin(C):
        ssim = float(torch.cat(C["ssim"]).mean()) if C["ssim"] else float("nan")
        return (C["mae"] / C["n"], C["mape"] / C["cnt"], C["rse"] / C["n"], C["rrmse"] / C["n"], ssim, C["maes"] / C["n"],
                (C["mapes"] * 100) / C["nonnan"], C["rses"] / C["n"], C["wrr"] / C["n"], C["corr"].calc_roi_corr())
    return fin(A), fin(P), fin(N)
