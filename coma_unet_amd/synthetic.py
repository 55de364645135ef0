"""Seeded synthetic MRI / tau-PET / ROI-label / covariate / ROI-prior inputs.

Shapes and value ranges follow the sample tuple of
/root/reference/VolumeDataset_ADNI_A4_combined.py:58-91 (``mri, tau, roi,
(abeta, covars), tau_path``) and SURVEY.md section 8(d).  Generation happens on
the CPU with a ``torch.Generator`` so that the CPU oracle and the GPU path see
bit-identical inputs; there are no files and no network involved.
"""
from __future__ import annotations

import torch

from .roi_tables import ROI_INDICES, ROI_INDEX_TO_NAME


def make_batch(batch: int, shape=(128, 128, 128), seed: int = 0, device="cpu"):
    """Returns dict(mri, tau, roi (B,1,D,H,W) fp32; covars (B,1,6) fp64;
    roi_pred_dicts list[B] of {name: {'loc','std'}})."""
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    cz, cy, cx = (D - 1) / 2, (H - 1) / 2, (W - 1) / 2
    inside = (((zz - cz) / (0.4 * D)) ** 2 + ((yy - cy) / (0.4 * H)) ** 2 + ((xx - cx) / (0.4 * W)) ** 2) <= 1.0
    # labels in contiguous blocks (8^3 at 128^3, scaled down for small volumes)
    bs = max(2, min(D, H, W) // 16)
    labels = torch.tensor(ROI_INDICES + [2], dtype=torch.float32)  # 36 ROIs + "other tissue"
    rois, mris, taus = [], [], []
    for _ in range(batch):
        nb = ((D + bs - 1) // bs, (H + bs - 1) // bs, (W + bs - 1) // bs)
        pick = torch.randint(0, len(labels), nb, generator=g)
        lab = labels[pick].repeat_interleave(bs, 0).repeat_interleave(bs, 1).repeat_interleave(bs, 2)[:D, :H, :W]
        roi = torch.where(inside, lab, torch.zeros(()))
        mri = torch.rand((D, H, W), generator=g) * (roi != 0)
        tau = torch.clamp(1.1 + 0.25 * torch.randn((D, H, W), generator=g), min=0.0) * (roi != 0)
        rois.append(roi), mris.append(mri), taus.append(tau)
    roi = torch.stack(rois).unsqueeze(1).contiguous()
    mri = torch.stack(mris).unsqueeze(1).contiguous()
    tau = torch.stack(taus).unsqueeze(1).contiguous()
    u = torch.rand((batch, 6), generator=g, dtype=torch.float64)
    covars = torch.stack([
        (u[:, 0] < 0.5).double(),            # abeta ~ Bernoulli(.5)
        u[:, 1],                             # age (min-max scaled upstream)
        (u[:, 2] < 0.5).double(),            # sex
        u[:, 3],                             # education / 30
        u[:, 4],                             # cognition
        0.9 + 1.1 * u[:, 5],                 # meta-ROI tau SUVR
    ], dim=1).unsqueeze(1)
    pri = torch.rand((batch, len(ROI_INDICES), 2), generator=g, dtype=torch.float64)
    dicts = []
    for b in range(batch):
        dicts.append({ROI_INDEX_TO_NAME[idx]: {"loc": float(0.9 + 1.6 * pri[b, i, 0]),
                                               "std": float(0.01 + 0.29 * pri[b, i, 1])}
                      for i, idx in enumerate(ROI_INDICES)})
    out = dict(mri=mri, tau=tau, roi=roi, covars=covars, roi_pred_dicts=dicts)
    if device != "cpu":
        for k in ("mri", "tau", "roi", "covars"):
            out[k] = out[k].to(device)
    return out
