"""CPU ORACLE (test infrastructure, NOT product code) -- CoMA-UNet losses.

CPU restatement of the live loss path of /root/reference/criterions.py:
  RoiMSE.forward ................. criterions.py:181-211
  GenerativeContrastiveLoss ...... criterions.py:485-575
  LabelDifference / FeatureSimilarity / RnCLoss ... criterions.py:579-644
and of the step-level wiring in attn_unet_data_parallel.py:717,842-845,853-856,878.

Pinning: RnCLoss / LabelDifference / FeatureSimilarity / GenerativeContrastiveLoss
are checked against the reference's own ``criterions.py`` imported in the build
container (``oracle/make_golden.py`` -> ``tests/golden/criterions_ref.npz``).
``RoiMSE.forward`` cannot execute on CPU tensors upstream
(``torch.zeros(..., device=roi.get_device())`` with get_device()==-1)
so its restatement is pinned by source text only: **parity unpinned** for RoiMSE.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class RoiMSE(nn.Module):
    """criterions.py:124-211.  ``voxel_wise=False`` is the only live path (validation.py:146):
    loss_b = mean_vox(mask_b) * mean_vox((pred_b-gt_b)^2); ``voxel_wise=True`` replaces the mask by the constant
    voxel-weight volume built from a label template (unpinned: the template file is private upstream)."""

    def __init__(self, roi_weights, roi_indices, reduction="mean", scale_factor=360, voxel_wise=False, template=None):
        super().__init__()
        assert not voxel_wise or template is not None, "voxel_wise=True needs the label template (data_util.load_template() upstream)"
        self.roi_weights = roi_weights
        self.roi_indices = roi_indices
        self.batch_reduction = reduction
        self.scale_factor = scale_factor
        self.voxel_wise = voxel_wise
        self.voxel_weights = None
        if voxel_wise:   # criterions.py:135-145, with the template handed in instead of data_util.load_template()
            voxel_weights = torch.ones(tuple(template.shape))
            roi_mask = torch.as_tensor(template)
            for i, idx in enumerate(self.roi_indices):
                voxel_weights[roi_mask == idx] = float(self.roi_weights[i])
            norm_voxel_weights = voxel_weights / torch.norm(voxel_weights)
            nscaling_factor = 5. / torch.mean(norm_voxel_weights)
            self.voxel_weights = nscaling_factor * norm_voxel_weights

    def calculate_new_weights(self, errors, with_update=False):  # criterions.py:154-159
        new_weights = self.roi_weights * (1 / 2) * errors.to(device=self.roi_weights.device)
        return self.scale_factor * (new_weights / torch.norm(new_weights))

    def forward(self, pred, gt, roi):
        mask = (torch.ones if self.voxel_wise else torch.zeros)(roi.size(), device=roi.device, dtype=pred.dtype)   # :182
        for i, idx in enumerate(self.roi_indices):
            mask[roi == idx] = float(self.roi_weights[i])
        if self.voxel_weights is not None:   # :189-190
            mask = self.voxel_weights.to(device=pred.device, dtype=pred.dtype).unsqueeze(0).expand_as(pred)
        l = torch.mean(torch.square(pred - gt), dim=(-3, -2, -1))
        loss = torch.zeros(l.size(), device=pred.device, dtype=pred.dtype)
        for b in range(pred.size(0)):
            loss[b] = torch.mean(mask[b] * l[b])
        if self.batch_reduction == "mean":
            return torch.mean(loss)
        return loss


class LabelDifference(nn.Module):  # criterions.py:579-590
    def forward(self, labels):
        return torch.abs(labels[:, None, :] - labels[None, :, :]).sum(dim=-1)


class FeatureSimilarity(nn.Module):  # criterions.py:593-604
    def forward(self, features):
        return -(features[:, None, :] - features[None, :, :]).norm(2, dim=-1)


class RnCLoss(nn.Module):  # criterions.py:607-644
    def __init__(self, temperature=2):
        super().__init__()
        self.t = temperature
        self.label_diff_fn = LabelDifference()
        self.feature_sim_fn = FeatureSimilarity()

    def forward(self, features, labels):
        if len(features.shape) == 2 * len(labels.shape):
            features = torch.cat([features[:, 0], features[:, 1]], dim=0)
            labels = labels.repeat(2, 1)
        label_diffs = self.label_diff_fn(labels)
        logits = self.feature_sim_fn(features).div(self.t)
        logits_max, _ = torch.max(logits, dim=1, keepdim=True)
        logits = logits - logits_max.detach()
        exp_logits = logits.exp()
        n = logits.shape[0]
        off = (1 - torch.eye(n, device=logits.device)).bool()
        logits = logits.masked_select(off).view(n, n - 1)
        exp_logits = exp_logits.masked_select(off).view(n, n - 1)
        label_diffs = label_diffs.masked_select(off).view(n, n - 1)
        loss = 0.0
        for k in range(n - 1):
            pos_logits = logits[:, k]
            pos_label_diffs = label_diffs[:, k]
            neg_mask = (label_diffs >= pos_label_diffs.view(-1, 1)).float()
            pos_log_probs = pos_logits - torch.log((neg_mask * exp_logits).sum(dim=-1))
            loss = loss + -(pos_log_probs / (n * (n - 1))).sum()
        return loss


class GenerativeContrastiveLoss(nn.Module):  # criterions.py:485-575
    def __init__(self, ds_contra_loss, gen_loss, pred_space_contra_loss, regulatory_weight, ds_regulatory_weight):
        super().__init__()
        self.ds_contra_loss = ds_contra_loss
        self.gen_loss = gen_loss
        self.pred_space_contra_loss = pred_space_contra_loss
        self.reg_weight = regulatory_weight
        self.ds_reg_weight = ds_regulatory_weight
        self.gen_weight = 1.0

    def forward(self, prediction, target, roi, final_representations, intermediate_extractions):
        gen_loss = self.gen_loss(prediction, target, roi)
        reduced = torch.sum(gen_loss) if self.gen_loss.batch_reduction is None else gen_loss
        ps = self.pred_space_contra_loss(*final_representations)
        total_ps = self.reg_weight * ps
        ds = self.ds_contra_loss(*intermediate_extractions)
        total_ds = self.ds_reg_weight * ds
        total = self.gen_weight * reduced + total_ps + total_ds
        return total, gen_loss, total_ps, total_ds


def build_reference_criterion(device="cpu"):
    """Criterion assembly of validation.py:130-154 with ``-rnc``: RoiMSE(225 x 36,
    voxel_wise=False) + 0.0 * TripletMarginLoss(margin=1) + 1.0 * RnCLoss(t=2);
    train_dp then sets gen_loss.batch_reduction=None (attn_unet_data_parallel.py:717)."""
    from .coma_oracle import ROI_INDICES
    w = torch.full((len(ROI_INDICES),), 225.0, device=device)
    crit = GenerativeContrastiveLoss(RnCLoss(temperature=2), RoiMSE(w, ROI_INDICES, voxel_wise=False),
                                     nn.TripletMarginLoss(margin=1.0), 0.0, 1.0)
    crit.gen_loss.batch_reduction = None
    return crit


def train_step_loss(model_outputs, tau, roi, covars, criterion):
    """Loss wiring of the ``rnc_loss`` branch of train_dp,
    attn_unet_data_parallel.py:842-845,853-856,874,878."""
    pred, projected, final_repr = model_outputs
    feats = torch.vstack([projected[-1]])
    labels = torch.vstack([covars[:, -1].to(feats.dtype)])
    pos = torch.zeros_like(final_repr)   # reference: fp16 zeros; value-identical
    neg = torch.zeros_like(final_repr)
    return criterion(pred, tau, roi, (final_repr, pos, neg), (feats, labels))
