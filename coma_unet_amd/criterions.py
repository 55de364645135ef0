"""MI355X drop-in for the live loss path of the reference ``criterions`` module.

Same class names / call signatures as /root/reference/criterions.py:
``RoiMSE`` (:124-211), ``GenerativeContrastiveLoss`` (:485-575),
``LabelDifference`` / ``FeatureSimilarity`` / ``RnCLoss`` (:579-644).  The voxel losses are
fused HIP kernels (one pass: label->weight LUT, squared/absolute difference, per-sample
reduction; gradient in one more pass).  RnC works on a (B, 512) feature matrix and a (B, 6)
label matrix -- a few thousand numbers -- and stays as torch glue.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .roi_tables import ROI_INDICES


def _vol_internal(t: torch.Tensor, dtype=None):
    """(B, 1, D, H, W) -> (B, D, H, W, 1)"""
    assert t.dim() == 5 and t.shape[1] == 1, t.shape
    v = t.permute(0, 2, 3, 4, 1)
    return v if dtype is None else v.to(dtype)


class RoiMSE(nn.Module):
    """loss_b = mean_vox(mask_b) * mean_vox((pred_b - gt_b)^2), mask = ROI weight of each voxel's label."""

    def __init__(self, roi_weights, roi_indices, reduction="mean", scale_factor=360, voxel_wise=False, template=None):
        super().__init__()
        self.roi_weights = roi_weights
        self.roi_indices = roi_indices
        self.batch_reduction = reduction
        self.scale_factor = scale_factor
        self.voxel_wise = voxel_wise
        self.voxel_weights = None
        if voxel_wise:
            # criterions.py:135-145.  The reference reads its label template from a private file
            # (data_util.load_template(), absent upstream); here it is the `template` argument (a (D, H, W) label volume).
            if template is None:
                try:
                    import data_util
                    template = data_util.load_template()
                except Exception as e:
                    raise NotImplementedError("RoiMSE(voxel_wise=True) needs the label template: pass template=<(D,H,W) label "
                                              "volume> (the reference loads it with data_util.load_template(), criterions.py:138)") from e
            roi_mask = torch.as_tensor(template)
            w = torch.as_tensor(roi_weights, dtype=torch.float32)
            voxel_weights = torch.ones(tuple(roi_mask.shape), dtype=torch.float32, device=w.device)
            roi_mask = roi_mask.to(w.device)
            for i, idx in enumerate(self.roi_indices):
                voxel_weights[roi_mask == idx] = w[i]
            norm_voxel_weights = voxel_weights / torch.norm(voxel_weights)
            nscaling_factor = 5. / torch.mean(norm_voxel_weights)
            self.voxel_weights = nscaling_factor * norm_voxel_weights

    def calculate_new_voxel_weights(self, errors, voxel_weights, with_update=False):   # criterions.py:161-168
        new_weights = voxel_weights * (1 + errors.to(device=self.voxel_weights.device))
        new_weights = new_weights / torch.norm(new_weights)
        scaling_factor = torch.mean(voxel_weights) / torch.mean(new_weights)
        new_weights *= scaling_factor
        if with_update:
            self.update_weights(new_weights)
        return new_weights

    def __str__(self):
        return (f"RoiMSE(\n  (roi_indices, roi_weights)={list(zip(self.roi_indices, self.roi_weights))}\n"
                f"  batch_reduciton={self.batch_reduction}\n)")

    def calculate_new_weights(self, errors, with_update=False):   # criterions.py:154-159
        new_weights = self.roi_weights * (1 / 2) * errors.to(device=self.roi_weights.device)
        new_weights = self.scale_factor * (new_weights / torch.norm(new_weights))
        if with_update:
            self.update_weights(new_weights)
        return new_weights

    def update_weights(self, weights):   # criterions.py:170-172 (a no-op upstream)
        return

    def _tables(self, dev):
        """ROI id / weight tables on the device, rebuilt only when the weights object changes
        (no host->device copy inside the step: keeps it hipGraph-capturable)."""
        key = (dev, id(self.roi_weights), getattr(self.roi_weights, "_version", 0), tuple(self.roi_indices))
        if getattr(self, "_tab_key", None) != key:
            self._tab = (torch.as_tensor(list(self.roi_indices), dtype=torch.int32, device=dev),
                         torch.as_tensor(self.roi_weights, dtype=torch.float32).to(dev).contiguous())
            self._tab_key = key
        return self._tab

    def forward(self, pred, gt, roi):
        dev = pred.device
        p = _vol_internal(pred)
        if self.voxel_weights is not None:
            # criterions.py:189-198: the mask is the (constant) voxel-weight volume, and the per-sample loss is
            # mean(mask * mean_vox((pred - gt)^2)) = mean(voxel_weights) * MSE_b: the same fused kernel with one label
            # (every voxel "0") whose weight is that mean (5 by construction, :143-145)
            key = (dev, tuple(p.shape[:4]))
            if getattr(self, "_vw_key", None) != key:
                self._vw = (torch.zeros(1, dtype=torch.int32, device=dev),
                            self.voxel_weights.float().mean().reshape(1).to(dev).contiguous(),
                            torch.zeros(tuple(p.shape[:4]) + (1,), dtype=torch.float32, device=dev))
                self._vw_key = key
            ids0, w0, zeros = self._vw
            loss = ops.RoiMSELoss.apply(p, _vol_internal(gt, p.dtype).contiguous(), zeros, ids0, w0)
            return torch.mean(loss) if self.batch_reduction == "mean" else loss
        ids, w = self._tables(dev)
        loss = ops.RoiMSELoss.apply(p, _vol_internal(gt, p.dtype).contiguous(),
                                    _vol_internal(roi, torch.float32).contiguous(), ids, w)   # (B, 1)
        if self.batch_reduction == "mean":
            return torch.mean(loss)
        return loss


class VoxelL1(nn.Module):
    """Per-sample voxel MAE (BASELINE config C2 names an L1 loss; the reference only has MAE as a metric,
    attn_unet_data_parallel.py:1215 -- provided as an optional generative loss)."""

    def __init__(self, reduction="mean"):
        super().__init__()
        self.batch_reduction = reduction

    def forward(self, pred, gt, roi=None):
        p = _vol_internal(pred)
        loss = ops.L1Loss.apply(p, _vol_internal(gt, p.dtype).contiguous())
        return torch.mean(loss) if self.batch_reduction == "mean" else loss


class LabelDifference(nn.Module):
    def __init__(self, distance_type="l1"):
        super().__init__()
        self.distance_type = distance_type

    def forward(self, labels):
        if self.distance_type != "l1":
            raise ValueError(self.distance_type)
        return torch.abs(labels[:, None, :] - labels[None, :, :]).sum(dim=-1)


class FeatureSimilarity(nn.Module):
    def __init__(self, similarity_type="l2"):
        super().__init__()
        self.similarity_type = similarity_type

    def forward(self, features):
        if self.similarity_type != "l2":
            raise ValueError(self.similarity_type)
        return -(features[:, None, :] - features[None, :, :]).norm(2, dim=-1)


class RnCLoss(nn.Module):
    """Rank-N-Contrast, criterions.py:607-644."""

    def __init__(self, temperature=2, label_diff="l1", feature_sim="l2"):
        super().__init__()
        self.t = temperature
        self.label_diff_fn = LabelDifference(label_diff)
        self.feature_sim_fn = FeatureSimilarity(feature_sim)

    def forward(self, features, labels):
        features = features.float()
        labels = labels.float()
        if len(features.shape) == 2 * len(labels.shape):
            features = torch.cat([features[:, 0], features[:, 1]], dim=0)
            labels = labels.repeat(2, 1)
        if features.shape[0] == 2:
            # n = 2 (the reference's batch size, run.sh:13): each row has ONE off-diagonal logit, which is its own only
            # "negative", so every log-probability is log(e^l / e^l) = 0 and the loss and its gradient are identically zero
            # (tests/golden/criterions_ref.npz rnc0 holds the reference's 0.0 and zero gradient for n = 2; n = 1 takes
            # the general path below, whose empty loop returns the python float 0.0 with NO gradient, as upstream).
            # Same value, same zero gradient into `features` (so the last projection head keeps receiving zeros and AdamW
            # keeps decaying it, as upstream) -- without the ~30 tiny launches of the general form.
            return (features * 0.0).sum()
        label_diffs = self.label_diff_fn(labels)
        logits = self.feature_sim_fn(features).div(self.t)
        logits_max, _ = torch.max(logits, dim=1, keepdim=True)
        logits = logits - logits_max.detach()
        exp_logits = logits.exp()
        n = logits.shape[0]
        # remove the diagonal with a static gather (masked_select would need a host sync)
        key = (n, logits.device)
        if getattr(self, "_offdiag_key", None) != key:
            cols = [[j for j in range(n) if j != i] for i in range(n)]
            self._offdiag = torch.tensor(cols, dtype=torch.long).reshape(n, max(n - 1, 0)).to(logits.device)
            self._offdiag_key = key
        idx = self._offdiag
        logits = torch.gather(logits, 1, idx)
        exp_logits = torch.gather(exp_logits, 1, idx)
        label_diffs = torch.gather(label_diffs, 1, idx)
        loss = 0.0
        for k in range(n - 1):
            pos_logits = logits[:, k]
            pos_label_diffs = label_diffs[:, k]
            neg_mask = (label_diffs >= pos_label_diffs.view(-1, 1)).float()
            pos_log_probs = pos_logits - torch.log((neg_mask * exp_logits).sum(dim=-1))
            loss = loss + -(pos_log_probs / (n * (n - 1))).sum()
        return loss


class _ZeroWeighted(torch.autograd.Function):
    """A loss term whose weight is exactly 0.0 (criterions.py:562 `regulatory_weight * triplet`, validation.py:154): value 0,
    gradient zeros -- what `0.0 * term` gives for any finite term -- without evaluating the term (TripletMarginLoss on
    (B,1,1,1,2048): ~35 tiny ATen launches forward + backward)."""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (x.shape, x.dtype, x.device)
        return torch.zeros((), dtype=torch.float32, device=x.device)

    @staticmethod
    def backward(ctx, g):
        shape, dtype, device = ctx.meta
        return torch.zeros(shape, dtype=dtype, device=device)


class GenerativeContrastiveLoss(nn.Module):
    """L = gen_weight * sum_b gen_b + reg_weight * pred_space + ds_reg_weight * ds  (criterions.py:544-575)."""

    def __init__(self, ds_contra_loss, gen_loss, pred_space_contra_loss, regulatory_weight, ds_regulatory_weight):
        super().__init__()
        self.ds_contra_loss = ds_contra_loss
        self.gen_loss = gen_loss
        self.pred_space_contra_loss = pred_space_contra_loss
        self.reg_weight = regulatory_weight
        self.ds_reg_weight = ds_regulatory_weight
        self.gen_weight = 1.0
        self.skip_zero_weighted = True      # False: always evaluate the 0.0-weighted term, as the reference does

    def get_pred_space_contra_loss(self, representations):
        return self.pred_space_contra_loss(*representations)

    def get_ds_contra_loss(self, intermediate_extractions):
        return self.ds_contra_loss(*intermediate_extractions)

    def forward(self, prediction, target, roi, final_representations, intermediate_extractions):
        gen_loss = self.gen_loss(prediction, target, roi)
        reduced_gen_loss = torch.sum(gen_loss) if self.gen_loss.batch_reduction is None else gen_loss
        if self.reg_weight == 0.0 and self.skip_zero_weighted:
            total_pred_space = _ZeroWeighted.apply(final_representations[0])      # == 0.0 * triplet(...) for any finite triplet
        else:
            pred_space = self.get_pred_space_contra_loss(tuple(r.float() for r in final_representations))
            total_pred_space = self.reg_weight * pred_space
        ds = self.get_ds_contra_loss(intermediate_extractions)
        total_ds = self.ds_reg_weight * ds
        total = self.gen_weight * reduced_gen_loss + total_pred_space + total_ds
        return total, gen_loss, total_pred_space, total_ds


def build_reference_criterion(device="cuda"):
    """validation.py:130-154 with ``-rnc`` + attn_unet_data_parallel.py:717."""
    w = torch.full((len(ROI_INDICES),), 225.0, device=device)
    crit = GenerativeContrastiveLoss(RnCLoss(), RoiMSE(w, ROI_INDICES, voxel_wise=False),
                                     nn.TripletMarginLoss(1), regulatory_weight=0.0, ds_regulatory_weight=1.0)
    crit.gen_loss.batch_reduction = None
    return crit
