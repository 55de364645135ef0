"""The reference's host loops with the reference's signatures (attn_unet_data_parallel.py:696 `train_dp`, :1129
`contrastive_test`), as exported by coma_unet_amd.attn_unet_data_parallel: evaluation accumulation against a literal CPU
restatement (oracle/metrics_oracle.contrastive_test_accumulate, quirks included), and a train -> checkpoint -> resume run
in the reference's call order (validation.py:158, :276-281, :348)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S = (32, 32, 32)


def _loader(n_batches, B, abetas, seed0=0, triplet=False):
    """Batches in the reference's sample layout (VolumeDataset_ADNI_A4_combined.py:91): (mri, tau, roi, (abeta, covars), paths),
    wrapped as (anchor, pos, neg) for the training loader (:784)."""
    from coma_unet_amd.synthetic import make_batch
    out, lookup = [], {}
    for i in range(n_batches):
        b = make_batch(B, S, seed=seed0 + i)
        ab = torch.tensor(abetas[i], dtype=torch.float32)
        cov = b["covars"].clone()
        cov[:, 0, 0] = ab
        paths = [f"/data/xnat/adni/{i:03d}-S-{j:04d}/PET_2020-01-0{j + 1}_FTP/analysis/rnu.nii" for j in range(B)]
        for j, p in enumerate(paths):
            lookup[f"{i:03d}-S-{j:04d}/PET_2020-01-0{j + 1}_FTP"] = b["roi_pred_dicts"][j]
        item = (b["mri"], b["tau"], b["roi"], (ab, cov), paths)
        out.append((item, item, item) if triplet else item)
    return out, lookup


def _model(seed=0, **kw):
    import coma_unet_amd as cu
    torch.manual_seed(seed)
    m = cu.build_model(volume_shape=S, **kw).cuda()
    m.set_save_attn(None)
    return m


def _close(a, b, tol, name):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    fin = np.isfinite(b)
    assert np.array_equal(fin, np.isfinite(a)), (name, a, b)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b)), name
    if fin.any():
        assert float(np.max(np.abs(a[fin] - b[fin]) / (np.abs(b[fin]) + 1e-9))) < tol, (name, a, b)


def test_contrastive_test_matches_reference_accumulation(tmp_path):
    from coma_unet_amd import attn_unet_data_parallel as A
    from coma_unet_amd.roi_tables import ROI_INDICES
    from oracle import metrics_oracle as mo
    loader, lookup = _loader(3, 2, [[1, 0], [0, 0], [1, 1]], seed0=50)
    m = _model(3)
    m.train(True)
    with torch.no_grad():        # move the BatchNorm running statistics off their initial values first
        for it in loader:
            m(it[0].cuda(), it[3][1].cuda(), roi_pred_dicts=[lookup[A.extract_id(p)] for p in it[4]], sample_roi_mask=it[2].cuda())
    res = A.contrastive_test(m, loader, ROI_INDICES, None, save_path=str(tmp_path), cuda_id=0, roi_vecs_dict=lookup)
    assert not m.training and len(res) == 3 and len(res[0]) == 11 and len(res[1]) == 10 and len(res[2]) == 10
    assert tuple(res[0][10].shape) == S and float(res[0][10].abs().max()) == 0.0          # (:1184,1358: never updated upstream)
    batches = []
    with torch.no_grad():
        for it in loader:
            pred = m(it[0].cuda(), it[3][1].cuda(), roi_pred_dicts=[lookup[A.extract_id(p)] for p in it[4]], sample_roi_mask=it[2].cuda())
            batches.append((pred.float().cpu().double(), it[1].double(), it[2].double(), it[3][0], it[4]))
    ref = mo.contrastive_test_accumulate(batches, ROI_INDICES)
    names = ("mae", "mape", "rse", "rrmse", "ssim", "roi_maes", "roi_mapes", "roi_rses", "roi_wrrmses", "roi_corr")
    for cls, got, want in zip(("all", "pos", "neg"), res, ref):
        for k, g, w in zip(names, got, want):
            _close(g, w, 5e-3 if k == "roi_corr" else 2e-4, f"{cls}.{k}")     # (correlations of fp32 ROI means: differences of nearly equal numbers)
    assert float(res[0][1]) == float("inf")          # overall MAPE: divided by a counter nothing increments (as upstream)
    for f in ("pred_means.csv", "gt_means.csv", "pos_pred_means.csv", "neg_gt_means.csv"):
        assert os.path.exists(os.path.join(str(tmp_path), f)), f


def test_train_dp_trains_checkpoints_validates_and_resumes(tmp_path):
    import coma_unet_amd as cu
    from coma_unet_amd import attn_unet_data_parallel as A
    from coma_unet_amd import checkpoint
    from coma_unet_amd.train import make_optimizer
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    train, lk1 = _loader(3, 2, [[1, 0], [0, 1], [1, 1]], seed0=70, triplet=True)
    val, lk2 = _loader(2, 2, [[1, 0], [0, 0]], seed0=90)
    lookup = {**lk1, **lk2}
    crit = cu.build_reference_criterion()
    m = _model(11)
    m.train(True)
    w_before = [float(w) for w in crit.gen_loss.roi_weights[:3]]
    losses = A.train_dp(m, crit, train, val, 2, 1e-3, save_path=str(tmp_path), cuda_id=0, roi_vecs_dict=lookup, fold_id=3, val_iter=1)
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0]
    assert crit.gen_loss.batch_reduction is None                                   # :717
    ck = os.path.join(str(tmp_path), "checkpoints", "checkpoint_latest_epoch.pth")
    raw = torch.load(ck, map_location="cpu", weights_only=True)
    assert set(raw) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss", "scheduler_state_dict"} and raw["epoch"] == 1
    assert float(raw["optimizer_state_dict"]["state"][0]["step"]) == 6.0           # 2 epochs x 3 batches
    d = os.path.join(str(tmp_path), "validation_metric_results")
    import pandas as pd
    assert list(pd.read_csv(os.path.join(d, "roi_maes.csv")).columns) == ["epoch_0", "epoch_1"]
    assert pd.read_csv(os.path.join(d, "roi_corr.csv")).shape == (36, 2)
    assert [float(w) for w in crit.gen_loss.roi_weights[:3]] == w_before         # :985-990 computes new weights; update_weights is a no-op upstream (criterions.py:170-172)
    assert m.training
    # resume: fresh model + optimizer, load, continue (validation.py:221-281, :348)
    m2 = _model(12)
    opt2 = make_optimizer(m2, 1e-3)
    sch2 = ReduceLROnPlateau(opt2, "min", patience=5)
    start = checkpoint.load_checkpoint(ck, m2, opt2, sch2)
    assert start == 2
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    m2.train(True)
    more = A.train_dp(m2, cu.build_reference_criterion(), train, None, 3, 1e-3, save_path="", cuda_id=0, from_checkpoint=True,
                      optimizer=opt2, scheduler=sch2, start_epoch=start, roi_vecs_dict=lookup)
    assert len(more) == 1 and np.isfinite(more[0]) and more[0] < losses[0]
    assert opt2._flat_step == 9


def test_train_dp_graphed_loop_matches_eager_loop():
    """train_dp(graph=...) on a static_prompts model: batches 1-2 eager, capture on batch 3, later batches copied into the
    graph's static inputs and replayed (the path bench.py times).  Same per-epoch losses as the all-eager loop -- lr 1e-5
    so that the comparison is not about Adam turning fp32-atomic merge order into sign noise (test_graphed_step_matches_
    eager_steps documents that spread) -- and the same number of optimizer steps; a short last batch falls back to eager."""
    import coma_unet_amd as cu
    from coma_unet_amd import attn_unet_data_parallel as A
    from coma_unet_amd import train as T
    train, lookup = _loader(5, 2, [[1, 0], [0, 1], [1, 1], [0, 0], [1, 0]], seed0=120, triplet=True)
    short, lk2 = _loader(1, 1, [[1]], seed0=140, triplet=True)          # B = 1: another shape
    lookup.update(lk2)
    runs = {}
    for mode in (False, True):
        m = _model(21, static_prompts=True, compute_dtype=torch.bfloat16)
        m.train(True)
        crit = cu.build_reference_criterion()
        opt = T.make_optimizer(m, 1e-5)
        captured = []
        orig = T.GraphedTrainStep._capture
        T.GraphedTrainStep._capture = lambda self, _o=orig: (captured.append(1), _o(self))[1]
        try:
            losses = A.train_dp(m, crit, train + short, None, 2, 1e-5, save_path="", cuda_id=0, roi_vecs_dict=lookup, fold_id=0,
                                graph=mode, optimizer=opt)
        finally:
            T.GraphedTrainStep._capture = orig
        runs[mode] = (losses, opt._flat_step, len(captured))
    (le, se, ce), (lg, sg, cg) = runs[False], runs[True]
    assert ce == 0 and cg == 1                       # one capture, no re-capture (the learning rate did not change)
    assert se == sg == 12                            # 2 epochs x 6 batches, every one a real optimizer step
    assert len(le) == len(lg) == 2
    for a, b in zip(lg, le):
        assert np.isfinite(a) and abs(a - b) <= 2e-2 * abs(b), (lg, le)
