"""the fused evaluation-statistics kernel against a literal CPU restatement of the
reference's calc_roi_metrics / global metrics / RoiCorrMetric (oracle/metrics_oracle.py)."""
import numpy as np
import pytest
import torch


def _case(B=3, S=(12, 14, 16), seed=0):
    from coma_unet_amd.synthetic import make_batch
    from coma_unet_amd.roi_tables import ROI_INDICES
    b = make_batch(B, S, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    pred = torch.clamp(b["tau"] + 0.2 * torch.randn(b["tau"].shape, generator=g), min=0.0)
    return b, pred, ROI_INDICES


def test_oracle_roi_metrics_closed_form():
    from oracle import metrics_oracle as mo
    tau = torch.tensor([1.0, 2.0, 4.0, 0.0]).view(1, 1, 1, 2, 2)
    pred = torch.tensor([1.5, 2.0, 3.0, 1.0]).view(1, 1, 1, 2, 2)
    roi = torch.tensor([17.0, 17.0, 18.0, 0.0]).view(1, 1, 1, 2, 2)
    z = torch.zeros(2)
    diff = pred - tau
    maes, mapes, rses, wrr, nn_ = mo.calc_roi_metrics([17, 18], None, z, z, z, z, z, tau, roi, pred, diff, torch.abs(diff / tau))
    assert abs(float(maes[0]) - 0.25) < 1e-6 and abs(float(maes[1]) - 1.0) < 1e-6
    assert abs(float(mapes[0]) - 0.5) < 1e-6 and float(nn_[0]) == 2.0
    assert abs(float(wrr[0]) - (0.25 / 5.0) ** 0.5) < 1e-6
    g = mo.batch_global_metrics(pred, tau)
    assert abs(float(g["mae"]) - 0.625) < 1e-6 and g["mape_count"] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_stats_kernel_vs_oracle(dtype):
    from oracle import metrics_oracle as mo
    from coma_unet_amd import metrics as pm
    b, pred, rois = _case()
    pred = pred.to(dtype).float()
    tau = b["tau"].to(dtype).float()
    diff = pred - tau
    z = torch.zeros(len(rois))
    ref = mo.calc_roi_metrics(rois, None, z, z, z, z, z, tau.double(), b["roi"].double(), pred.double(), diff.double(),
                              torch.abs(diff.double() / tau.double()))
    got = pm.calc_roi_metrics(rois, None, z, z, z, z, z, tau.cuda().to(dtype), b["roi"].cuda(), pred.cuda().to(dtype))
    for name, r, g in zip(("mae", "mape", "rse", "wrrmse", "nonnan"), ref, got):
        r, g = r.double(), g.double().cpu()
        fin = torch.isfinite(r)
        assert torch.equal(fin, torch.isfinite(g)), name
        assert float(((g[fin] - r[fin]).abs() / (r[fin].abs() + 1e-12)).max()) < 2e-5, name
    gr = mo.batch_global_metrics(pred.double(), tau.double())
    gg = pm.batch_global_metrics(pred.cuda().to(dtype), tau.cuda().to(dtype), roi=b["roi"].cuda(), roi_indices=rois)
    for k in ("mae", "mape_sum", "rse", "rrmse"):
        assert abs(float(gg[k]) - float(gr[k])) <= 2e-5 * abs(float(gr[k])), k
    assert gg["mape_count"] == gr["mape_count"]
    c_ref, c_got = mo.RoiCorrMetric(rois), pm.RoiCorrMetric(rois)
    for s in range(3):
        bb, pp, _ = _case(seed=10 + s)
        pp = pp.to(dtype).float()
        c_ref.acc_roi_corr(pp.double(), bb["tau"].to(dtype).double(), bb["roi"].double())
        c_got.acc_roi_corr(pp.cuda().to(dtype), bb["tau"].cuda().to(dtype), bb["roi"].cuda())
    a, r = c_got.calc_roi_corr(), c_ref.calc_roi_corr()
    ok = np.isfinite(r)
    assert np.allclose(a[ok], r[ok], atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kernel_type,win", [((2, 1, 24, 20, 27), "gaussian", 11), ((1, 1, 16, 33, 18), "uniform", 7),
                                                   ((3, 1, 12, 12, 12), "gaussian", 11)])
def test_ssim_matches_oracle(shape, kernel_type, win):
    """3-D SSIM tile kernel vs the restated MONAI formulas (fp64 torch convs); fp32 window arithmetic: 1e-4 absolute."""
    from coma_unet_amd import metrics as M
    from oracle import metrics_oracle as O
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.rand(shape, generator=g)
    p = (y + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
    want = O.ssim3d(p, y, 1.0, kernel_type, win)
    got = M.ssim3d(p.cuda(), y.cuda(), 1.0, kernel_type, win).cpu()
    assert got.shape == want.shape and float((got - want).abs().max()) < 1e-4, (got, want)
    same = M.ssim3d(y.cuda(), y.cuda(), 1.0, kernel_type, win).cpu()
    assert float((same - 1).abs().max()) < 1e-5                      # identical volumes: SSIM = 1
    m = M.SSIMMetric(spatial_dims=3, data_range=torch.tensor([1.0]), kernel_type=kernel_type, win_size=win)
    m(y_pred=p.cuda(), y=y.cuda()); m(y_pred=y.cuda(), y=y.cuda())
    assert abs(float(m.aggregate()) - float(torch.cat([want, torch.ones(shape[0], dtype=torch.float64)]).mean())) < 1e-4
    m.reset()
    assert m._vals == []
    # bf16 prediction volumes (what the bf16 model emits) go through the same kernel
    gb = M.ssim3d(p.cuda().bfloat16(), y.cuda(), 1.0, kernel_type, win).cpu()
    wb = O.ssim3d(p.bfloat16().float(), y.bfloat16().float(), 1.0, kernel_type, win)
    assert float((gb - wb).abs().max()) < 1e-3
