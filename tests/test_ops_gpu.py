"""Per-kernel parity: every C-ABI entry point against a CPU fp64 torch restatement of the
same reference op (the ATen/MONAI op it replaces), on seeded inputs.  Tolerances: fp32 path
1e-5 rel-L2 (summation-order differences only), bf16 path 2e-2 (bf16 storage rounding)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from coma_unet_amd import ops, _lib
    return ops, _lib


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if n > 0 else float(d)


def to_int(x):      # (B,C,D,H,W) -> (B,D,H,W,C) contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ext(x):
    return x.permute(0, 4, 1, 2, 3)


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}

CONV_CASES = [
    # cin, cout, k, stride, transposed, size
    (1, 32, 3, 1, False, 10), (32, 32, 3, 1, False, 8), (32, 64, 3, 2, False, 8), (64, 32, 3, 2, True, 4),
    (64, 32, 3, 1, False, 6), (32, 16, 1, 1, False, 6), (16, 1, 1, 1, False, 6), (3, 16, 3, 1, False, 7),
    (16, 1, 3, 1, False, 6), (2, 8, 3, 1, False, 5), (8, 8, 3, 1, False, 5), (2, 1, 1, 1, False, 5),
    (128, 64, 3, 2, True, 3), (64, 128, 3, 2, False, 6), (5, 7, 3, 2, False, 9), (7, 5, 3, 2, True, 5),
    (32, 1, 1, 1, False, 6), (1, 1, 1, 1, False, 5), (128, 1, 1, 1, False, 4), (24, 1, 1, 1, False, 5),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, per_sample, dtype):
    _conv_case(case, per_sample, dtype, 1)


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("case", CONV_CASES + [(1, 32, 3, 1, False, 33), (3, 16, 3, 1, False, 34), (16, 1, 3, 1, False, 35),
                                               (2, 8, 3, 1, False, 33), (8, 8, 3, 1, False, 40), (8, 1, 3, 1, False, 32),
                                               (16, 16, 3, 1, False, 32), (12, 20, 3, 1, False, 9), (32, 16, 1, 1, False, 12)])
def test_conv_fwd_bwd_fp32_auto(case, per_sample):
    """fp32 tensors with algo 0 (what the model runs in fp32 mode): whichever kernel family the library picks per shape --
    direct, streaming 1-channel, fp32 MFMA, and the tap-packed fp32 MFMA weight gradient of the few-channel layers --
    must meet the fp32 bound against the fp64 reference."""
    _conv_case(case, per_sample, torch.float32, 0)


def _conv_case(case, per_sample, dtype, algo):
    ops, L = _ops()
    cin, cout, k, s, tr, S = case
    B, E = 2, 3
    g = torch.Generator().manual_seed(hash(case) % 1000)
    dims = (S, S + 1, S + 2) if not (s == 2 and not tr) else (S, S + 2, S + 4)
    x = torch.randn((B, cin, *dims), generator=g, dtype=torch.float64)
    wshape = (cin, cout, k, k, k) if tr else (cout, cin, k, k, k)
    if per_sample:
        master = torch.randn((E, *wshape), generator=g, dtype=torch.float64) * 0.2
        r = torch.rand((B, E), generator=g, dtype=torch.float64)
        bias = torch.randn((B, cout), generator=g, dtype=torch.float64)
    else:
        master = torch.randn(wshape, generator=g, dtype=torch.float64) * 0.2
        r = None
        bias = torch.randn((cout,), generator=g, dtype=torch.float64)
    if dtype == torch.bfloat16:   # make inputs exactly representable so only accumulation differs
        x = x.bfloat16().double()
    xr = x.clone().requires_grad_(True)
    mr = master.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if per_sample else None
    br = bias.clone().requires_grad_(True)
    p = (k - 1) // 2

    def ref_conv(xb, w, b):
        if tr:
            return F.conv_transpose3d(xb, w, b, stride=s, padding=p, output_padding=s - 1)
        return F.conv3d(xb, w, b, stride=s, padding=p)

    if per_sample:
        wmix = torch.einsum("be,e...->b...", rr, mr)
        yr = torch.cat([ref_conv(xr[i:i + 1], wmix[i], br[i]) for i in range(B)], 0)
    else:
        yr = ref_conv(xr, mr, br)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().double()
    yr.backward(gy)

    dev = "cuda"
    xi = to_int(x).to(dev, dtype).requires_grad_(True)
    mg = master.float().to(dev).requires_grad_(True)
    rg = r.float().to(dev).requires_grad_(True) if per_sample else None
    bg = bias.float().to(dev).requires_grad_(True)
    wk_f, wk_d = ops.PrepWeights.apply(mg, rg, tr, torch.float32, torch.float32)
    y = ops.Conv.apply(xi, wk_f, wk_d, bg, k, s, tr, per_sample, algo, None)
    assert rel(to_ext(y), yr) < TOL[dtype]
    y.backward(to_int(gy).to(dev, dtype))
    tol = TOL[dtype]
    assert rel(to_ext(xi.grad), xr.grad) < tol
    assert rel(mg.grad, mr.grad) < tol
    assert rel(bg.grad, br.grad) < tol
    if per_sample:
        assert rel(rg.grad, rr.grad) < tol * 5


ACTS = ["none", "relu", "prelu", "leaky", "sigmoid", "prelu_relu"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["batch", "instance"])
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("C", [1, 3, 16, 32, 96])
def test_norm_act(C, act, mode, dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(C * 7 + len(act))
    B, dims = 2, (5, 6, 7)
    x = torch.randn((B, C, *dims), generator=g, dtype=torch.float64) * 1.5 + 0.7
    if dtype == torch.bfloat16:
        x = x.bfloat16().double()
    gamma = torch.rand((C,), generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn((C,), generator=g, dtype=torch.float64) * 0.3
    slope = torch.tensor([-0.3 if act == "prelu_relu" else 0.2], dtype=torch.float64)
    affine = mode == "batch"
    xr, gr, br, sr = (t.clone().requires_grad_(True) for t in (x, gamma, beta, slope))
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    if mode == "batch":
        z = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    else:
        z = F.instance_norm(xr, eps=1e-5)
    yr = {"none": lambda t: t, "relu": F.relu, "prelu": lambda t: F.prelu(t, sr),
          "leaky": lambda t: F.leaky_relu(t, 0.01), "sigmoid": torch.sigmoid,
          "prelu_relu": lambda t: F.relu(F.prelu(t, sr))}[act](z)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().double()
    yr.backward(gy)

    dev = "cuda"
    xi = to_int(x).to(dev, dtype).requires_grad_(True)
    gg = gamma.float().to(dev).requires_grad_(True) if affine else None
    bg = beta.float().to(dev).requires_grad_(True) if affine else None
    sg = slope.float().to(dev).requires_grad_(True) if act in ("prelu", "prelu_relu") else None
    rmg = torch.zeros(C, device=dev) if affine else None
    rvg = torch.ones(C, device=dev) if affine else None
    code = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "prelu": L.ACT_PRELU, "leaky": L.ACT_LEAKY,
            "sigmoid": L.ACT_SIGMOID, "prelu_relu": L.ACT_PRELU_RELU}[act]
    y = ops.NormAct.apply(xi, gg, bg, sg, rmg, rvg, L.NORM_BATCH if affine else L.NORM_INSTANCE, code, 0.1, 1e-5, True, None)
    tol = TOL[dtype]
    assert rel(to_ext(y), yr) < tol
    y.backward(to_int(gy).to(dev, dtype))
    assert rel(to_ext(xi.grad), xr.grad) < tol * 3
    if affine:
        assert rel(gg.grad, gr.grad) < tol * 3
        assert rel(bg.grad, br.grad) < tol * 3
        assert rel(rmg, rm) < 1e-5 and rel(rvg, rv) < 1e-5
    if sg is not None:
        assert rel(sg.grad, sr.grad) < tol * 3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [32, 64, 256])
def test_gate_pieces(C, dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(C)
    B, dims = 2, (4, 5, 6)
    mk = lambda c: (torch.randn((B, c, *dims), generator=g, dtype=torch.float64)).to(dtype).double()
    a, b, x, psi = mk(C), mk(C), mk(C), torch.sigmoid(mk(1))
    psi = psi.to(dtype).double()
    ar, br, xr, pr = (t.clone().requires_grad_(True) for t in (a, b, x, psi))
    out_r = F.relu(ar + br) * 1.0
    att_r = xr * pr
    gy1, gy2 = mk(C), mk(C)
    (out_r * gy1).sum().backward()
    (att_r * gy2).sum().backward()
    dev = "cuda"
    ai, bi, xi, pi = (to_int(t).to(dev, dtype).requires_grad_(True) for t in (a, b, x, psi))
    out = ops.AddRelu.apply(ai, bi)
    # write the gate output into a channel slice of a wider buffer (concat-free path)
    cat = torch.zeros((B, *dims, 2 * C), device=dev, dtype=dtype)
    att = ops.GateMul.apply(xi, pi, ops.Out(cat[..., :C]))
    tol = TOL[dtype]
    assert rel(to_ext(out), out_r) < tol and rel(to_ext(att), att_r) < tol
    assert float(cat[..., C:].abs().max()) == 0.0
    out.backward(to_int(gy1).to(dev, dtype))
    att.backward(to_int(gy2).to(dev, dtype))
    assert rel(to_ext(ai.grad), ar.grad) < tol and rel(to_ext(bi.grad), br.grad) < tol
    assert rel(to_ext(xi.grad), xr.grad) < tol
    assert rel(to_ext(pi.grad), pr.grad) < tol * 3


def test_roimse_matches_reference_golden():
    """The fused RoiMSE kernels against tests/golden/criterions_ref.npz `roimse{B}_*`: loss vectors and pred gradients
    computed by the REFERENCE's own criterions.RoiMSE.forward (criterions.py:181-211; oracle/make_golden.py), fp32,
    225/100/7.5 weight mix, B = 1, 2, 3; and the module-level `RoiMSE` of coma_unet_amd.criterions with both reductions."""
    import os
    import numpy as np
    ops, L = _ops()
    from coma_unet_amd.roi_tables import ROI_INDICES
    from coma_unet_amd import criterions as prod
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "criterions_ref.npz"))
    dev = "cuda"
    ids = torch.tensor(ROI_INDICES, dtype=torch.int32, device=dev)
    w = torch.from_numpy(g["gcl_w"])
    for Bn in (1, 2, 3):
        pg = to_int(torch.from_numpy(g[f"roimse{Bn}_pred"])).to(dev).requires_grad_(True)
        lg = ops.RoiMSELoss.apply(pg, to_int(torch.from_numpy(g[f"roimse{Bn}_gt"])).to(dev),
                                  to_int(torch.from_numpy(g[f"roimse{Bn}_roi"])).to(dev), ids, w.to(dev))
        assert tuple(lg.shape) == (Bn, 1)
        assert rel(lg, torch.from_numpy(g[f"roimse{Bn}_loss"])) < 1e-5
        lg.sum().backward()
        assert rel(to_ext(pg.grad), torch.from_numpy(g[f"roimse{Bn}_grad"])) < 1e-5
        crit = prod.RoiMSE(w.to(dev), ROI_INDICES, reduction=None)
        pe = torch.from_numpy(g[f"roimse{Bn}_pred"]).to(dev).requires_grad_(True)
        lv = crit(pe, torch.from_numpy(g[f"roimse{Bn}_gt"]).to(dev), torch.from_numpy(g[f"roimse{Bn}_roi"]).to(dev))
        assert rel(lv, torch.from_numpy(g[f"roimse{Bn}_loss"])) < 1e-5
        lv.sum().backward()
        assert rel(pe.grad, torch.from_numpy(g[f"roimse{Bn}_grad"])) < 1e-5
    crit = prod.RoiMSE(w.to(dev), ROI_INDICES)      # reduction "mean" (criterions.py:206-208)
    lm = crit(torch.from_numpy(g["roimse3_pred"]).to(dev), torch.from_numpy(g["roimse3_gt"]).to(dev),
              torch.from_numpy(g["roimse3_roi"]).to(dev))
    assert abs(float(lm) - float(g["roimse3_mean"])) < 1e-5 * abs(float(g["roimse3_mean"]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_paint_and_losses(dtype):
    ops, L = _ops()
    from coma_unet_amd.synthetic import make_batch
    from coma_unet_amd.roi_tables import ROI_INDICES, ROI_INDEX_TO_NAME
    from oracle.criterions_oracle import RoiMSE as ORoiMSE
    B, S = 3, (8, 10, 12)
    b = make_batch(B, S, seed=5)
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    pos = torch.randn((1, 1, *S), generator=g)
    neg = torch.randn((1, 1, *S), generator=g)
    prior = torch.tensor([[[d[ROI_INDEX_TO_NAME[i]]["loc"], d[ROI_INDEX_TO_NAME[i]]["std"]] for i in ROI_INDICES]
                          for d in b["roi_pred_dicts"]], dtype=torch.float32)
    abeta = b["covars"][:, 0, 0].float()
    # oracle painting (attn_unet_data_parallel.py:630-651)
    x = b["mri"].to(dtype).float()
    suvr = torch.zeros_like(x)
    sal = torch.zeros_like(x)
    for bb in range(B):
        for i, idx in enumerate(ROI_INDICES):
            m = b["roi"][bb] == idx
            suvr[bb][m] = prior[bb, i, 0]
            sal[bb][m] = prior[bb, i, 1]
    suvr = torch.where(x < 1e-4, torch.zeros_like(suvr), suvr)
    sal = torch.where(x < 1e-4, torch.zeros_like(sal), sal)
    dyn = torch.vstack([pos if abeta[bb] == 1 else neg for bb in range(B)])
    ref3 = torch.cat((dyn, sal, suvr), dim=1)
    posg, negg = pos.to(dev).requires_grad_(True), neg.to(dev).requires_grad_(True)
    ids = torch.tensor(ROI_INDICES, dtype=torch.int32, device=dev)
    out3 = ops.RoiPaint.apply(posg, negg, to_int(b["roi"]).to(dev), to_int(b["mri"]).to(dev, dtype), prior.to(dev), ids,
                              abeta.to(dev), dtype)
    assert rel(to_ext(out3), ref3.to(dtype)) < 1e-6
    gy = torch.randn(ref3.shape, generator=g)
    out3.backward(to_int(gy).to(dev, dtype))
    gyq = gy.to(dtype).float()
    dpos = sum(gyq[bb, 0] for bb in range(B) if abeta[bb] == 1)
    dneg = sum(gyq[bb, 0] for bb in range(B) if abeta[bb] != 1)
    if torch.is_tensor(dpos):
        assert rel(posg.grad[0, 0], dpos) < 1e-5
    if torch.is_tensor(dneg):
        assert rel(negg.grad[0, 0], dneg) < 1e-5
    # RoiMSE
    w = torch.full((36,), 225.0)
    w[5] = 17.0
    pred = torch.rand((B, 1, *S), generator=g).to(dtype)
    predr = pred.double().requires_grad_(True)
    orc = ORoiMSE(w.double(), ROI_INDICES)
    orc.batch_reduction = None
    lr = orc(predr, b["tau"].to(dtype).double(), b["roi"].double())
    cw = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64).view(B, 1)
    (lr * cw).sum().backward()
    pg = to_int(pred).to(dev).requires_grad_(True)
    lg = ops.RoiMSELoss.apply(pg, to_int(b["tau"]).to(dev, dtype), to_int(b["roi"]).to(dev), ids, w.to(dev))
    assert rel(lg, lr) < 1e-5
    (lg * cw.float().to(dev)).sum().backward()
    assert rel(to_ext(pg.grad), predr.grad) < TOL[dtype]
    # L1
    predr2 = pred.double().requires_grad_(True)
    l1r = (predr2 - b["tau"].to(dtype).double()).abs().mean(dim=(-3, -2, -1))
    (l1r * cw).sum().backward()
    pg2 = to_int(pred).to(dev).requires_grad_(True)
    l1g = ops.L1Loss.apply(pg2, to_int(b["tau"]).to(dev, dtype))
    assert rel(l1g, l1r) < 1e-5
    (l1g * cw.float().to(dev)).sum().backward()
    assert rel(to_ext(pg2.grad), predr2.grad) < TOL[dtype]


def test_add_bcast_copy_mean():
    ops, L = _ops()
    dev = "cuda"
    g = torch.Generator().manual_seed(2)
    a = torch.randn((1, 4, 5, 6, 1), generator=g).to(dev).requires_grad_(True)
    b = torch.randn((3, 4, 5, 6, 1), generator=g).to(dev).requires_grad_(True)
    cat = torch.zeros((3, 4, 5, 6, 2), device=dev)
    y = ops.AddBcast.apply(a, b, ops.Out(cat[..., 0:1]))
    assert rel(y, a.detach() + b.detach()) < 1e-7
    c = ops.Copy.apply(b, ops.Out(cat[..., 1:2]))
    assert rel(cat[..., 1:2], b) == 0.0
    gy = torch.randn((3, 4, 5, 6, 1), generator=g).to(dev)
    (y * gy).sum().backward()
    assert rel(a.grad, gy.sum(0, keepdim=True)) < 1e-6 and rel(b.grad, gy) < 1e-7
    x = torch.randn((2, 6, 7, 8, 5), generator=g).to(dev)
    m = ops.SpatialMean.apply(x)
    assert rel(m, x.mean(dim=(1, 2, 3))) < 1e-6


def test_adamw_matches_torch():
    ops, L = _ops()
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(10007, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3)
    p = p0.clone().to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        gr = torch.randn(10007, generator=g)
        pr.grad = gr.clone()
        opt.step()
        ops.adamw_(p, gr.to(dev), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
    assert rel(p, pr) < 1e-6


THIN_CASES = [
    # cin, cout, k, stride, transposed, dims  (bf16 wgrad goes through the MFMA kernel with zero-padded channels)
    (1, 32, 3, 1, False, (6, 9, 37)), (32, 1, 1, 1, False, (5, 8, 33)), (3, 16, 3, 1, False, (4, 6, 35)),
    (16, 16, 3, 1, False, (4, 5, 18)), (16, 1, 3, 1, False, (5, 5, 17)), (2, 8, 3, 1, False, (3, 4, 40)),
    (32, 16, 1, 1, False, (6, 7, 33)), (256, 128, 1, 1, False, (4, 4, 4)), (2, 1, 1, 1, False, (4, 4, 20)),
    (48, 40, 3, 1, False, (3, 5, 9)),
    # pointwise MFMA kernel (>= 4096 voxels): gate W_g / W_x shapes and their data-gradients
    (32, 16, 1, 1, False, (17, 16, 19)), (16, 32, 1, 1, False, (16, 16, 17)), (64, 32, 1, 1, False, (16, 17, 16)),
    (24, 40, 1, 1, False, (16, 16, 16)), (16, 8, 1, 1, False, (18, 16, 16)),
]


@pytest.mark.parametrize("case", THIN_CASES)
def test_thin_conv_bf16_auto_algo(case):
    """Thin / 1x1x1 layers in bf16 with algo=auto: whichever kernels the library picks must agree with fp64."""
    ops, L = _ops()
    cin, cout, k, s, tr, dims = case
    B = 2
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn((B, cin, *dims), generator=g).bfloat16().double()
    w = (torch.randn((cout, cin, k, k, k), generator=g) * 0.2).bfloat16().double()
    bias = torch.randn((cout,), generator=g).double()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, bias, stride=s, padding=(k - 1) // 2)
    gy = torch.randn(yr.shape, generator=g).bfloat16().double()
    yr.backward(gy)
    dev = "cuda"
    xi = to_int(x).to(dev, torch.bfloat16).requires_grad_(True)
    mg = w.float().to(dev).requires_grad_(True)
    a_f, a_d = ops.pick_algo(xi.shape, xi.dtype, cout, k, s, tr, False, xi.device)
    wd = lambda a: torch.bfloat16 if a == 2 else torch.float32
    wk_f, wk_d = ops.PrepWeights.apply(mg, None, tr, wd(a_f), wd(a_d))
    y = ops.Conv.apply(xi, wk_f, wk_d, bias.float().to(dev), k, s, tr, False, 0, None)
    assert rel(to_ext(y), yr) < 8e-3
    y.backward(to_int(gy).to(dev, torch.bfloat16))
    assert rel(to_ext(xi.grad), xr.grad) < 8e-3
    assert rel(mg.grad, wr.grad) < 8e-3


MFMA_CASES = [
    # cin, cout, k, stride, transposed, dims
    (32, 32, 3, 1, False, (6, 9, 37)), (64, 32, 3, 1, False, (5, 8, 33)), (32, 64, 3, 2, False, (8, 10, 36)),
    (64, 32, 3, 2, True, (4, 5, 17)), (128, 128, 3, 1, False, (4, 4, 8)), (256, 128, 3, 2, True, (2, 3, 4)),
    (32, 64, 1, 1, False, (5, 6, 7)), (64, 64, 3, 1, False, (3, 3, 3)), (32, 32, 3, 2, False, (7, 9, 11)),
    # persistent halo kernel (W >= 32): resident weights (C == 32) and streamed kz-planes (C > 32), ragged grids
    (32, 32, 3, 1, False, (20, 21, 70)), (32, 64, 3, 1, False, (17, 10, 33)), (64, 32, 3, 1, False, (18, 9, 64)),
    (96, 64, 3, 1, False, (3, 5, 40)), (32, 96, 3, 1, False, (35, 4, 32)),
    # deep layers: split-K forward / data-gradient, single-producer (plain-store) weight-gradient tiles
    (512, 384, 3, 1, False, (4, 4, 8)), (256, 512, 3, 2, False, (4, 6, 8)), (512, 256, 3, 2, True, (2, 3, 4)),
    # voxels-along-K weight gradients (conv_bf16_wgrad16_k / conv_f32_wgrad16_k): stride 2 and transposed, coarse width >= 32
    (32, 64, 3, 2, False, (6, 8, 66)), (64, 32, 3, 2, True, (3, 4, 33)), (64, 64, 3, 2, False, (5, 6, 70)),
]


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("case", MFMA_CASES)
def test_conv_mfma_fwd_dgrad_wgrad(case, per_sample):
    """bf16 MFMA implicit-GEMM path (algo=2) against an fp64 reference on bf16-exact operands;
    outputs are rounded to bf16 once (2^-9 relative), accumulation is fp32."""
    ops, L = _ops()
    cin, cout, k, s, tr, dims = case
    B, E = 2, 3
    g = torch.Generator().manual_seed(sum(dims) + cin)
    x = (torch.randn((B, cin, *dims), generator=g) * 1.0).bfloat16().double()
    wshape = (cin, cout, k, k, k) if tr else (cout, cin, k, k, k)
    p = (k - 1) // 2
    if per_sample:
        master = torch.randn((E, *wshape), generator=g) * 0.1
        r = torch.rand((B, E), generator=g)
        wmix = torch.einsum("be,e...->b...", r.double(), master.double()).float().bfloat16().double()
        bias = torch.randn((B, cout), generator=g)
    else:
        master = (torch.randn(wshape, generator=g) * 0.1).bfloat16().float()
        r = None
        wmix = master.double().unsqueeze(0).expand(B, *wshape)
        bias = torch.randn((cout,), generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wmix.clone().requires_grad_(True)

    def ref_conv(xb, w, b):
        if tr:
            return F.conv_transpose3d(xb, w, b, stride=s, padding=p, output_padding=s - 1)
        return F.conv3d(xb, w, b, stride=s, padding=p)

    yr = torch.cat([ref_conv(xr[i:i + 1], wr[i], (bias[i] if per_sample else bias).double()) for i in range(B)], 0)
    gy = torch.randn(yr.shape, generator=g).bfloat16().double()
    yr.backward(gy)
    dev = "cuda"
    xi = to_int(x).to(dev, torch.bfloat16).requires_grad_(True)
    mg = master.to(dev).requires_grad_(True)
    rg = r.to(dev) if per_sample else None
    wk_f, wk_d = ops.PrepWeights.apply(mg, rg, tr, torch.bfloat16, torch.bfloat16)
    y = ops.Conv.apply(xi, wk_f, wk_d, bias.to(dev), k, s, tr, per_sample, 2, None)
    assert rel(to_ext(y), yr) < 5e-3
    wk_f.retain_grad()
    y.backward(to_int(gy).to(dev, torch.bfloat16))
    assert rel(to_ext(xi.grad), xr.grad) < 5e-3
    # kernel-layout weight gradient [Bw, taps, cout, cin] vs reference per-sample weight gradient
    gw = wr.grad    # (B, *wshape)
    if tr:
        gw = gw.permute(0, 2, 1, 3, 4, 5)
    gw = gw.reshape(B, cout, cin, k ** 3).permute(0, 3, 1, 2)
    if not per_sample:
        gw = gw.sum(0, keepdim=True)
    assert rel(wk_f.grad, gw) < 5e-3


F32_MFMA_CASES = MFMA_CASES + [
    # fp32 mode also takes 16-channel inputs and 16-channel outputs on the halo kernels (16 channels = one 64-byte row)
    (16, 16, 3, 1, False, (6, 5, 40)), (16, 32, 3, 1, False, (5, 9, 17)), (48, 16, 3, 1, False, (4, 6, 33)),
    (16, 32, 3, 2, False, (8, 10, 12)), (64, 64, 3, 1, False, (9, 12, 64)), (128, 64, 3, 1, False, (3, 8, 8)),
]


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("case", F32_MFMA_CASES)
def test_conv_f32_mfma_fwd_dgrad_wgrad(case, per_sample):
    """fp32 mode: the same shapes on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation) against an fp64
    reference; only the summation order differs from a CPU fp32 conv, so the bound is the fp32 one (2e-5)."""
    ops, L = _ops()
    cin, cout, k, s, tr, dims = case
    B, E = 2, 3
    g = torch.Generator().manual_seed(sum(dims) + cin + 1)
    x = torch.randn((B, cin, *dims), generator=g).double()
    wshape = (cin, cout, k, k, k) if tr else (cout, cin, k, k, k)
    p = (k - 1) // 2
    if per_sample:
        master = torch.randn((E, *wshape), generator=g) * 0.1
        r = torch.rand((B, E), generator=g)
        wmix = torch.einsum("be,e...->b...", r.double(), master.double())
        bias = torch.randn((B, cout), generator=g)
    else:
        master = torch.randn(wshape, generator=g) * 0.1
        r = None
        wmix = master.double().unsqueeze(0).expand(B, *wshape)
        bias = torch.randn((cout,), generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wmix.clone().requires_grad_(True)

    def ref_conv(xb, w, b):
        if tr:
            return F.conv_transpose3d(xb, w, b, stride=s, padding=p, output_padding=s - 1)
        return F.conv3d(xb, w, b, stride=s, padding=p)

    yr = torch.cat([ref_conv(xr[i:i + 1], wr[i], (bias[i] if per_sample else bias).double()) for i in range(B)], 0)
    gy = torch.randn(yr.shape, generator=g).double()
    yr.backward(gy)
    dev = "cuda"
    xi = to_int(x).to(dev, torch.float32).requires_grad_(True)
    # the library must pick the fp32 MFMA kernels for these shapes (forward, data gradient and weight gradient)
    Bo, Do, Ho, Wo = ops.conv_out_grid(xi.shape, k, s, tr)
    a_f, a_d = ops.pick_algo(tuple(xi.shape), torch.float32, cout, k, s, tr, per_sample, xi.device, 0)
    assert a_f == 3 and (a_d == 3 or (cin % 32 != 0 and s == 2)), (a_f, a_d)   # (a 16-channel strided dgrad output stays on the direct kernel)
    mg = master.to(dev).requires_grad_(True)
    rg = r.to(dev) if per_sample else None
    wk_f, wk_d = ops.PrepWeights.apply(mg, rg, tr, torch.float32, torch.float32)
    y = ops.Conv.apply(xi, wk_f, wk_d, bias.to(dev), k, s, tr, per_sample, 0, None)
    assert rel(to_ext(y), yr) < 2e-5
    wk_f.retain_grad()
    y.backward(to_int(gy).to(dev, torch.float32))
    assert rel(to_ext(xi.grad), xr.grad) < 2e-5
    gw = wr.grad
    if tr:
        gw = gw.permute(0, 2, 1, 3, 4, 5)
    gw = gw.reshape(B, cout, cin, k ** 3).permute(0, 3, 1, 2)
    if not per_sample:
        gw = gw.sum(0, keepdim=True)
    assert rel(wk_f.grad, gw) < 2e-5
    yt = L.Tensor(None, L.F32, B, Do, Ho, Wo, cout, cout, 0)
    xt = L.Tensor(None, L.F32, B, *xi.shape[1:4], cin, cin, 0)
    assert L.lib.coma_conv_wgrad_algo(ops._desc(k, s, 1 if tr else 0, per_sample, 0), xt, yt) == 3


@pytest.mark.parametrize("B,NC,E,N", [(1, 5, 8, 32), (2, 6, 8, 512), (4, 5, 8, 1), (8, 6, 8, 64)])
def test_routing_matches_torch(B, NC, E, N):
    """CondConv routing: sigmoid(Linear(cov)) and the per-sample bias mix, forward and backward,
    against the same algebra in stock torch (fp32 both sides; tolerance 1e-5 relative)."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(B * 100 + N)
    cov = torch.rand((B, NC), generator=g).cuda()
    Wr = (torch.randn((E, NC), generator=g) * 0.5).cuda().requires_grad_(True)
    br = (torch.randn((E,), generator=g) * 0.5).cuda().requires_grad_(True)
    be = torch.randn((E, N), generator=g).cuda().requires_grad_(True)
    gr = torch.randn((B, E), generator=g).cuda()
    gb = torch.randn((B, N), generator=g).cuda()
    r, bm = ops.Routing.apply(cov, Wr, br, be)
    ((r * gr).sum() + (bm * gb).sum()).backward()
    got = [r.detach(), bm.detach(), Wr.grad.clone(), br.grad.clone(), be.grad.clone()]
    for p in (Wr, br, be):
        p.grad = None
    r2 = torch.sigmoid(torch.nn.functional.linear(cov, Wr, br))
    bm2 = r2 @ be
    ((r2 * gr).sum() + (bm2 * gb).sum()).backward()
    want = [r2.detach(), bm2.detach(), Wr.grad, br.grad, be.grad]
    for a, w in zip(got, want):
        assert a.shape == w.shape
        assert float((a - w).abs().max()) <= 1e-5 * max(1.0, float(w.abs().max()))
    # only one of the two outputs used downstream (the other gradient arrives as None)
    for p in (Wr, br, be):
        p.grad = None
    r3, _ = ops.Routing.apply(cov, Wr, br, be)
    (r3 * gr).sum().backward()
    g3 = Wr.grad.clone()
    Wr.grad = None
    (torch.sigmoid(torch.nn.functional.linear(cov, Wr, br)) * gr).sum().backward()
    assert float((g3 - Wr.grad).abs().max()) <= 1e-5 * max(1.0, float(Wr.grad.abs().max()))


@pytest.mark.parametrize("cin,cout", [(1, 32), (2, 8), (3, 16), (8, 1), (16, 3)])
def test_thin_conv_on_padded_pitch_buffers_masks_foreign_lanes(cin, cout):
    """The model keeps 1..3-channel volumes in buffers with an 8-channel pitch (ops._new); the MFMA staging then reads
    16-byte pieces and must mask the lanes that are not the tensor's channels.  Here those lanes hold NaN: forward,
    data-gradient and weight-gradient have to come out finite and equal to the fp64 reference."""
    ops, L = _ops()
    B, dims, k = 2, (6, 9, 37), 3
    g = torch.Generator().manual_seed(cin * 10 + cout)
    x = torch.randn((B, cin, *dims), generator=g).bfloat16().double()
    w = (torch.randn((cout, cin, k, k, k), generator=g) * 0.2).bfloat16().float()
    gy = torch.randn((B, cout, *dims), generator=g).bfloat16().double()
    xr, wr = x.clone().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, padding=1)
    yr.backward(gy)

    def padded(t_ext):                      # (B,C,D,H,W) -> internal view with pitch 8 whose foreign lanes are NaN
        v = to_int(t_ext).to("cuda", torch.bfloat16)
        C = v.shape[-1]
        if C % 8 == 0:
            return v
        buf = torch.full(tuple(v.shape[:4]) + ((C + 7) // 8 * 8,), float("nan"), dtype=torch.bfloat16, device="cuda")
        buf[..., :C] = v
        return buf[..., :C]

    xi = padded(x).requires_grad_(True)
    mg = w.cuda().requires_grad_(True)
    a_f, a_d = ops.pick_algo(xi.shape, xi.dtype, cout, k, 1, False, False, xi.device, 0)
    assert a_f == 2, "expected the MFMA path for this shape"
    wd = lambda a: torch.bfloat16 if a == 2 else torch.float32
    y = ops.ConvLayer.apply(xi, mg, None, None, k, 1, False, 0, None, None, wd(a_f), wd(a_d), True)
    assert torch.isfinite(y.float()).all() and rel(to_ext(y), yr) < 8e-3
    y.backward(padded(gy))
    assert torch.isfinite(xi.grad.float()).all() and rel(to_ext(xi.grad), xr.grad) < 8e-3
    assert torch.isfinite(mg.grad).all() and rel(mg.grad, wr.grad) < 8e-3


THIN16_CASES = [
    # cin, cout, dims, per_sample       (W >= 32: conv_thin16_k; ragged in every direction; several tiles per block)
    (16, 16, (5, 9, 37), False), (16, 16, (4, 6, 70), True), (8, 8, (6, 7, 40), False), (16, 1, (3, 5, 33), False),
    (8, 1, (5, 4, 32), False), (3, 16, (4, 6, 35), False), (2, 8, (3, 4, 40), True), (1, 32, (6, 9, 37), False),
    (16, 3, (4, 5, 36), False), (1, 16, (7, 8, 45), False), (8, 2, (4, 4, 33), False), (12, 10, (5, 5, 34), False),
    (16, 16, (40, 44, 64), False),
]


@pytest.mark.parametrize("mode", ["plain", "instance", "batch"])
@pytest.mark.parametrize("case", THIN16_CASES)
def test_thin16_kernel_fwd_dgrad_and_fused_stats(case, mode):
    """conv_thin16_k (16x16x32 MFMA, taps packed along K) through the C ABI: forward and data-gradient against fp64 on
    bf16-exact operands held in pitch-8 buffers whose foreign lanes are NaN; with `mode` != plain the (mean, rstd) that
    come out of the same launch must equal the statistics of the bf16 output it stored."""
    ops, L = _ops()
    from coma_unet_amd._lib import lib
    cin, cout, dims, per_sample = case
    B, E, k = 2, 3, 3
    g = torch.Generator().manual_seed(cin * 37 + cout + dims[2])
    x = torch.randn((B, cin, *dims), generator=g).bfloat16().double()
    if per_sample:
        master = torch.randn((E, cout, cin, k, k, k), generator=g) * 0.2
        r = torch.rand((B, E), generator=g)
        wmix = torch.einsum("be,e...->b...", r.double(), master.double()).float().bfloat16().double()
        bias = torch.randn((B, cout), generator=g)
    else:
        master = (torch.randn((cout, cin, k, k, k), generator=g) * 0.2).bfloat16().float()
        r = None
        wmix = master.double().unsqueeze(0).expand(B, cout, cin, k, k, k)
        bias = torch.randn((cout,), generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wmix.clone().requires_grad_(True)
    yr = torch.cat([F.conv3d(xr[i:i + 1], wr[i], (bias[i] if per_sample else bias).double(), padding=1) for i in range(B)], 0)
    gy = torch.randn(yr.shape, generator=g).bfloat16().double()
    yr.backward(gy)

    def padded(t_ext):
        v = to_int(t_ext).to("cuda", torch.bfloat16)
        C = v.shape[-1]
        if C % 8 == 0:
            return v
        buf = torch.full(tuple(v.shape[:4]) + ((C + 7) // 8 * 8,), float("nan"), dtype=torch.bfloat16, device="cuda")
        buf[..., :C] = v
        return buf[..., :C]

    xi = padded(x)
    rg = r.cuda() if per_sample else None
    wk_f, wk_d = ops.PrepWeights.apply(master.cuda(), rg, False, torch.bfloat16, torch.bfloat16)
    norm = None
    if mode != "plain":
        n = cout
        norm = (L.NORM_INSTANCE if mode == "instance" else L.NORM_BATCH, 1e-5, None, None, 0.1)
    y, sums = ops._conv_fwd(xi, wk_f, bias.cuda(), k, 1, 0, per_sample, 2, None, norm)
    assert lib.coma_last_kernel().decode().startswith("conv_thin16_k"), lib.coma_last_kernel()
    if norm is not None:      # the kernels derive (mean, rstd) from the fp64 {sum, sumsq} record: same arithmetic here
        mean, rstd = ops.stats_from_sums(sums, y.shape[0] if mode == "instance" else 1, y.shape[4], y.shape[1] * y.shape[2] * y.shape[3] * (1 if mode == "instance" else y.shape[0]), 1e-5)
    assert torch.isfinite(y.float()).all() and rel(to_ext(y), yr) < 5e-3
    if norm is not None:
        yf = y.double()
        red = (1, 2, 3) if mode == "instance" else (0, 1, 2, 3)
        m_ref = yf.mean(red).reshape(mean.shape)
        v_ref = yf.var(red, unbiased=False).reshape(mean.shape)
        assert float((mean.double() - m_ref).abs().max()) < 1e-5 * (1.0 + float(m_ref.abs().max()))
        assert rel(rstd.double(), (v_ref + 1e-5).rsqrt()) < 1e-5
    if mode == "plain":
        dyi = padded(gy)
        dx, _, _ = ops._conv_bwd(xi, wk_d, dyi, k, 1, 0, per_sample, 2, None, True, False, 0, None)
        assert cout > 16 or lib.coma_last_kernel().decode().startswith("conv_thin16_k"), lib.coma_last_kernel()
        assert torch.isfinite(dx.float()).all() and rel(to_ext(dx), xr.grad) < 5e-3
        # weight gradient on the voxels-along-K 16x16x32 kernel (transposed LDS reads): kernel layout [Bw, taps, cout, cin], fp32
        _, dwk, _ = ops._conv_bwd(xi, wk_d, dyi, k, 1, 0, per_sample, 2, tuple(wk_f.shape), False, True, 0, None)
        assert lib.coma_last_kernel().decode().startswith("conv_thin16_wgrad_k"), lib.coma_last_kernel()
        gw = wr.grad.reshape(B, cout, cin, 27).permute(0, 3, 1, 2)
        if not per_sample:
            gw = gw.sum(0, keepdim=True)
        assert torch.isfinite(dwk).all() and rel(dwk, gw) < 2e-5, rel(dwk, gw)      # bf16-exact operands, fp32 accumulation


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roimse_voxel_wise_matches_oracle(dtype):
    """RoiMSE(voxel_wise=True) (criterions.py:135-145,189-198) with the template passed in: per-sample loss and its
    gradient against the CPU restatement (unpinned: the reference's template file is private)."""
    _ops()
    from coma_unet_amd.criterions import RoiMSE
    from coma_unet_amd.roi_tables import ROI_INDICES
    from oracle.criterions_oracle import RoiMSE as ORoiMSE
    g = torch.Generator().manual_seed(11)
    D, H, W = 10, 12, 33
    template = torch.tensor(list(ROI_INDICES) + [0, 0, 0])[torch.randint(0, len(ROI_INDICES) + 3, (D, H, W), generator=g)]
    w = torch.rand(len(ROI_INDICES), generator=g) * 300 + 1
    pred = torch.randn((3, 1, D, H, W), generator=g).to(dtype)
    gt = torch.randn((3, 1, D, H, W), generator=g).to(dtype)
    roi = template.float().expand(3, 1, D, H, W).contiguous()
    oc = ORoiMSE(w, ROI_INDICES, reduction=None, voxel_wise=True, template=template)
    pr = pred.double().requires_grad_(True)
    lo = oc(pr, gt.double(), roi.double())
    go = torch.rand(lo.shape, generator=g).double()
    (lo * go).sum().backward()
    crit = RoiMSE(w, ROI_INDICES, reduction=None, voxel_wise=True, template=template)
    assert abs(float(crit.voxel_weights.mean()) - 5.0) < 1e-4
    pg = pred.cuda().requires_grad_(True)
    lg = crit(pg, gt.cuda(), roi.cuda())
    assert tuple(lg.shape) == tuple(lo.shape)
    assert rel(lg.double().cpu(), lo) < 1e-5
    (lg * go.float().cuda()).sum().backward()
    assert rel(pg.grad.double().cpu(), pr.grad) < (1e-5 if dtype == torch.float32 else 6e-3)
    crit.batch_reduction = "mean"
    assert abs(float(crit(pg, gt.cuda(), roi.cuda())) - float(lo.mean())) < 1e-5 * abs(float(lo.mean()))
    with pytest.raises(NotImplementedError):
        RoiMSE(w, ROI_INDICES, voxel_wise=True)        # no template and no data_util module: the reference's private file


@pytest.mark.parametrize("mode", ["plain", "instance", "batch"])
@pytest.mark.parametrize("case", THIN16_CASES[:12])
def test_thin16f_kernel_fp32_fwd_dgrad_and_fused_stats(case, mode):
    """conv_thin16f_k (exact fp32: v_mfma_f32_16x16x4_f32), the fp32 twin of the test above: fp32 tensors in pitch-8
    buffers whose foreign lanes are NaN, forward / data-gradient against fp64 at fp32 accuracy, fused statistics."""
    ops, L = _ops()
    from coma_unet_amd._lib import lib
    cin, cout, dims, per_sample = case
    B, E, k = 2, 3, 3
    g = torch.Generator().manual_seed(cin * 41 + cout + dims[2])
    x = torch.randn((B, cin, *dims), generator=g).double()
    if per_sample:
        master = torch.randn((E, cout, cin, k, k, k), generator=g) * 0.2
        r = torch.rand((B, E), generator=g)
        wmix = torch.einsum("be,e...->b...", r.double(), master.double())
        bias = torch.randn((B, cout), generator=g)
    else:
        master = torch.randn((cout, cin, k, k, k), generator=g) * 0.2
        r = None
        wmix = master.double().unsqueeze(0).expand(B, cout, cin, k, k, k)
        bias = torch.randn((cout,), generator=g)
    x = x.float().double()
    xr = x.clone().requires_grad_(True)
    wr = wmix.float().double().clone().requires_grad_(True)
    yr = torch.cat([F.conv3d(xr[i:i + 1], wr[i], (bias[i] if per_sample else bias).double(), padding=1) for i in range(B)], 0)
    gy = torch.randn(yr.shape, generator=g).float().double()
    yr.backward(gy)

    def padded(t_ext):
        v = to_int(t_ext).to("cuda", torch.float32)
        C = v.shape[-1]
        if C % 8 == 0:
            return v
        buf = torch.full(tuple(v.shape[:4]) + ((C + 7) // 8 * 8,), float("nan"), dtype=torch.float32, device="cuda")
        buf[..., :C] = v
        return buf[..., :C]

    xi = padded(x)
    rg = r.cuda() if per_sample else None
    wk_f, wk_d = ops.PrepWeights.apply(master.cuda(), rg, False, torch.float32, torch.float32)
    norm = None
    if mode != "plain":
        norm = (L.NORM_INSTANCE if mode == "instance" else L.NORM_BATCH, 1e-5, None, None, 0.1)
    y, sums = ops._conv_fwd(xi, wk_f, bias.cuda(), k, 1, 0, per_sample, 0, None, norm)
    assert lib.coma_last_kernel().decode().startswith("conv_thin16f_k"), lib.coma_last_kernel()
    if norm is not None:
        mean, rstd = ops.stats_from_sums(sums, y.shape[0] if mode == "instance" else 1, y.shape[4], y.shape[1] * y.shape[2] * y.shape[3] * (1 if mode == "instance" else y.shape[0]), 1e-5)
    assert torch.isfinite(y).all() and rel(to_ext(y), yr) < 3e-6
    if norm is not None:
        yf = y.double()
        red = (1, 2, 3) if mode == "instance" else (0, 1, 2, 3)
        m_ref = yf.mean(red).reshape(mean.shape)
        v_ref = yf.var(red, unbiased=False).reshape(mean.shape)
        assert float((mean.double() - m_ref).abs().max()) < 1e-5 * (1.0 + float(m_ref.abs().max()))
        assert rel(rstd.double(), (v_ref + 1e-5).rsqrt()) < 1e-5
    if mode == "plain":
        dyi = padded(gy)
        dx, _, _ = ops._conv_bwd(xi, wk_d, dyi, k, 1, 0, per_sample, 0, None, True, False, 0, None)
        assert cout > 16 or lib.coma_last_kernel().decode().startswith("conv_thin16f_k"), lib.coma_last_kernel()
        assert torch.isfinite(dx).all() and rel(to_ext(dx), xr.grad) < 3e-6
        # weight gradient (kernel layout [Bw, taps, cout, cin]) on the voxels-along-K fp32 MFMA kernel
        _, dwk, _ = ops._conv_bwd(xi, wk_d, dyi, k, 1, 0, per_sample, 0, tuple(wk_f.shape), False, True, 0, None)
        assert lib.coma_last_kernel().decode().startswith("wgrad_replica_sum_k") or \
            lib.coma_last_kernel().decode().startswith("conv_thin16f_wgrad_k"), lib.coma_last_kernel()
        gw = wr.grad.reshape(B, cout, cin, 27).permute(0, 3, 1, 2)
        if not per_sample:
            gw = gw.sum(0, keepdim=True)
        assert torch.isfinite(dwk).all() and rel(dwk, gw) < 2e-5, rel(dwk, gw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [32, 64, 256])
def test_fused_gate_block_matches_torch(C, dtype):
    """ObservableAttentionBlock in training mode = ops.GateFused (csrc/gate.hip: two launches forward, three backward)
    behind the W_g / W_x MFMA convolutions, against the fp64 torch composition of MONAI's AttentionBlock
    (attn_unet_data_parallel.py:139-150): output, psi, every input and parameter gradient, the running statistics; the
    gate output is written into a channel slice of a wider buffer and x carries a GradFork with a second consumer (the
    gate's dx must then ACCUMULATE into the shared buffer)."""
    ops, L = _ops()
    from coma_unet_amd.attn_unet_data_parallel import ObservableAttentionBlock
    from coma_unet_amd.layers import Config
    torch.manual_seed(C)
    Fi = C // 2
    B, dims = 2, (4, 6, 8)
    blk = ObservableAttentionBlock(Config(compute_dtype=dtype), f_int=Fi, f_g=C, f_l=C).cuda()
    blk.train()
    blk.save_attn = True
    with torch.no_grad():
        for bn in (blk.W_g[1], blk.W_x[1], blk.psi[1]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(C + 1)
    mk = lambda c: torch.randn((B, c, *dims), generator=g, dtype=torch.float64).to(dtype).double()
    gq, xq, gy, gy2 = mk(C), mk(C), mk(C), mk(C)
    # ---- fp64 reference
    P = {k: v.detach().double().cpu().clone().requires_grad_(True) for k, v in blk.named_parameters()}
    gr, xr = gq.clone().requires_grad_(True), xq.clone().requires_grad_(True)
    rm = {n: [torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)] for n, c in (("W_g", Fi), ("W_x", Fi), ("psi", 1))}

    def cbn(name, t):
        y = F.conv3d(t, P[f"{name}.0.conv.weight"], P[f"{name}.0.conv.bias"])
        return F.batch_norm(y, rm[name][0], rm[name][1], P[f"{name}.1.weight"], P[f"{name}.1.bias"], True, 0.1, 1e-5)

    s_r = F.relu(cbn("W_g", gr) + cbn("W_x", xr))
    psi_r = torch.sigmoid(cbn("psi", s_r))
    att_r = xr * psi_r
    ((att_r * gy).sum() + (xr * gy2).sum()).backward()          # (second consumer of x: a plain weighted sum)
    # ---- HIP
    dev = "cuda"
    gi = to_int(gq).to(dev, dtype).requires_grad_(True)
    xi = to_int(xq).to(dev, dtype).requires_grad_(True)
    xf = ops.fork(xi)
    cat = torch.zeros((B, *dims, 2 * C), device=dev, dtype=dtype)
    att, psi = blk(gi, xf, out=cat[..., :C])
    assert L.lib.coma_last_kernel() is not None
    other = ops.GateMul.apply(xf, torch.ones((B, *dims, 1), device=dev, dtype=dtype), None)     # second consumer of the fork
    tol = TOL[dtype]
    assert rel(to_ext(att), att_r) < tol and rel(to_ext(psi), psi_r) < tol
    assert float(cat[..., C:].abs().max()) == 0.0
    torch.autograd.backward([att, other], [to_int(gy).to(dev, dtype), to_int(gy2).to(dev, dtype)])
    gt = 4 * tol
    assert rel(to_ext(gi.grad), gr.grad) < gt, rel(to_ext(gi.grad), gr.grad)
    assert rel(to_ext(xi.grad), xr.grad) < gt, rel(to_ext(xi.grad), xr.grad)
    for k, v in blk.named_parameters():
        if k.endswith("0.conv.bias"):          # removed by the BatchNorm behind it: exact zeros here, rounding noise in torch
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        assert v.grad is not None, k
        assert rel(v.grad, P[k].grad) < gt, (k, rel(v.grad, P[k].grad))
    for name in ("W_g", "W_x", "psi"):
        bn = getattr(blk, name)[1]
        assert rel(bn.running_mean, rm[name][0]) < 10 * tol and rel(bn.running_var, rm[name][1]) < 10 * tol, name
        assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grad_fork_accumulates_in_kernel_epilogues(dtype):
    """ops.GradFork: three consumers of one activation -- a 1x1x1 convolution (pointwise / gather kernel: COMA_ACCUMULATE),
    a stride-2 3x3x3 convolution (gather kernel) and a 3x3x3 stride-1 convolution (halo kernel: cannot accumulate, returns
    its gradient the ordinary way) -- must give the sum of the three data gradients, whatever order autograd runs them in."""
    ops, L = _ops()
    from coma_unet_amd.layers import Config, Convolution
    torch.manual_seed(3)
    C, B, dims = 32, 2, (8, 16, 32)          # 4096 voxels: the 1x1x1 layer runs on conv_mfma_pw_k (bf16), 16-byte epilogue
    cfg = Config(compute_dtype=dtype)
    convs = [Convolution(cfg, C, 16, kernel_size=1, conv_only=True).cuda(), Convolution(cfg, C, 64, strides=2, conv_only=True).cuda(),
             Convolution(cfg, C, 32, conv_only=True).cuda()]
    g = torch.Generator().manual_seed(5)
    xq = torch.randn((B, C, *dims), generator=g, dtype=torch.float64).to(dtype).double()
    xr = xq.clone().requires_grad_(True)
    tot = 0.0
    gys, yrs = [], []
    for cv in convs:
        w, b = cv.conv.weight.detach().double().cpu(), cv.conv.bias.detach().double().cpu()
        if dtype == torch.bfloat16:
            w = w.bfloat16().double()
        y = F.conv3d(xr, w, b, stride=cv.s, padding=(cv.k - 1) // 2)
        yrs.append(y.detach())
        gys.append(torch.randn(y.shape, generator=g, dtype=torch.float64).to(dtype).double())
        tot = tot + (y * gys[-1]).sum()
    tot.backward()
    xi = to_int(xq).to("cuda", dtype).requires_grad_(True)
    xf = ops.fork(xi)
    assert getattr(xf, "_coma_fork", None) is not None
    outs = [cv(xf) for cv in convs]
    for o, yr in zip(outs, yrs):
        assert rel(to_ext(o), yr) < TOL[dtype], rel(to_ext(o), yr)
    torch.autograd.backward(outs, [to_int(t).to("cuda", dtype) for t in gys])
    assert rel(to_ext(xi.grad), xr.grad) < 3 * TOL[dtype], rel(to_ext(xi.grad), xr.grad)
    assert xf._coma_fork.buf is None          # the meeting point was handed back


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("dims", [(3, 5, 34), (2, 4, 64), (5, 3, 32)])
def test_tconv_halo_kernel_stats_and_accumulate(dims, dtype):
    """conv_mfma_tconv_k (round 3: the stride-2 transposed convolution / stride-2 data gradient with the coarse halo in
    LDS, 8 output-parity classes from 27 (class, tap) pairs): forward with per-sample weights and bias against fp64
    F.conv_transpose3d, the fused InstanceNorm statistics against the stored output, and COMA_ACCUMULATE (y += conv(x):
    a data gradient added into a buffer another consumer has already written) -- with the kernel that ran asserted."""
    ops, L = _ops()
    lib = L.lib
    cin, cout, B, E = 64, 32, 2, 3
    g = torch.Generator().manual_seed(sum(dims))
    q = (lambda t: t.bfloat16().double()) if dtype == torch.bfloat16 else (lambda t: t.double())
    x = q(torch.randn((B, cin, *dims), generator=g))
    master = torch.randn((E, cin, cout, 3, 3, 3), generator=g) * 0.1
    r = torch.rand((B, E), generator=g)
    wmix = torch.einsum("be,e...->b...", r.double(), master.double())
    if dtype == torch.bfloat16:
        wmix = wmix.float().bfloat16().double()
    bias = torch.randn((B, cout), generator=g)
    yr = torch.cat([F.conv_transpose3d(x[i:i + 1], wmix[i], bias[i].double(), stride=2, padding=1, output_padding=1) for i in range(B)], 0)
    xi = to_int(x).to("cuda", dtype)
    wk_f, _ = ops.PrepWeights.apply(master.cuda(), r.cuda(), True, dtype, None)
    algo = 2 if dtype == torch.bfloat16 else 0
    y, sums = ops._conv_fwd(xi, wk_f, bias.cuda(), 3, 2, 1, True, algo, None, L.NORM_INSTANCE)
    tag = lib.coma_last_kernel().decode()
    assert tag.startswith("conv_mfma_tconv_k<%s" % ("__bf16" if dtype == torch.bfloat16 else "float")), tag
    tol = 5e-3 if dtype == torch.bfloat16 else 2e-5
    assert torch.isfinite(y.float()).all() and rel(to_ext(y), yr) < tol, rel(to_ext(y), yr)
    mean, rstd = ops.stats_from_sums(sums, B, cout, y.shape[1] * y.shape[2] * y.shape[3], 1e-5)
    yf = y.double()
    m_ref, v_ref = yf.mean((1, 2, 3)), yf.var((1, 2, 3), unbiased=False)
    assert float((mean.double() - m_ref).abs().max()) < 1e-5 * (1.0 + float(m_ref.abs().max()))
    assert rel(rstd.double(), (v_ref + 1e-5).rsqrt()) < 1e-5
    # y2 += conv(x) (no bias): the accumulate epilogue
    base = q(torch.randn(yr.shape, generator=g))
    y2 = to_int(base).to("cuda", dtype).contiguous()
    d = ops._desc(3, 2, 1, True, algo)
    cx, cy = L.ct(xi), L.ct(y2)
    assert lib.coma_conv_accumulate_ok(d, cx, cy) == 1
    ws = L.workspace(lib.coma_conv_fwd_ws_bytes(d, cx, cy), xi.device)
    L.check(lib.coma_conv_fwd_ws(d, cx, L.ptr(wk_f), L.dtype_code(wk_f.dtype), None, cy, L.ptr(ws), ws.numel(), L.ACCUMULATE, L.stream()),
            "coma_conv_fwd_ws(accumulate)")
    tag = lib.coma_last_kernel().decode()
    assert tag.startswith("conv_mfma_tconv_k<") and tag.endswith(", 0>"), tag
    want = base + torch.cat([F.conv_transpose3d(x[i:i + 1], wmix[i], None, stride=2, padding=1, output_padding=1) for i in range(B)], 0)
    assert rel(to_ext(y2), want) < 2 * tol, rel(to_ext(y2), want)


@pytest.mark.parametrize("case", [
    # cin, cout, dims, what: the two-group kernel's corner cases -- partial tiles on every axis, one tile only (a group with
    # nothing to do), an odd number of tiles per block, two 32-channel output tiles, the data-gradient (flipped taps) form
    (32, 32, (4, 8, 32), "fwd"), (32, 32, (5, 7, 45), "fwd"), (32, 64, (2, 4, 32), "fwd"), (32, 32, (6, 12, 96), "fwd"),
    (64, 32, (5, 9, 40), "dgrad"), (32, 32, (3, 5, 33), "dgrad"),
    # more than two 16-channel chunks: the weights go through the fragment-order re-layout (duo_relayout_k) and are refetched per step
    (64, 32, (5, 7, 45), "fwd"), (128, 64, (4, 8, 32), "fwd"), (32, 64, (3, 5, 40), "dgrad"), (64, 128, (2, 8, 64), "dgrad"),
    # the 16-wide form (2 x 8 x 16 tiles, an M-tile = two x-rows): exact and ragged grids, many chunks, two output tiles
    (32, 32, (4, 16, 16), "fwd"), (64, 64, (5, 9, 20), "fwd"), (256, 32, (2, 8, 31), "fwd"), (32, 64, (3, 17, 16), "dgrad"),
])
def test_duo_kernel_matches_fp64_reference(case):
    """conv_mfma_duo_k (two 4-wave groups per CU alternating matrix and staging phases; the thick stride-1 layers):
    per-sample weights + bias + fused InstanceNorm statistics against fp64 F.conv3d, with the kernel that ran asserted."""
    ops, L = _ops()
    lib = L.lib
    cin, cout, dims, what = case
    B, E = 2, 3
    g = torch.Generator().manual_seed(cin + cout + sum(dims))
    q = lambda t: t.bfloat16().double()
    master = torch.randn((E, cout, cin, 3, 3, 3), generator=g) * 0.1
    r = torch.rand((B, E), generator=g)
    wmix = torch.einsum("be,e...->b...", r.double(), master.double()).float().bfloat16().double()
    if what == "fwd":
        x = q(torch.randn((B, cin, *dims), generator=g))
        bias = torch.randn((B, cout), generator=g)
        yr = torch.cat([F.conv3d(x[i:i + 1], wmix[i], bias[i].double(), padding=1) for i in range(B)], 0)
        wk_f, _ = ops.PrepWeights.apply(master.cuda(), r.cuda(), False, torch.bfloat16, None)
        y, sums = ops._conv_fwd(to_int(x).to("cuda", torch.bfloat16), wk_f, bias.cuda(), 3, 1, 0, True, 2, None, L.NORM_INSTANCE)
        tag = lib.coma_last_kernel().decode()
        assert tag == "conv_mfma_duo_k<1, %d>" % (5 if dims[2] >= 32 else 4), tag
        assert torch.isfinite(y.float()).all() and rel(to_ext(y), yr) < 5e-3, rel(to_ext(y), yr)
        mean, rstd = ops.stats_from_sums(sums, B, cout, dims[0] * dims[1] * dims[2], 1e-5)
        yf = y.double()
        m_ref, v_ref = yf.mean((1, 2, 3)), yf.var((1, 2, 3), unbiased=False)
        assert float((mean.double() - m_ref).abs().max()) < 1e-5 * (1.0 + float(m_ref.abs().max()))
        assert rel(rstd.double(), (v_ref + 1e-5).rsqrt()) < 1e-5
    else:
        # data gradient of a cin -> cout layer: dy has cout channels... the kernel's input is dy (C = cout = 32)
        dy = q(torch.randn((B, cout, *dims), generator=g))
        dxr = torch.cat([F.conv_transpose3d(dy[i:i + 1], wmix[i], None, padding=1) for i in range(B)], 0)
        _, wk_d = ops.PrepWeights.apply(master.cuda(), r.cuda(), False, torch.bfloat16, torch.bfloat16)
        xi = torch.empty((B, *dims, cin), dtype=torch.bfloat16, device="cuda")
        dx, _, _ = ops._conv_bwd(xi, wk_d, to_int(dy).to("cuda", torch.bfloat16), 3, 1, 0, True, 2, (B, 27, cout, cin), True, False, 0, None)
        tag = lib.coma_last_kernel().decode()
        assert tag == "conv_mfma_duo_k<0, %d>" % (5 if dims[2] >= 32 else 4), tag
        assert torch.isfinite(dx.float()).all() and rel(to_ext(dx), dxr) < 5e-3, rel(to_ext(dx), dxr)


DISPATCH_ROWS = [
    # cin, cout, k, stride, transposed, coarse/in dims, dtype -> kernel that must run forward / data gradient / weight gradient
    (32, 32, 3, 1, False, (4, 8, 32), torch.bfloat16, "conv_mfma_duo_k<0, 5>", "conv_mfma_duo_k<0, 5>", "conv_mfma_wgrad2_k<1, 2, 3, 1>"),
    (64, 32, 3, 1, False, (4, 8, 32), torch.bfloat16, "conv_mfma_duo_k<0, 5>", "conv_mfma_duo_k<0, 5>", "conv_mfma_wgrad2_k<1, 2, 3, 1>"),
    (64, 32, 3, 2, True, (3, 4, 33), torch.bfloat16, "conv_mfma_tconv_k<__bf16", "conv_mfma_gather_k<64, 0, __bf16>", "conv_bf16_wgrad16_k<2, 1>"),
    (32, 64, 3, 2, False, (6, 8, 66), torch.bfloat16, "conv_mfma_gather_k<64, 0, __bf16>", "conv_mfma_tconv_k<__bf16, 0>", "conv_bf16_wgrad16_k<2, 0>"),
    (64, 32, 3, 2, True, (3, 4, 33), torch.float32, "conv_mfma_tconv_k<float", "conv_mfma_gather_k<64, 0, float>", "conv_f32_wgrad16_k<2, 1>"),
    (32, 64, 3, 2, False, (6, 8, 66), torch.float32, "conv_mfma_gather_k<64, 0, float>", "conv_mfma_tconv_k<float, 0>", "conv_f32_wgrad16_k<2, 0>"),
    (16, 16, 3, 1, False, (4, 8, 32), torch.bfloat16, "conv_thin16_k<16, 1>", "conv_thin16_k<16, 1>", "conv_thin16_wgrad_k<16, 1>"),
    (32, 16, 1, 1, False, (8, 16, 32), torch.bfloat16, "conv_mfma_pw_k<2, 1>", "conv_mfma_pw_k<1, 1>", "conv_mfma_wgrad2_k<1, 2, 1, 1>"),
    (32, 32, 3, 1, False, (4, 8, 32), torch.float32, "conv_mfma_halo2_k<2, 16, 1, 1, float>", "conv_mfma_halo2_k<2, 16, 1, 1, float>", "conv_f32_wgrad16_k<1, 0>"),
    (128, 64, 3, 2, True, (4, 4, 16), torch.bfloat16, "conv_mfma_gather_k<64, 1, __bf16>", "conv_mfma_gather_k<128, 0, __bf16>", None),
]


@pytest.mark.parametrize("row", DISPATCH_ROWS)
def test_kernel_dispatch_is_what_the_tables_say(row):
    """Which kernel a layer shape runs is a property of the library's dispatch (and of six environment switches): a
    regression there would silently test -- and time -- an older kernel.  Every row names the variant that must run."""
    ops, L = _ops()
    lib = L.lib
    cin, cout, k, s, tr, dims, dtype, want_f, want_d, want_w = row
    B = 2
    g = torch.Generator().manual_seed(cin + cout)
    xi = torch.randn((B, *dims, cin), generator=g).to("cuda", dtype)
    wshape = (cin, cout, k, k, k) if tr else (cout, cin, k, k, k)
    master = (torch.randn(wshape, generator=g) * 0.1).cuda()
    algo = 0
    a_f, a_d = ops.pick_algo(xi.shape, dtype, cout, k, s, tr, False, xi.device, algo)
    wdt = lambda a: torch.bfloat16 if a == 2 else torch.float32
    wk_f, wk_d = ops.PrepWeights.apply(master, None, tr, wdt(a_f), wdt(a_d))
    form = 1 if tr else 0
    y, _ = ops._conv_fwd(xi, wk_f, None, k, s, form, False, algo, None, None)
    got_f = lib.coma_last_kernel().decode()
    assert got_f.startswith(want_f), (got_f, want_f)
    dy = torch.randn(y.shape, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda").to(dtype)
    ops._conv_bwd(xi, wk_d, dy, k, s, form, False, algo, tuple(wk_f.shape), True, False, 0, None)
    got_d = lib.coma_last_kernel().decode()
    assert got_d.startswith(want_d), (got_d, want_d)
    if want_w is not None:
        ops._conv_bwd(xi, wk_d, dy, k, s, form, False, algo, tuple(wk_f.shape), False, True, 0, None)
        got_w = lib.coma_last_kernel().decode()
        assert got_w.startswith(want_w) or got_w.startswith("wgrad_replica_sum_k"), (got_w, want_w)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("mode", ["instance", "batch"])
@pytest.mark.parametrize("case", [(32, 64, 2, (6, 8, 66)), (64, 128, 2, (8, 8, 16)), (128, 128, 1, (4, 4, 8))])
def test_gather_kernel_fused_stats(case, mode, dtype):
    """conv_mfma_gather_k (strided and small-grid layers) with the norm statistics out of its epilogue: the record must
    describe the STORED output (split-K launches fall back to the separate statistics pass: same record either way)."""
    ops, L = _ops()
    cin, cout, s, dims = case
    B = 2
    g = torch.Generator().manual_seed(cin + sum(dims))
    xi = torch.randn((B, *dims, cin), generator=g).to("cuda", dtype)
    master = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.05).cuda()
    algo = 2 if dtype == torch.bfloat16 else 0
    wk_f, _ = ops.PrepWeights.apply(master, None, False, dtype, None)
    bias = torch.randn((cout,), generator=g).cuda()
    y, sums = ops._conv_fwd(xi, wk_f, bias, 3, s, 0, False, algo, None, L.NORM_INSTANCE if mode == "instance" else L.NORM_BATCH)
    G = B if mode == "instance" else 1
    mean, rstd = ops.stats_from_sums(sums, G, cout, y.shape[1] * y.shape[2] * y.shape[3] * (1 if mode == "instance" else B), 1e-5)
    yf = y.double()
    red = (1, 2, 3) if mode == "instance" else (0, 1, 2, 3)
    m_ref = yf.mean(red).reshape(mean.shape)
    v_ref = yf.var(red, unbiased=False).reshape(mean.shape)
    assert float((mean.double() - m_ref).abs().max()) < 1e-5 * (1.0 + float(m_ref.abs().max()))
    assert rel(rstd.double(), (v_ref + 1e-5).rsqrt()) < 1e-5
