"""Generates the committed fixtures under tests/golden/ (run in the BUILD container only).

ORACLE / TEST INFRASTRUCTURE.  Two kinds of fixture:

1. ``criterions_ref.npz`` -- produced by importing the REFERENCE's own
   ``/root/reference/criterions.py`` (two empty stand-in modules named
   ``data_util`` and ``VolumeDataset`` are put in ``sys.modules`` first; the
   file uses them only in code paths that are not exercised, criterions.py:138,
   401,671).  Pins RnCLoss / LabelDifference / FeatureSimilarity /
   GenerativeContrastiveLoss of ``criterions_oracle.py`` AND RoiMSE
   (criterions.py:181-211): the reference's ``RoiMSE.forward`` allocates its
   mask with ``device=roi.get_device()`` which is -1 for a plain CPU tensor
   (SURVEY F11); handing it ``roi`` as a ``torch.Tensor`` subclass whose
   ``get_device()`` answers "cpu" lets the UNMODIFIED reference code run on the
   CPU.  ``roimse{B}_*`` hold its loss vector and ``pred.grad`` for B = 1, 2, 3
   with the 225/100/7.5 weight mix, and the reference's
   GenerativeContrastiveLoss is driven with the reference's own RoiMSE.
2. ``model32_oracle.npz`` -- outputs of THIS REPO'S oracle model at 32^3
   (BASELINE config C1) from seeded weights and seeded synthetic inputs.  The
   reference model cannot be imported (MONAI / CondConv absent): this fixture
   is a regression pin of the oracle, not a reference output (parity unpinned).

Usage:  python oracle/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference_criterions():
    for name in ("data_util", "VolumeDataset"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, "/root/reference")
    import criterions as ref  # noqa: the reference's own file, this container only
    sys.path.pop(0)
    return ref


class CpuRoi(torch.Tensor):
    """A CPU tensor that answers ``get_device()`` with "cpu" (a plain one answers -1, which the reference's
    ``torch.zeros(..., device=roi.get_device())`` at criterions.py:182 rejects)."""

    def get_device(self):
        return "cpu"


def golden_criterions():
    ref = import_reference_criterions()
    from oracle import criterions_oracle as orc
    g = torch.Generator().manual_seed(1234)
    out = {}
    cases = [(2, 512), (3, 512), (4, 512), (6, 64), (8, 2048), (1, 512)]
    for ci, (n, d) in enumerate(cases):
        feats = torch.relu(torch.randn((n, d), generator=g))
        labels = torch.rand((n, 6), generator=g, dtype=torch.float64).float()
        out[f"rnc{ci}_features"] = feats.numpy()
        out[f"rnc{ci}_labels"] = labels.numpy()
        out[f"rnc{ci}_labeldiff"] = ref.LabelDifference()(labels).numpy()
        out[f"rnc{ci}_featsim"] = ref.FeatureSimilarity()(feats).numpy()
        f = feats.clone().requires_grad_(True)
        loss = ref.RnCLoss()(f, labels)
        out[f"rnc{ci}_loss"] = np.asarray(float(loss))
        if torch.is_tensor(loss) and loss.requires_grad:
            loss.backward()
            out[f"rnc{ci}_grad"] = f.grad.numpy()
        else:
            out[f"rnc{ci}_grad"] = np.zeros_like(feats.numpy())
    # GenerativeContrastiveLoss combination (validation.py:137-154 assembly)
    from oracle.coma_oracle import ROI_INDICES
    B, S = 3, 8
    pred = torch.rand((B, 1, S, S, S), generator=g)
    gt = torch.rand((B, 1, S, S, S), generator=g)
    lab = torch.tensor(ROI_INDICES + [0, 2], dtype=torch.float32)
    roi = lab[torch.randint(0, len(lab), (B, 1, S, S, S), generator=g)]
    w = torch.full((36,), 225.0)
    w[3] = 100.0
    w[20] = 7.5
    gen = ref.RoiMSE(w, ROI_INDICES, voxel_wise=False)        # the REFERENCE's class (validation.py:146)
    gen.batch_reduction = None                                 # attn_unet_data_parallel.py:717
    crit = ref.GenerativeContrastiveLoss(ref.RnCLoss(), gen, torch.nn.TripletMarginLoss(1), 0.0, 1.0)
    feats = torch.relu(torch.randn((B, 512), generator=g))
    labels = torch.rand((B, 6), generator=g)
    fin = torch.relu(torch.randn((B, 1, 1, 1, 2048), generator=g))
    tot, genl, ps, ds = crit(pred, gt, roi.as_subclass(CpuRoi), (fin, torch.zeros_like(fin), torch.zeros_like(fin)),
                             (feats, labels))
    # RoiMSE on its own: loss vector + d(sum loss)/d pred for B = 1, 2, 3 (what train_dp back-propagates, criterions.py:560)
    for Bn in (1, 2, 3):
        p_ = torch.rand((Bn, 1, S, S, S), generator=g).requires_grad_(True)
        g_ = torch.rand((Bn, 1, S, S, S), generator=g)
        r_ = lab[torch.randint(0, len(lab), (Bn, 1, S, S, S), generator=g)]
        lv = gen(p_, g_, r_.as_subclass(CpuRoi))
        torch.sum(lv).backward()
        out.update({f"roimse{Bn}_pred": p_.detach().numpy(), f"roimse{Bn}_gt": g_.numpy(), f"roimse{Bn}_roi": r_.numpy(),
                    f"roimse{Bn}_loss": lv.detach().numpy(), f"roimse{Bn}_grad": p_.grad.numpy()})
    gen_mean = ref.RoiMSE(w, ROI_INDICES, voxel_wise=False)    # default batch_reduction="mean" (criterions.py:206-208)
    out["roimse3_mean"] = np.asarray(float(gen_mean(torch.from_numpy(out["roimse3_pred"]), torch.from_numpy(out["roimse3_gt"]),
                                                    torch.from_numpy(out["roimse3_roi"]).as_subclass(CpuRoi))))
    out.update(gcl_pred=pred.numpy(), gcl_gt=gt.numpy(), gcl_roi=roi.numpy(), gcl_w=w.numpy(),
               gcl_feats=feats.numpy(), gcl_labels=labels.numpy(), gcl_fin=fin.numpy(),
               gcl_total=np.asarray(float(tot)), gcl_gen=genl.detach().numpy(),
               gcl_ps=np.asarray(float(ps)), gcl_ds=np.asarray(float(ds)))
    np.savez_compressed(os.path.join(GOLD, "criterions_ref.npz"), **out)
    print("wrote criterions_ref.npz", len(out), "arrays")


def golden_model32():
    from oracle.coma_oracle import build_reference_model
    from oracle.criterions_oracle import build_reference_criterion, train_step_loss
    from coma_unet_amd.synthetic import make_batch
    out = {}
    for B in (1, 2):
        torch.manual_seed(100 + B)
        m = build_reference_model(volume_shape=(32, 32, 32))
        m.set_save_attn(None)
        m.train(True)
        b = make_batch(B, (32, 32, 32), seed=7 + B)
        res = m(b["mri"], b["covars"], roi_pred_dicts=b["roi_pred_dicts"], sample_roi_mask=b["roi"])
        crit = build_reference_criterion()
        tot, genl, ps, ds = train_step_loss(res, b["tau"], b["roi"], b["covars"], crit)
        tot.backward()
        out[f"b{B}_out"] = res[0].detach().numpy()
        out[f"b{B}_proj4"] = res[1][-1].detach().numpy()
        out[f"b{B}_final_proj_mean"] = np.asarray(float(res[2].mean()))
        out[f"b{B}_total"] = np.asarray(float(tot))
        out[f"b{B}_gen"] = genl.detach().numpy()
        gn = {n: float(p.grad.norm()) for n, p in m.named_parameters() if p.grad is not None}
        keys = ["model.0.conv.1.conv.weight", "model.1.merge.conv.weight", "model.2.conv.weight",
                "model.1.upconv.up.conv.routing.weight", "pos_dynamic_prompt", "neg_dynamic_prompt",
                "general_dynamic_prompt", "final_pred_head.conv.weight",
                "model.1.submodule.1.submodule.1.submodule.1.submodule.conv.1.conv.weight"]
        for k in keys:
            out[f"b{B}_gradnorm::{k}"] = np.asarray(gn.get(k, -1.0))  # -1: grad is None
        out[f"b{B}_n_grad_none"] = np.asarray(sum(p.grad is None for p in m.parameters()))
        m.eval()
        with torch.no_grad():
            ev = m(b["mri"], b["covars"], roi_pred_dicts=b["roi_pred_dicts"], sample_roi_mask=b["roi"])
        out[f"b{B}_eval_out"] = ev.numpy()
    np.savez_compressed(os.path.join(GOLD, "model32_oracle.npz"), **out)
    print("wrote model32_oracle.npz")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    golden_criterions()
    golden_model32()
