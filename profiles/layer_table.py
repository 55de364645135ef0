"""Per-layer convolution table of one eager training step (HIP-event timing from ops.KernelTimer).
usage: python profiles/layer_table.py [--size 128] [--batch 2]  -> prints (kind, x shape, Cout, k, s, form): launches, us, TFLOP/s"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coma_unet_amd as cu
from coma_unet_amd import ops, synthetic, train
from coma_unet_amd.criterions import build_reference_criterion

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = (a.size,) * 3
model = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32, static_prompts=True).to(dev)
model.set_save_attn(None)
model.train(True)
crit = build_reference_criterion(dev)
opt = train.make_optimizer(model, 1e-3)
b = synthetic.make_batch(a.batch, S, seed=1000)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["roi_pred_dicts"] = model._priors(b["roi_pred_dicts"], a.batch, dev)
for _ in range(2):
    train.train_step(model, crit, opt, batch)
torch.cuda.synchronize()
ops.KernelTimer.enabled = True
ops.KernelTimer.records = []
train.train_step(model, crit, opt, batch)
torch.cuda.synchronize()
rows = sorted(ops.KernelTimer.by_layer().items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total conv time {tot:.2f} ms")
for (kind, tag), (n, ms, fl) in rows:
    print(f"{kind:11s} {str(tag):52s} n={n} {ms * 1e3 / n:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {ms:6.3f} ms")
