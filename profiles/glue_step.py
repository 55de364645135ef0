"""Every device activity of ONE eager training step that is NOT a kernel of libcoma_unet.so (ATen kernels, memsets,
device-to-device copies), with the CPU op / autograd node that launched it, its shape and its device time: the launch-diet
worklist.  Also prints the in-order device timeline around every memset / copy (which library kernel follows it).

    python profiles/glue_step.py [size] [--dtype bf16|fp32]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coma_unet_amd as cu  # noqa: E402
from coma_unet_amd.synthetic import make_batch  # noqa: E402
from coma_unet_amd.train import train_step, make_optimizer  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
dt = torch.float32 if "fp32" in sys.argv else torch.bfloat16
S = (size,) * 3
dev = torch.device("cuda")
torch.manual_seed(0)
m = cu.build_model(volume_shape=S, compute_dtype=dt, static_prompts=True).to(dev)
m.set_save_attn(None)
m.train(True)
crit = cu.build_reference_criterion(dev)
opt = make_optimizer(m, 1e-3)
b = make_batch(2, S, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
for _ in range(3):
    train_step(m, crit, opt, batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(m, crit, opt, batch)
    torch.cuda.synchronize()

OURS = ("conv_", "norm_", "stats_", "weight_prep", "routing_", "p1_", "ew3_", "gate_", "roi_paint", "loss_", "adamw_k", "colsum",
        "gather_finalize", "wgrad_replica", "batch_sum", "resample", "_Z")


def ours(name):
    n = name.replace("void ", "")
    return any(n.startswith(p) or ("_k" in n[:60] and p in n[:40]) for p in OURS) and "at::native" not in n and "rocclr" not in n


evs = list(prof.events())
agg = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if any(c.kernels for c in (e.cpu_children or [])):      # keep the leaf-most op that owns the launches
        continue
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name[:48])
        p = p.cpu_parent
    for k in e.kernels:
        if ours(k.name):
            continue
        key = (e.name[:40], " < ".join(chain)[:110], str(e.input_shapes)[:60], k.name.replace("at::native::", "").replace("void ", "")[:70])
        agg[key][0] += 1
        agg[key][1] += k.duration
tot = sum(v[1] for v in agg.values())
print(f"--- non-library device activities of one eager step at {size}^3 {dt}: {sum(v[0] for v in agg.values())} launches, {tot / 1e3:.3f} ms")
for (op, chain, shp, kn), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} {us:9.1f} us  {op:40s} {shp:60s} {kn}\n{'':20s}<- {chain}")

dev_evs = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
print("--- memsets / copies in device order (with the next activity):")
seen = collections.Counter()
for i, e in enumerate(dev_evs):
    n = e.name
    if "emset" in n or "emcpy" in n or "copyBuffer" in n or "fillBuffer" in n:
        nxt = dev_evs[i + 1].name[:70] if i + 1 < len(dev_evs) else "-"
        seen[(n[:40], nxt)] += 1
for (n, nxt), c in seen.most_common(60):
    print(f"{c:4d} {n:40s} -> {nxt}")
