"""Host-side cost of one graph replay (time inside graph.replay(), no synchronisation) against the device time of the step.

    python profiles/replay_host_time.py            (COMA_WGRAD_SIDE=0/1 selects the one- or two-branch graph)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coma_unet_amd as cu  # noqa: E402
from coma_unet_amd.synthetic import make_batch  # noqa: E402
from coma_unet_amd.train import GraphedTrainStep, make_optimizer  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = (size,) * 3
dev = torch.device("cuda")
torch.manual_seed(0)
m = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16, static_prompts=True).to(dev)
m.set_save_attn(None)
m.train(True)
crit = cu.build_reference_criterion(dev)
opt = make_optimizer(m, 1e-3)
b = make_batch(2, S, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
step = GraphedTrainStep(m, crit, opt, batch)
for _ in range(3):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    t0 = time.perf_counter()
    step.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print(f"WGRAD_SIDE={os.environ.get('COMA_WGRAD_SIDE', '1')}: replay() returns after {sorted(host)[5]:.2f} ms on the host; step done after {sorted(total)[5]:.2f} ms")
