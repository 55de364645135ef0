"""Resample kernel rate: one raw 256 x 256 x 170 (1 x 1 x 1.5 mm) volume -> 128^3 at 2 mm, plus the full 3-volume sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_unet_amd import input_pipeline as P
v = torch.rand((170, 256, 256), device="cuda")
roi = (torch.rand((170, 256, 256), device="cuda") > 0.3).float()
sp = (1.0, 1.0, 1.5)
for _ in range(3):
    P.prepare_sample(v, v, roi, sp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = P.resample_nearest(v, sp, default_value=8.0)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
nout = out.numel()
print(f"resample_nearest {tuple(v.shape)} -> {tuple(out.shape)}: {us:.1f} us, {nout * 8 / us / 1e6:.2f} TB/s algorithmic (4 B read + 4 B write per output voxel)")
e0.record()
for _ in range(20):
    P.prepare_sample(v, v, roi, sp)
e1.record(); torch.cuda.synchronize()
print(f"prepare_sample (3 volumes): {e0.elapsed_time(e1) * 1e3 / 20:.1f} us")
