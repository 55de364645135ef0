"""CondConv weight preparation per layer shape: expert mix + re-layout (forward) and its backward (expert-gradient scatter + dr),
microseconds and algorithmic GB/s.   python profiles/microbench_wprep.py      (COMA_WPREP_BWD_OLD=1: the per-pair backward kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coma_unet_amd import ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


E, B = 8, 2
for (cout, cin, tr) in [(512, 512, False), (512, 256, False), (256, 256, False), (128, 128, False), (64, 64, False), (32, 32, False),
                        (256, 512, True), (64, 128, True)]:
    shape = (E, cin, cout, 3, 3, 3) if tr else (E, cout, cin, 3, 3, 3)
    master = torch.randn(shape, device="cuda") * 0.05
    r = torch.rand((B, E), device="cuda")
    nw = cout * cin * 27
    fwd = lambda: ops._prep_fwd(master, r, tr, torch.bfloat16, torch.bfloat16)
    wk_f, wk_d, rr, meta = fwd()
    dwk = torch.randn((B, 27, cout, cin), device="cuda")
    bwd = lambda: ops._prep_bwd(dwk, master, rr, meta, None)
    tf, tb = timed(fwd), timed(bwd)
    bf = E * nw * 4 + 2 * B * nw * 2
    bb = B * nw * 4 + 2 * E * nw * 4
    print(f"{cin:4d}->{cout:4d} {'T' if tr else ' '}: fwd {tf:7.1f} us = {bf / tf / 1e3:6.0f} GB/s   bwd {tb:7.1f} us = {bb / tb / 1e3:6.0f} GB/s")
