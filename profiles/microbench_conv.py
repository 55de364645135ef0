"""Single-layer micro-benchmark of the convolution kernels (used for rocprofv3 counter passes).

    python profiles/microbench_conv.py --cin 32 --cout 32 --size 128 --batch 2 --what fwd --iters 10
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coma_unet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--transposed", action="store_true")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--what", default="fwd", choices=["fwd", "dgrad", "wgrad", "all"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--per-sample", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    S = a.size
    x = torch.randn((a.batch, S, S, S, a.cin), device=dev).to(torch.bfloat16 if a.dtype == 'bf16' else torch.float32).requires_grad_(True)
    wshape = (a.cin, a.cout, a.k, a.k, a.k) if a.transposed else (a.cout, a.cin, a.k, a.k, a.k)
    if a.per_sample:
        master = (torch.randn((8, *wshape), device=dev) * 0.05).requires_grad_(True)
        r = torch.rand((a.batch, 8), device=dev)
    else:
        master = (torch.randn(wshape, device=dev) * 0.05).requires_grad_(True)
        r = None
    a_f, a_d = ops.pick_algo(x.shape, x.dtype, a.cout, a.k, a.stride, a.transposed, a.per_sample, x.device)
    wd = lambda al: torch.bfloat16 if al == 2 else torch.float32
    wk_f, wk_d = ops.PrepWeights.apply(master, r, a.transposed, wd(a_f), wd(a_d))
    y = ops.Conv.apply(x, wk_f, wk_d, None, a.k, a.stride, a.transposed, a.per_sample, 0, None)
    gy = torch.randn_like(y)
    flops, _bytes = ops.conv_flops(x.shape, y.shape, a.k, a.stride)

    def run_fwd():
        return ops.Conv.apply(x.detach(), wk_f.detach(), wk_d, None, a.k, a.stride, a.transposed, a.per_sample, 0, None)

    def run_bwd(which):
        xx = x.detach().requires_grad_(which in ("dgrad", "all"))
        ww = wk_f.detach().requires_grad_(which in ("wgrad", "all"))
        yy = ops.Conv.apply(xx, ww, wk_d, None, a.k, a.stride, a.transposed, a.per_sample, 0, None)
        yy.backward(gy)

    ops.KernelTimer.enabled = True
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize()
            ops.KernelTimer.records = []
        if a.what == "fwd":
            run_fwd()
        else:
            run_bwd(a.what)
    torch.cuda.synchronize()
    for (kind, algo), (n, ms, fl, by) in sorted(ops.KernelTimer.summary().items()):
        print(f"{kind}/{algo}: {n} launches, avg {ms / n * 1e3:.1f} us, {fl / (ms * 1e-3) / 1e12:.1f} TFLOP/s, "
              f"{by / (ms * 1e-3) / 1e9:.0f} GB/s algorithmic ({flops / 1e9:.1f} GFLOP per launch)")


if __name__ == "__main__":
    main()
