"""Bandwidth of the normalisation kernels per tensor shape (bf16, channels-last): statistics pass, forward apply,
backward (partial sums + reduce + apply).  Bytes are algorithmic: stats 1T, forward 2T, backward 5T (T = tensor bytes).

    python profiles/microbench_norm.py [--mode instance|batch]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coma_unet_amd import ops, _lib as L  # noqa: E402


def timed(fn, iters=10):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="instance")
    a = ap.parse_args()
    mode = L.NORM_INSTANCE if a.mode == "instance" else L.NORM_BATCH
    for S, C in [(128, 32), (128, 16), (128, 8), (64, 64), (32, 128), (16, 256)]:
        x = torch.randn((2, S, S, S, C), device="cuda").bfloat16()
        gamma = torch.ones(C, device="cuda").requires_grad_(True)
        beta = torch.zeros(C, device="cuda").requires_grad_(True)
        T = x.numel() * 2
        gy = torch.randn_like(x)

        def fwd_full():
            return ops.NormAct.apply(x, gamma, beta, None, None, None, mode, 1, 0.1, 1e-5, True, None)

        xs = x.clone().requires_grad_(True)
        y = ops.NormAct.apply(xs, gamma, beta, None, None, None, mode, 1, 0.1, 1e-5, True, None)

        def bwd():
            torch.autograd.grad(y, (xs, gamma, beta), gy, retain_graph=True)

        t_f, t_b = timed(fwd_full), timed(bwd)
        print(f"{S}^3 x {C:3d} ch (T = {T / 1e6:6.1f} MB): stats+apply {t_f:7.1f} us = {3 * T / t_f / 1e6:5.2f} TB/s (3T), "
              f"backward {t_b:7.1f} us = {5 * T / t_b / 1e6:5.2f} TB/s (5T)")


if __name__ == "__main__":
    main()
