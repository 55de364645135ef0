"""When do the side-stream weight gradients (ops.WgradSide) actually start?  Device wall-clock stamps (100 MHz counter) written
by a one-thread kernel at every layer's fork point on the main stream and in front of its weight gradient on the side stream,
inside the replayed hipGraph and in eager steps -- no profiler attached (rocprofv3's queue interception perturbs the
cross-queue timing it would be asked to show).  Builds its own tiny diagnostic library with hipcc.

    python profiles/side_stamps.py [--eager]
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coma_unet_amd as cu  # noqa: E402
from coma_unet_amd import ops  # noqa: E402
from coma_unet_amd.synthetic import make_batch  # noqa: E402
from coma_unet_amd.train import GraphedTrainStep, make_optimizer, train_step  # noqa: E402

SRC = r"""
#include <hip/hip_runtime.h>
__global__ void stamp_k(unsigned long long* out) { out[0] = wall_clock64(); }
extern "C" void stamp(void* out, void* stream) { hipLaunchKernelGGL(stamp_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out); }
"""
d = tempfile.mkdtemp()
open(os.path.join(d, "stamp.hip"), "w").write(SRC)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libstamp.so"),
                       os.path.join(d, "stamp.hip")])
lib = ctypes.CDLL(os.path.join(d, "libstamp.so"))
lib.stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.stamp.restype = None

dev = torch.device("cuda")
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
state = {"k": 0, "names": []}


def stamp(slot):
    lib.stamp(buf.data_ptr() + 8 * slot, torch.cuda.current_stream().cuda_stream)


orig_bwd, orig_begin = ops._conv_bwd, ops.WgradSide.begin.__func__


def conv_bwd(x, wk_d, dy, *a, **kw):
    state["cur"] = state["k"]
    state["k"] += 1
    if len(state["names"]) <= state["cur"]:
        state["names"].append(f"{tuple(x.shape)} -> {dy.shape[4]}")
    stamp(2 * state["cur"])                       # main stream, entry of the layer's backward
    return orig_bwd(x, wk_d, dy, *a, **kw)


def begin(cls, devc, entry, *tensors):
    st = orig_begin(cls, devc, entry, *tensors)
    with torch.cuda.stream(st):
        stamp(2 * state["cur"] + 1)               # side stream, in front of the weight gradient
    return st


ops._conv_bwd = conv_bwd
ops.WgradSide.begin = classmethod(begin)

S = (128,) * 3
torch.manual_seed(0)
m = cu.build_model(volume_shape=S, compute_dtype=torch.bfloat16, static_prompts=True).to(dev)
m.set_save_attn(None)
m.train(True)
crit = cu.build_reference_criterion(dev)
opt = make_optimizer(m, 1e-3)
b = make_batch(2, S, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["roi_pred_dicts"] = m._priors(b["roi_pred_dicts"], 2, dev)
eager = "--eager" in sys.argv
if eager:
    for _ in range(4):
        state["k"] = 0
        train_step(m, crit, opt, batch)
else:
    def reset_then(fn):
        def w(*a, **kw):
            state["k"] = 0
            return fn(*a, **kw)
        return w
    import coma_unet_amd.train as T
    T.train_step = reset_then(T.train_step)
    step = GraphedTrainStep(m, crit, opt, batch)
    for _ in range(4):
        step()
torch.cuda.synchronize()
v = buf.cpu().numpy()
n = state["k"]
t0 = min(int(v[2 * i]) for i in range(n) if v[2 * i])
print(f"{'layer':44s} {'fork (main) us':>15s} {'wgrad starts (side) us':>24s} {'lag us':>8s}   [{'eager' if eager else 'graph replay'}]")
for i in range(n):
    a, bb = int(v[2 * i]), int(v[2 * i + 1])
    if not a:
        continue
    fa = (a - t0) / 100.0
    print(f"{state['names'][i][:44]:44s} {fa:15.1f} " + (f"{(bb - t0) / 100.0:24.1f} {(bb - a) / 100.0:8.1f}" if bb else f"{'(inline)':>24s}"))
