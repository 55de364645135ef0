"""Where a tile-chunk of conv_mfma_halo2_k<2, 32, 1, 1, __bf16> spends its cycles: builds a DIAGNOSTIC copy of the library
with -DCOMA_STAMPS (s_memtime brackets around the phases; never shipped, never timed as a whole) and prints the phase shares.

    python profiles/stamps_halo2.py [--cin 64 --cout 32 --size 128]
Phases per (tile, 32-channel chunk), summed over every wave:  0 wait at the top barrier, 1 wait for the prefetched
pieces (vmcnt(0)), 2 LDS stores of halo + weights, 3 second barrier, 4 issue of the next chunk's loads, 5 the 27-tap MFMA
loop, 6 epilogue (per tile), 7 whole kernel, 8 waves."""
import argparse, ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64); ap.add_argument("--cout", type=int, default=32)
ap.add_argument("--size", type=int, default=128); ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--what", default="fwd")
ap.add_argument("--no-stamps", action="store_true", help="diagnostic build with the --define flags only: wall time per launch")
ap.add_argument("--define", action="append", default=[], help="extra -D flags of the diagnostic build (e.g. COMA_ABLATE_STORE)")
a = ap.parse_args()
out = os.path.join(ROOT, "gpurun_out", "stamps")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libcoma_unet_stamps.so")
srcs = ["api.hip", "conv_direct.hip", "conv_point1.hip", "conv_mfma.hip", "norm.hip", "elementwise.hip", "weights.hip", "metrics.hip", "comm.hip"]
objs = []
procs = []
for s_ in srcs:
    o = os.path.join(out, s_.replace(".hip", ".o")); objs.append(o)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", os.path.join(ROOT, "coma_unet_amd", "csrc", s_), "-o", o]
    if s_ == "conv_mfma.hip":
        if not a.no_stamps:
            cmd.insert(1, "-DCOMA_STAMPS")
        for d_ in a.define:
            cmd.insert(1, "-D" + d_)
        procs.append(subprocess.Popen(cmd))
    else:
        src_o = os.path.join(ROOT, "coma_unet_amd", "csrc", s_.replace(".hip", ".o"))
        if os.path.exists(src_o):
            objs[-1] = src_o
        else:
            procs.append(subprocess.Popen(cmd))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
os.environ["COMA_UNET_LIB"] = lib
sys.path.insert(0, ROOT)
import torch
from coma_unet_amd import ops, _lib
dbg = ctypes.CDLL(lib)
S = a.size
x = torch.randn((a.batch, S, S, S, a.cin), device="cuda").bfloat16().requires_grad_(True)
w = (torch.randn((a.cout, a.cin, 3, 3, 3), device="cuda") * 0.05).requires_grad_(True)
wk_f, wk_d = ops.PrepWeights.apply(w, None, False, torch.bfloat16, torch.bfloat16)
buf = (ctypes.c_ulonglong * 16)()
def run():
    if a.what == "fwd":
        ops.Conv.apply(x.detach(), wk_f.detach(), wk_d, None, 3, 1, False, False, 0, None)
    else:
        xx = x.detach().requires_grad_(True)
        ops.Conv.apply(xx, wk_f.detach(), wk_d, None, 3, 1, False, False, 0, None).backward(torch.randn((a.batch, S, S, S, a.cout), device="cuda").bfloat16())
for _ in range(3):
    run()
torch.cuda.synchronize()
if not a.no_stamps:
    dbg.coma_debug_read_stamps(buf, 1)
N = 5
for _ in range(N):
    run()
torch.cuda.synchronize()
if not a.no_stamps:
    dbg.coma_debug_read_stamps(buf, 0)
v = list(buf)
names = ["top barrier", "vmcnt(0) wait", "LDS stores", "2nd barrier", "load issue", "MFMA loop", "epilogue"]
if a.cin <= 16 and a.cout <= 32:      # conv_thin16_k's phases
    names = ["prologue (weights)", "top barrier", "vmcnt(0) wait", "LDS stores + barrier", "MFMA loop (+ prefetch issue)", "epilogue", "-"]
tot = v[7]
import time
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
print(f"wall per launch ({'diagnostic' if a.no_stamps else 'stamped'} build{', ' + ','.join(a.define) if a.define else ''}): {(time.perf_counter() - t0) / 10 * 1e6:.1f} us")
if a.no_stamps:
    sys.exit(0)
print(f"{a.what} {a.cin}->{a.cout} at {S}^3 B={a.batch}: {v[8] / N:.0f} waves per launch, {tot / max(v[8], 1):.0f} cycles per wave (stamped build)")
for n, c in zip(names, v[:7]):
    print(f"  {n:30s} {100.0 * c / tot:5.1f} %   {c / max(v[8], 1):10.0f} cycles per wave")
print(f"  {'other':14s} {100.0 * (tot - sum(v[:7])) / tot:5.1f} %")
