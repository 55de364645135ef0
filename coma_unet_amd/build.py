"""Builds libcoma_unet.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcoma_unet.so")
SOURCES = ["api.hip", "conv_direct.hip", "conv_point1.hip", "conv_mfma.hip", "norm.hip", "gate.hip", "elementwise.hip", "weights.hip", "metrics.hip", "comm.hip"]


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "coma_unet.h")]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
