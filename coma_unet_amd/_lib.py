"""ctypes binding of libcoma_unet.so (include/coma_unet.h).

The library is the product: if it is missing or does not export every symbol the
header declares, importing this module raises -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COMA_UNET_LIB") or os.path.join(_HERE, "libcoma_unet.so")   # (override: diagnostic builds, profiles/stamps_halo2.py)

F32, BF16 = 0, 1
ZEROED_OUT, ZEROED_WS, ACCUMULATE = 1, 2, 4
ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY, ACT_SIGMOID, ACT_PRELU_RELU = range(6)
NORM_BATCH, NORM_INSTANCE = 0, 1


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("B", C.c_int32), ("D", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("ld", C.c_int64), ("sb", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("form", C.c_int32),
                ("per_sample_w", C.c_int32), ("algo", C.c_int32)]


_TP = C.POINTER(Tensor)
_DP = C.POINTER(ConvDesc)
_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every function of include/coma_unet.h
SIGNATURES = {
    "coma_abi_version": (_i32, []),
    "coma_last_error": (C.c_char_p, []),
    "coma_last_kernel": (C.c_char_p, []),
    "coma_weight_prep": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _vp]),
    "coma_weight_prep_pair": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    "coma_weight_prep_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp, _vp, _i32, _vp]),
    "coma_routing_fwd": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "coma_routing_bwd": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "coma_conv_pick_algo": (_i32, [_DP, _TP, _TP]),
    "coma_conv_fwd": (_i32, [_DP, _TP, _vp, _i32, _vp, _TP, _vp]),
    "coma_conv_fwd_ws_bytes": (_sz, [_DP, _TP, _TP]),
    "coma_conv_accumulate_ok": (_i32, [_DP, _TP, _TP]),
    "coma_conv_fwd_ws": (_i32, [_DP, _TP, _vp, _i32, _vp, _TP, _vp, _sz, _i32, _vp]),
    "coma_conv_fwd_norm_stats": (_i32, [_DP, _TP, _vp, _i32, _vp, _TP, _i32, _vp, _vp, _sz, _i32, _vp]),
    "coma_conv_wgrad_algo": (_i32, [_DP, _TP, _TP]),
    "coma_conv_wgrad_ws_bytes": (_sz, [_DP, _TP, _TP]),
    "coma_conv_wgrad_zs_bytes": (_sz, [_DP, _TP, _TP]),
    "coma_conv_wgrad": (_i32, [_DP, _TP, _TP, _vp, _vp, _vp, _sz, _i32, _vp]),
    "coma_norm_ws_bytes": (_sz, [_TP]),
    "coma_norm_stats": (_i32, [_TP, _i32, _vp, _vp]),
    "coma_norm_act_fwd": (_i32, [_TP, _i32, _vp, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _f32, _TP, _vp]),
    "coma_norm_act_bwd": (_i32, [_TP, _TP, _i32, _vp, _f32, _vp, _vp, _i32, _vp, _TP, _vp, _vp, _vp, _vp, _vp]),
    "coma_add_relu_fwd": (_i32, [_TP, _TP, _TP, _vp]),
    "coma_add_relu_bwd": (_i32, [_TP, _TP, _TP, _vp]),
    "coma_gate_mul_fwd": (_i32, [_TP, _TP, _TP, _vp]),
    "coma_gate_mul_bwd": (_i32, [_TP, _TP, _TP, _TP, _i32, _TP, _vp]),
    "coma_gate_mid_fwd": (_i32, [_TP, _TP, _vp, _f32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32,
                                 _TP, _TP, _vp, _vp]),
    "coma_gate_apply_fwd": (_i32, [_TP, _TP, _vp, _f32, _vp, _vp, _vp, _vp, _f32, _TP, _TP, _vp]),
    "coma_gate_apply_bwd": (_i32, [_TP, _TP, _TP, _TP, _vp, _f32, _vp, _vp, _TP, _i32, _TP, _vp, _vp]),
    "coma_gate_mid_bwd": (_i32, [_TP, _TP, _TP, _TP, _TP, _vp, _f32, _vp, _vp, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp,
                                 _TP, _TP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "coma_add": (_i32, [_TP, _TP, _TP, _vp]),
    "coma_cast_copy": (_i32, [_TP, _TP, _vp]),
    "coma_zero_ranges": (_i32, [_vp, _vp, _i32, _i64, _vp]),
    "coma_batch_sum": (_i32, [_TP, _TP, _vp]),
    "coma_spatial_mean": (_i32, [_TP, _vp, _vp, _sz, _vp]),
    "coma_roi_paint_fwd": (_i32, [_TP, _TP, _vp, _vp, _i32, _vp, _vp, _vp, _TP, _vp]),
    "coma_roi_paint_bwd": (_i32, [_TP, _vp, _vp, _vp, _vp]),
    "coma_loss_ws_bytes": (_sz, [_TP]),
    "coma_roi_mse_fwd": (_i32, [_TP, _TP, _TP, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "coma_roi_mse_bwd": (_i32, [_TP, _TP, _vp, _vp, _TP, _vp]),
    "coma_l1_fwd": (_i32, [_TP, _TP, _vp, _vp, _sz, _vp]),
    "coma_l1_bwd": (_i32, [_TP, _TP, _vp, _TP, _vp]),
    "coma_eval_stats": (_i32, [_TP, _TP, _TP, _vp, _i32, _vp, _vp]),
    "coma_ssim_ws_bytes": (_sz, [_TP, _i32]),
    "coma_ssim_partial": (_i32, [_TP, _TP, _vp, _i32, _f32, _f32, _vp, _sz, _vp, _vp]),
    "coma_resample_nearest": (_i32, [_vp, _i32, _i32, _i32, C.c_double, C.c_double, C.c_double, _vp, _i32, _i32, _i32,
                              C.c_double, C.c_double, C.c_double, _f32, _i32, _vp, _vp]),
    "coma_comm_unique_id": (_i32, [_vp]),
    "coma_comm_init": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_void_p)]),
    "coma_comm_destroy": (_i32, [_vp]),
    "coma_allreduce_sum_f32": (_i32, [_vp, _vp, _i64, _vp]),
    "coma_reduce_scatter_sum_f32": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "coma_allgather_f32": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "coma_broadcast_f32": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "coma_event_create": (_i32, [C.POINTER(C.c_void_p)]),
    "coma_event_destroy": (_i32, [_vp]),
    "coma_event_record_external": (_i32, [_vp, _vp]),
    "coma_stream_wait_external": (_i32, [_vp, _vp]),
    "coma_adamw": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
}


class ComaError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"libcoma_unet.so does not export {name}")
        fn.restype, fn.argtypes = res, args
    if lib.coma_abi_version() != 3:
        raise ImportError("libcoma_unet.so ABI version mismatch")
    return lib


lib = _load()


def check(rc: int, what: str):
    if rc != 0:
        raise ComaError(f"{what} failed (rc={rc}): {lib.coma_last_error().decode()}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {dt}")


def ct(t: torch.Tensor) -> Tensor:
    """Describe a (B, D, H, W, C) tensor (possibly a channel slice of a wider buffer)."""
    assert t.dim() == 5, t.shape
    B, D, H, W, Cc = t.shape
    sb, sd, sh, sw, sc = t.stride()
    if Cc == 1:
        sc = 1
    if W == 1:
        sw = sw if sw else Cc
    ld = sw
    assert sc == 1 and (H == 1 or sh == W * ld) and (D == 1 or sd == H * W * ld), (t.shape, t.stride())
    if B == 1:
        sb = D * H * W * ld
    assert t.is_cuda, "the HIP hot path only takes device tensors (no CPU fallback)"
    return Tensor(t.data_ptr(), dtype_code(t.dtype), B, D, H, W, Cc, ld, sb)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_WS = {}
_WS_PINNED = set()      # devices whose scratch buffer's address is baked into a captured hipGraph


def workspace(nbytes: int, device, lane: int = 0) -> torch.Tensor:
    """Stream-ordered scratch buffer, grown on demand: one per device, `lane` and CURRENT STREAM (kernels of the side streams --
    ops.WgradSide's weight gradients, ops.Branch's independent model branches -- run beside the main stream's and must not
    share its scratch).  While a captured graph holds a buffer's raw pointer in its kernel arguments (pin_workspace) the
    cached buffers are never replaced: a larger request gets a one-off allocation instead, so replays keep writing into
    memory that is still theirs."""
    dev = (device.index if device.index is not None else torch.cuda.current_device())
    key = (dev, lane, torch.cuda.current_stream(dev).cuda_stream if torch.cuda.is_available() else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        new = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        if buf is not None and dev in _WS_PINNED:
            return new
        _WS[key] = buf = new
    return buf


def pin_workspace(device) -> torch.Tensor:
    """Called by a graph capture: returns the current scratch buffer (the caller keeps the reference alive)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    _WS_PINNED.add(key)
    return workspace(1, device)
