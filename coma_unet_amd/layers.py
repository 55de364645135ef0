"""Building blocks of the MI355X CoMA-UNet: same module tree / state_dict keys as the
reference stack (MONAI ``Convolution``/``ADN``/``attentionunet`` blocks + the reference's
``CondConv`` module), forward and backward executed by the HIP library.

``nn.Conv3d`` / ``nn.BatchNorm3d`` / ``nn.PReLU`` / ``nn.Linear`` instances below are used
as PARAMETER HOLDERS only (identical initialisation and state_dict keys to the reference
stack); their ``forward`` is never called.  All tensors inside the model are channels-last
``(B, D, H, W, C)``.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .ops import Out


class Config:
    """Shared by every layer of one model."""

    def __init__(self, compute_dtype=torch.float32, conv_algo=0, bn_updates_per_forward=1, zero_bias_grad_under_norm=True):
        # A convolution bias that feeds BatchNorm(train)/InstanceNorm is removed again by the normalisation's mean
        # subtraction: its gradient is identically zero.  True returns exact zeros; False reduces dy over the voxels
        # like the reference's autograd does (one more pass over dy per layer, producing rounding noise).
        self.zero_bias_grad_under_norm = zero_bias_grad_under_norm
        self.compute_dtype = compute_dtype
        self.conv_algo = conv_algo            # 0 auto, 1 direct VALU fp32, 2 MFMA bf16
        self.bn_updates_per_forward = bn_updates_per_forward
        self.nbt_pending = None               # list while a model forward collects BatchNorm step counters
        # round 3: the attention gate behind its 1x1x1 convolutions as two fused kernels each way (ops.GateFused) and shared
        # gradient buffers for activations with several consumers (ops.GradFork); COMA_FUSED_GATE=0 / COMA_GRAD_FORKS=0
        # restore the piecewise path (A/B measurements, and the reference composition for the tests)
        import os
        self.fused_gate = os.environ.get("COMA_FUSED_GATE", "1") not in ("0", "")
        self.grad_forks = os.environ.get("COMA_GRAD_FORKS", "1") not in ("0", "")

    def begin_forward(self):
        self.nbt_pending = []

    def end_forward(self):
        if self.nbt_pending:
            torch._foreach_add_(self.nbt_pending, self.bn_updates_per_forward)
        self.nbt_pending = None


def _wdtype(algo):
    return torch.bfloat16 if algo == 2 else torch.float32


_ACTS = {None: L.ACT_NONE, "relu": L.ACT_RELU, "prelu": L.ACT_PRELU, "leakyrelu": L.ACT_LEAKY,
         "sigmoid": L.ACT_SIGMOID, "prelu_relu": L.ACT_PRELU_RELU}


class ADN(nn.Module):
    """MONAI ADN, ordering "NDA" with p=0 dropout: norm (``N``) then activation (``A``)."""

    def __init__(self, cfg, channels, act="prelu", norm="instance"):
        super().__init__()
        self.cfg = cfg
        self.mode = L.NORM_BATCH if norm == "batch" else L.NORM_INSTANCE
        self.act_name = act
        if norm == "batch":
            self.N = nn.BatchNorm3d(channels)
        else:
            self.N = nn.InstanceNorm3d(channels)
        if act in ("prelu", "prelu_relu"):
            self.A = nn.PReLU()
        elif act == "relu":
            self.A = nn.ReLU()
        elif act == "leakyrelu":
            self.A = nn.LeakyReLU(0.01, inplace=True)

    def forward(self, x, out=None):
        return norm_act(self.cfg, x, self.N if self.mode == L.NORM_BATCH else None, self.mode,
                        _ACTS[self.act_name], self.A.weight if self.act_name in ("prelu", "prelu_relu") else None,
                        self.training, out)


def _norm_params(cfg, bn, training):
    gamma = beta = rmean = rvar = None
    momentum, eps = 0.1, 1e-5
    if bn is not None:
        gamma, beta, rmean, rvar = bn.weight, bn.bias, bn.running_mean, bn.running_var
        eps = bn.eps
        k = cfg.bn_updates_per_forward
        momentum = 1.0 - (1.0 - bn.momentum) ** k     # k identical updates folded into one
        if training:
            if cfg.nbt_pending is not None:       # the model adds all counters in one multi-tensor launch
                cfg.nbt_pending.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += k
    return gamma, beta, rmean, rvar, momentum, eps


def norm_act(cfg, x, bn, mode, act, slope, training, out=None, pre=None):
    """`pre`: the statistics record the convolution that wrote `x` produced (ops.ConvLayer with norm=mode)."""
    gamma, beta, rmean, rvar, momentum, eps = _norm_params(cfg, bn, training)
    return ops.NormAct.apply(x, gamma, beta, slope, rmean, rvar, mode, act, momentum, eps, training,
                             Out(out) if out is not None else None, pre)


def conv_norm_act(cfg, x, conv_fn, adn, out=None):
    """conv -> ADN with the norm statistics coming out of the conv pass (MONAI Convolution / CondConvolution)."""
    bn = adn.N if adn.mode == L.NORM_BATCH else None
    slope = adn.A.weight if adn.act_name in ("prelu", "prelu_relu") else None
    if adn.mode == L.NORM_BATCH and not adn.training:     # eval: running statistics, plain conv
        return norm_act(cfg, conv_fn(None), bn, adn.mode, _ACTS[adn.act_name], slope, False, out)
    y, sums = conv_fn(adn.mode)
    return norm_act(cfg, y, bn, adn.mode, _ACTS[adn.act_name], slope, adn.training, out, pre=sums)


def conv_then_bn(cfg, x, cv, bn, act, training, out=None):
    """``nn.Sequential(Convolution(conv_only=True), BatchNorm3d[, act])`` (the attention gate's W_g / W_x / psi,
    attn_unet_data_parallel.py:104-137 via MONAI AttentionBlock): the batch statistics come out of the convolution pass
    and the convolution's bias -- removed again by the BatchNorm -- gets its exact zero gradient."""
    if not training:
        return norm_act(cfg, cv(x), bn, L.NORM_BATCH, act, None, False, out)
    y, sums = conv_plain(cfg, x, cv.conv, cv.k, cv.s, cv.transposed, None, L.NORM_BATCH)
    return norm_act(cfg, y, bn, L.NORM_BATCH, act, None, True, out, pre=sums)


def conv_plain(cfg, x, conv: nn.Module, ksize, stride, transposed, out=None, norm=None):
    """nn.Conv3d / nn.ConvTranspose3d semantics with shared weights."""
    n_out = conv.weight.shape[1] if transposed else conv.weight.shape[0]
    a_f, a_d = ops.pick_algo(x.shape, x.dtype, n_out, ksize, stride, transposed, False, x.device, cfg.conv_algo)
    need_dx = x.requires_grad
    fdt, ddt = _wdtype(a_f), (_wdtype(a_d) if need_dx else None)
    pre = ops.PrepAhead.take(conv.weight, transposed, fdt, ddt)        # prepared at the start of this forward?
    ops.PrepAhead.note(conv.weight, None, None, transposed, fdt, ddt)
    return ops.ConvLayer.apply(x, conv.weight, None, conv.bias, ksize, stride, transposed, cfg.conv_algo,
                               Out(out) if out is not None else None, norm, fdt, ddt,
                               cfg.zero_bias_grad_under_norm, pre[2:] if pre is not None else None)


class Convolution(nn.Module):
    """MONAI ``Convolution`` (3-D): conv (+ ADN)."""

    def __init__(self, cfg, in_channels, out_channels, strides=1, kernel_size=3, act="prelu", norm="instance",
                 conv_only=False, is_transposed=False):
        super().__init__()
        self.cfg, self.k, self.s, self.transposed = cfg, kernel_size, strides, is_transposed
        p = (kernel_size - 1) // 2
        if is_transposed:
            self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=strides, padding=p,
                                           output_padding=strides - 1)
        else:
            self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=strides, padding=p)
        if not conv_only:
            self.adn = ADN(cfg, out_channels, act=act, norm=norm)
        else:
            self.adn = None

    def forward(self, x, out=None):
        if self.adn is None:
            return conv_plain(self.cfg, x, self.conv, self.k, self.s, self.transposed, out)
        return conv_norm_act(self.cfg, x, lambda norm: conv_plain(self.cfg, x, self.conv, self.k, self.s, self.transposed,
                                                                  None, norm), self.adn, out)


class MonaiConvBlock(nn.Module):
    """MONAI ``attentionunet.ConvBlock`` (used by ProjectionHead with kernel_size=1)."""

    def __init__(self, cfg, in_channels, out_channels, kernel_size=3, strides=1):
        super().__init__()
        self.conv = nn.Sequential(
            Convolution(cfg, in_channels, out_channels, strides=strides, kernel_size=kernel_size, act="relu", norm="batch"),
            Convolution(cfg, out_channels, out_channels, strides=1, kernel_size=kernel_size, act="relu", norm="batch"))

    def forward(self, x):
        return self.conv(x)


# ---------------------------------------------------------------------------------------
# CondConv (spec: DESIGN.md; the upstream module is missing, attn_unet_data_parallel.py:28)
# ---------------------------------------------------------------------------------------
class CondConv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, num_experts, num_covars, is_transposed=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.is_transposed = kernel_size, stride, is_transposed
        self.num_experts, self.num_covars = num_experts, num_covars
        k = kernel_size
        shape = (in_channels, out_channels, k, k, k) if is_transposed else (out_channels, in_channels, k, k, k)
        self.weight = nn.Parameter(torch.empty(num_experts, *shape))
        self.bias = nn.Parameter(torch.empty(num_experts, out_channels))
        self.routing = nn.Linear(num_covars, num_experts)
        for e in range(num_experts):
            nn.init.kaiming_uniform_(self.weight[e], a=5 ** 0.5)
        fan_in = self.weight[0].size(1) * k ** 3
        nn.init.uniform_(self.bias, -1.0 / fan_in ** 0.5, 1.0 / fan_in ** 0.5)


_COV_ROWS = {}


def reset_cov_cache():
    """Start of a model forward: the contiguous covariate rows below are per forward (input buffers are updated in place)."""
    _COV_ROWS.clear()


def cov_rows(covariate, B, device):
    """(B, n) fp32 contiguous rows of a covariate tensor or of a slice of it (the U-Net hands `covariate[:, :, :5]` to its
    5-covariate layers: one copy per forward instead of one per layer)."""
    cov = covariate.reshape(B, -1)
    if cov.dtype == torch.float32 and cov.device == device and cov.is_contiguous():
        return cov
    key = (covariate.data_ptr(), tuple(covariate.shape), tuple(covariate.stride()), covariate.dtype)
    c = _COV_ROWS.get(key)
    if c is None:
        c = cov.to(device=device, dtype=torch.float32).contiguous()
        _COV_ROWS[key] = c
    return c


def conv_cond(cfg, x, cc: CondConv3d, covariate, out=None, norm=None):
    B = x.shape[0]
    cov = cov_rows(covariate, B, x.device)
    assert cov.shape[1] == cc.num_covars, (cov.shape, cc.num_covars)
    if ops.SidePrep._on:     # routing reads covariates and parameters only: on the weight-preparation stream (ops.SidePrep)
        with torch.cuda.stream(ops.SidePrep.stream(x.device)):
            r, bias = ops.Routing.apply(cov, cc.routing.weight, cc.routing.bias, cc.bias)
        # allocated from the side stream's pool, read by kernels of this stream (bias: the convolution's epilogue; r: the
        # backward scatter): without this the allocator hands the block to the NEXT layer's routing as soon as python drops it
        main = torch.cuda.current_stream(x.device)
        r.record_stream(main)
        bias.record_stream(main)
        a_f, a_d = ops.pick_algo(x.shape, x.dtype, cc.out_channels, cc.kernel_size, cc.stride, cc.is_transposed, True,
                                 x.device, cfg.conv_algo)
        need_dx = x.requires_grad
        return ops.ConvLayer.apply(x, cc.weight, r, bias, cc.kernel_size, cc.stride, cc.is_transposed, cfg.conv_algo,
                                   Out(out) if out is not None else None, norm, _wdtype(a_f), _wdtype(a_d) if need_dx else None,
                                   cfg.zero_bias_grad_under_norm)
    a_f, a_d = ops.pick_algo(x.shape, x.dtype, cc.out_channels, cc.kernel_size, cc.stride, cc.is_transposed, True,
                             x.device, cfg.conv_algo)
    need_dx = x.requires_grad
    fdt, ddt = _wdtype(a_f), (_wdtype(a_d) if need_dx else None)
    pre = ops.PrepAhead.take(cc.weight, cc.is_transposed, fdt, ddt)    # routing + expert mix done at the start of this forward?
    ops.PrepAhead.note(cc.weight, (cc.routing.weight, cc.routing.bias, cc.bias), cov.shape[1], cc.is_transposed, fdt, ddt)
    r, bias = ops.Routing.apply(cov, cc.routing.weight, cc.routing.bias, cc.bias, pre[:2] if pre is not None else None)    # (B, E), (B, Cout)
    return ops.ConvLayer.apply(x, cc.weight, r, bias, cc.kernel_size, cc.stride, cc.is_transposed, cfg.conv_algo,
                               Out(out) if out is not None else None, norm, fdt, ddt,
                               cfg.zero_bias_grad_under_norm, pre[2:] if pre is not None else None)


class CondConvolution(nn.Module):
    def __init__(self, cfg, in_channels, out_channels, strides=1, kernel_size=3, act="prelu", norm="instance",
                 conv_only=False, is_transposed=False, num_experts=8, num_covars=5):
        super().__init__()
        self.cfg = cfg
        self.conv = CondConv3d(in_channels, out_channels, kernel_size, strides, num_experts, num_covars, is_transposed)
        self.adn = None if conv_only else ADN(cfg, out_channels, act=act, norm=norm)

    def forward(self, x, covariate=None, out=None):
        if self.adn is None:
            return conv_cond(self.cfg, x, self.conv, covariate, out)
        return conv_norm_act(self.cfg, x, lambda norm: conv_cond(self.cfg, x, self.conv, covariate, None, norm),
                             self.adn, out)


class CondConvBlock(nn.Module):
    def __init__(self, cfg, in_channels, out_channels, kernel_size=3, strides=1, num_covars=5, num_experts=8):
        super().__init__()
        self.conv = nn.ModuleList([
            CondConvolution(cfg, in_channels, out_channels, strides=strides, kernel_size=kernel_size, act="relu",
                            norm="batch", num_experts=num_experts, num_covars=num_covars),
            CondConvolution(cfg, out_channels, out_channels, strides=1, kernel_size=kernel_size, act="relu",
                            norm="batch", num_experts=num_experts, num_covars=num_covars)])

    def forward(self, x, covariate=None):
        for c in self.conv:
            x = c(x, covariate)
        return x
