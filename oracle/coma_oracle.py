"""CPU ORACLE (test infrastructure, NOT product code) -- CoMA-UNet forward path.

This file is a CPU PyTorch restatement of the reference hot path, built from
stock ``torch.nn`` primitives only.  It exists so that the HIP path in
``coma_unet_amd/`` has something to be checked against; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product package never imports it.

What it follows (all paths relative to /root/reference, the read-only upstream):

* wiring, covariate slicing, tuple unrolling, modulator tail, projection heads,
  return contract ........ attn_unet_data_parallel.py:120-241,243-434,436-454,
                           480-501,503-693
* MONAI building blocks (``Convolution``/``ADN``/``attentionunet.ConvBlock``/
  ``UpConv``/``AttentionBlock``/``AttentionLayer``): third-party, NOT vendored
  in the reference and NOT installed here (MONAI version unpinned upstream,
  era ~1.2-1.3).  Restated from the published MONAI source semantics:
  ``Convolution`` = Sequential(conv, ADN) with defaults kernel 3, stride 1,
  same padding, ``adn_ordering="NDA"``, ``act="PRELU"``, ``norm="INSTANCE"``,
  bias=True, transposed => output_padding = stride-1.
* ``CondConv.CondConvBlock`` / ``CondConv.CondConvolution``: the module is
  MISSING upstream (attn_unet_data_parallel.py:28 imports it, the repo does not
  ship it).  The layer semantics implemented here are THIS REPO'S OWN SPEC
  (DESIGN.md "CondConv spec"), recovered from the call sites
  attn_unet_data_parallel.py:126,285-306,318-325,354-367.

PARITY STATUS: **parity unpinned** for the model.  The reference ships no
tests, no golden vectors and no checkpoints for this path, cannot be imported
(MONAI, CondConv absent) and MONAI itself is absent, so nothing in this
container can pin this restatement to reference outputs.  The loss restatement
in ``criterions_oracle.py`` IS pinned against the importable part of the
reference's own ``criterions.py`` (see ``oracle/make_golden.py``).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# ROI tables (attn_unet_data_parallel.py:561-597)
# --------------------------------------------------------------------------
ROI_INDICES = [
    1001, 1006, 1007, 1009, 1015, 1016, 1030, 1034, 1033, 1008, 1025, 1029, 1031, 1022, 17, 18,
    2001, 2006, 2007, 2009, 2015, 2016, 2030, 2034, 2033, 2008, 2025, 2029, 2031, 2022, 49, 50, 51, 52, 53, 54,
]
ROI_INDEX_TO_NAME = {
    1001: "ctx-lh-bankssts", 1006: "ctx-lh-entorhinal", 1007: "ctx-lh-fusiform",
    1009: "ctx-lh-inferiortemporal", 1015: "ctx-lh-middletemporal",
    1016: "ctx-lh-parahippocampal", 1030: "ctx-lh-superiortemporal",
    1034: "ctx-lh-transversetemporal", 1033: "ctx-lh-temporalpole",
    1008: "ctx-lh-inferiorparietal", 1025: "ctx-lh-precuneus",
    1029: "ctx-lh-superiorparietal", 1031: "ctx-lh-supramarginal", 1022: "ctx-lh-postcentral",
    17: "Left-Hippocampus", 18: "Left-Amygdala", 2001: "ctx-rh-bankssts",
    2006: "ctx-rh-entorhinal", 2007: "ctx-rh-fusiform", 2009: "ctx-rh-inferiortemporal",
    2015: "ctx-rh-middletemporal", 2016: "ctx-rh-parahippocampal",
    2030: "ctx-rh-superiortemporal", 2034: "ctx-rh-transversetemporal",
    2033: "ctx-rh-temporalpole", 2008: "ctx-rh-inferiorparietal",
    2025: "ctx-rh-precuneus", 2029: "ctx-rh-superiorparietal", 2031: "ctx-rh-supramarginal",
    2022: "ctx-rh-postcentral", 49: "Right-Thalamus-Proper", 50: "Right-Caudate",
    51: "Right-Putamen", 52: "Right-Pallidum", 53: "Right-Hippocampus",
    54: "Right-Amygdala",
}


# --------------------------------------------------------------------------
# MONAI restatement (third-party; see module docstring)
# --------------------------------------------------------------------------
def _same_padding(kernel_size: int) -> int:
    return (kernel_size - 1) // 2


def _make_act(act):
    """MONAI ``get_act_layer``: a name, or a (name-or-ctor, kwargs) pair."""
    if act is None:
        return None
    kwargs = {}
    if isinstance(act, tuple):
        act, kwargs = act
    if callable(act):
        return act(**kwargs)
    name = act.upper()
    if name == "PRELU":
        return nn.PReLU(**kwargs)  # one shared slope, init 0.25
    if name == "RELU":
        return nn.ReLU(**kwargs)
    if name == "LEAKYRELU":
        return nn.LeakyReLU(**kwargs)
    if name == "SIGMOID":
        return nn.Sigmoid()
    raise ValueError(act)


def _make_norm(norm, channels: int):
    if norm is None:
        return None
    name = norm.upper()
    if name == "INSTANCE":
        return nn.InstanceNorm3d(channels)  # eps 1e-5, no affine, no running stats
    if name == "BATCH":
        return nn.BatchNorm3d(channels)  # eps 1e-5, momentum 0.1, affine
    raise ValueError(norm)


class ADN(nn.Sequential):
    """MONAI ``ADN`` with ordering "NDA"; dropout p=0 is the identity and has no
    parameters, so it is omitted (state_dict unaffected)."""

    def __init__(self, channels: int, act="PRELU", norm="INSTANCE"):
        super().__init__()
        n = _make_norm(norm, channels)
        a = _make_act(act)
        if n is not None:
            self.add_module("N", n)
        if a is not None:
            self.add_module("A", a)


class Convolution(nn.Sequential):
    """MONAI ``blocks.convolutions.Convolution`` (3-D only)."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3,
                 adn_ordering="NDA", act="PRELU", norm="INSTANCE", dropout=None, bias=True,
                 conv_only=False, is_transposed=False, padding=None, output_padding=None):
        super().__init__()
        assert spatial_dims == 3 and adn_ordering == "NDA"
        if padding is None:
            padding = _same_padding(kernel_size)
        if is_transposed:
            if output_padding is None:
                output_padding = strides - 1
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=strides,
                                      padding=padding, output_padding=output_padding, bias=bias)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=strides,
                             padding=padding, bias=bias)
        self.add_module("conv", conv)
        if conv_only:
            return
        if act is None and norm is None:
            return
        self.add_module("adn", ADN(out_channels, act=act, norm=norm))


class MonaiConvBlock(nn.Module):
    """MONAI ``attentionunet.ConvBlock``: 2 x Convolution(relu, BATCH)."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size=3, strides=1, dropout=0.0):
        super().__init__()
        self.conv = nn.Sequential(
            Convolution(spatial_dims, in_channels, out_channels, kernel_size=kernel_size,
                        strides=strides, act="relu", norm="BATCH"),
            Convolution(spatial_dims, out_channels, out_channels, kernel_size=kernel_size,
                        strides=1, act="relu", norm="BATCH"),
        )

    def forward(self, x):
        return self.conv(x)


# --------------------------------------------------------------------------
# CondConv -- THIS REPO'S SPEC (reference module missing)
# --------------------------------------------------------------------------
class CondConv3d(nn.Module):
    """Covariate-routed mixture-of-experts convolution.

    r_b   = sigmoid(routing(covariate_b))                 (E,)
    W_b   = sum_e r_b[e] * weight[e]       b_b = sum_e r_b[e] * bias[e]
    y_b   = conv3d(x_b, W_b) + b_b         (or conv_transpose3d)
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding,
                 num_experts, num_covars, is_transposed=False, output_padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.num_experts, self.num_covars = num_experts, num_covars
        self.is_transposed, self.output_padding = is_transposed, output_padding
        k = kernel_size
        shape = (in_channels, out_channels, k, k, k) if is_transposed else (out_channels, in_channels, k, k, k)
        self.weight = nn.Parameter(torch.empty(num_experts, *shape))
        self.bias = nn.Parameter(torch.empty(num_experts, out_channels)) if bias else None
        self.routing = nn.Linear(num_covars, num_experts)
        self.reset_parameters()

    def reset_parameters(self):
        # each expert initialised like nn.Conv3d / nn.ConvTranspose3d
        for e in range(self.num_experts):
            nn.init.kaiming_uniform_(self.weight[e], a=5 ** 0.5)
        if self.bias is not None:
            fan_in = self.weight[0].size(1) * self.kernel_size ** 3
            bound = 1.0 / fan_in ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, covariate):
        B = x.size(0)
        cov = covariate.reshape(B, -1).to(dtype=self.weight.dtype)
        assert cov.size(1) == self.num_covars, (cov.shape, self.num_covars)
        r = torch.sigmoid(self.routing(cov))  # (B, E)
        w = torch.einsum("be,e...->b...", r, self.weight)
        b = r @ self.bias if self.bias is not None else None
        outs = []
        for i in range(B):
            bi = b[i] if b is not None else None
            if self.is_transposed:
                y = F.conv_transpose3d(x[i:i + 1], w[i], bi, stride=self.stride, padding=self.padding,
                                       output_padding=self.output_padding)
            else:
                y = F.conv3d(x[i:i + 1], w[i], bi, stride=self.stride, padding=self.padding)
            outs.append(y)
        return torch.cat(outs, 0)


class CondConvolution(nn.Module):
    """``CondConv.CondConvolution``: MONAI ``Convolution`` signature + defaults
    (IN + PReLU unless ``conv_only``) with the conv replaced by ``CondConv3d``;
    call sites attn_unet_data_parallel.py:126,296-306."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3,
                 adn_ordering="NDA", act="PRELU", norm="INSTANCE", dropout=None, bias=True,
                 conv_only=False, is_transposed=False, padding=None, output_padding=None,
                 num_experts=8, num_covars=5):
        super().__init__()
        assert spatial_dims == 3
        if padding is None:
            padding = _same_padding(kernel_size)
        if is_transposed and output_padding is None:
            output_padding = strides - 1
        self.conv = CondConv3d(in_channels, out_channels, kernel_size, strides, padding,
                               num_experts, num_covars, is_transposed=is_transposed,
                               output_padding=output_padding or 0, bias=bias)
        self.adn = None
        if not conv_only and not (act is None and norm is None):
            self.adn = ADN(out_channels, act=act, norm=norm)

    def forward(self, x, covariate=None):
        x = self.conv(x, covariate)
        if self.adn is not None:
            x = self.adn(x)
        return x


class CondConvBlock(nn.Module):
    """``CondConv.CondConvBlock``: MONAI ``ConvBlock`` shape -- 2 x
    CondConvolution(relu, BATCH), the first carries the stride; call sites
    attn_unet_data_parallel.py:289-294,318-325,360-367."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size=3, strides=1,
                 dropout=0.0, num_covars=5, num_experts=8):
        super().__init__()
        self.conv = nn.ModuleList([
            CondConvolution(spatial_dims, in_channels, out_channels, strides=strides,
                            kernel_size=kernel_size, act="relu", norm="BATCH",
                            num_experts=num_experts, num_covars=num_covars),
            CondConvolution(spatial_dims, out_channels, out_channels, strides=1,
                            kernel_size=kernel_size, act="relu", norm="BATCH",
                            num_experts=num_experts, num_covars=num_covars),
        ])

    def forward(self, x, covariate=None):
        for c in self.conv:
            x = c(x, covariate)
        return x


# --------------------------------------------------------------------------
# Reference wiring (attn_unet_data_parallel.py)
# --------------------------------------------------------------------------
class UpBlock(nn.Module):
    """attn_unet_data_parallel.py:120-131 (conditional=True branch only: F2)."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=2, kernel_size=3, num_covars=6):
        super().__init__()
        self.up = CondConvolution(spatial_dims, in_channels, out_channels, strides=strides,
                                  kernel_size=kernel_size, is_transposed=True, dropout=0.0,
                                  num_covars=num_covars)

    def forward(self, x, covariate=None):
        return self.up(x, covariate)


class ObservableAttentionBlock(nn.Module):
    """MONAI ``AttentionBlock`` + attn_unet_data_parallel.py:134-150."""

    def __init__(self, spatial_dims, f_int, f_g, f_l):
        super().__init__()
        self.W_g = nn.Sequential(
            Convolution(spatial_dims, f_g, f_int, kernel_size=1, strides=1, padding=0, conv_only=True),
            nn.BatchNorm3d(f_int))
        self.W_x = nn.Sequential(
            Convolution(spatial_dims, f_l, f_int, kernel_size=1, strides=1, padding=0, conv_only=True),
            nn.BatchNorm3d(f_int))
        self.psi = nn.Sequential(
            Convolution(spatial_dims, f_int, 1, kernel_size=1, strides=1, padding=0, conv_only=True),
            nn.BatchNorm3d(1), nn.Sigmoid())
        self.relu = nn.ReLU()
        self.save_attn = None

    def forward(self, g, x):
        g1 = self.W_g(g)
        x1 = self.W_x(x)
        psi = self.relu(g1 + x1)
        psi = self.psi(psi)
        return x * psi


class AttentionLayer(nn.Module):
    """attn_unet_data_parallel.py:152-241 over MONAI ``AttentionLayer``."""

    def __init__(self, spatial_dims, in_channels, out_channels, submodule, up_kernel_size=3,
                 strides=2, num_covars=6):
        super().__init__()
        self.attention = ObservableAttentionBlock(spatial_dims, f_g=in_channels, f_l=in_channels,
                                                  f_int=in_channels // 2)
        self.upconv = UpBlock(spatial_dims, out_channels, in_channels, strides=strides,
                              kernel_size=up_kernel_size, num_covars=num_covars)
        self.merge = Convolution(spatial_dims, 2 * in_channels, in_channels)  # IN + PReLU
        self.submodule = submodule
        self.save_attn = None

    def forward(self, x, covariate=None):
        if isinstance(self.submodule, nn.Sequential):
            x_sub = x
            for sub in self.submodule:
                if isinstance(sub, AttentionLayer):
                    x_sub = sub(x_sub, covariate=covariate)
                else:
                    x_sub = sub(x_sub, covariate=covariate[:, :, :5])
        else:
            x_sub = self.submodule(x, covariate=covariate[:, :, :5])
        if isinstance(x_sub, tuple):
            x_sub, rest = x_sub
        else:
            rest = x_sub
        fromlower = self.upconv(x_sub, covariate)
        att = self.attention(g=fromlower, x=x)
        att_m = self.merge(torch.cat((att, fromlower), dim=1))
        rest = (x_sub, rest)
        return att_m, (x, rest)


class ObservableAttentionUnet(nn.Module):
    """attn_unet_data_parallel.py:243-434 (conditional=True)."""

    def __init__(self, spatial_dims, in_channels, out_channels, channels: Sequence[int],
                 strides: Sequence[int], kernel_size=3, up_kernel_size=3, dropout=0.0,
                 conditional=True):
        super().__init__()
        assert conditional, "only conditional=True is constructible upstream (SURVEY F2)"
        self.dimensions, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.channels, self.strides = list(channels), list(strides)
        self.with_regression = True
        head = CondConvBlock(spatial_dims, in_channels, channels[0], num_covars=5)
        reduce_channels = CondConvolution(spatial_dims, channels[0], out_channels, kernel_size=1,
                                          strides=1, padding=0, conv_only=True, num_experts=8,
                                          num_covars=5 + int(self.with_regression))

        def _create_block(ch, st):
            if len(ch) > 2:
                sub = _create_block(ch[1:], st[1:])
                return AttentionLayer(spatial_dims, ch[0], ch[1],
                                      submodule=nn.Sequential(
                                          CondConvBlock(spatial_dims, ch[0], ch[1], strides=st[0], num_covars=5),
                                          sub),
                                      up_kernel_size=up_kernel_size, strides=st[0], num_covars=6)
            return AttentionLayer(spatial_dims, ch[0], ch[1],
                                  submodule=CondConvBlock(spatial_dims, ch[0], ch[1], strides=st[0], num_covars=5),
                                  up_kernel_size=up_kernel_size, strides=st[0], num_covars=6)

        self.model = nn.ModuleList([head, _create_block(self.channels, self.strides), reduce_channels])

    def forward(self, x, covariate=None):
        enc, dec = [], []
        x = self.model[0](x, covariate=covariate[:, :, :5])
        x, x_submod = self.model[1](x, covariate)
        dec.append(x)
        while isinstance(x_submod, tuple):
            e_i, d_next_rest = x_submod
            enc.append(e_i)
            d_next, rest = d_next_rest
            (dec if isinstance(rest, tuple) else enc).append(d_next)
            x_submod = rest
        x = self.model[2](x, covariate=covariate)
        return x, enc, dec


class ProjectionHead(nn.Module):
    """attn_unet_data_parallel.py:436-454."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = MonaiConvBlock(3, in_channels, 1, kernel_size=1)
        self.act_fn = nn.ReLU()

    def forward(self, x):
        return self.act_fn(self.conv(x).flatten(1))


class StackedFusionConvLayers(nn.Module):
    """attn_unet_data_parallel.py:480-501 (IN + LeakyReLU(0.01))."""

    def __init__(self, cin, cmid, cout, num_convs):
        super().__init__()
        act = (nn.LeakyReLU, {"negative_slope": 1e-2, "inplace": True})
        self.blocks = nn.Sequential(
            *([Convolution(3, cin, cmid, act=act)] +
              [Convolution(3, cmid, cmid, act=act) for _ in range(num_convs - 2)] +
              [Convolution(3, cmid, cout, act=act)]))

    def forward(self, x):
        return self.blocks(x)


class ContrastiveAttentionUNET_DP(ObservableAttentionUnet):
    """attn_unet_data_parallel.py:503-693.

    Stated deviations: ``volume_shape`` kwarg (reference hard-codes 128^3: F9);
    ``double_forward`` reproduces the reference's duplicated U-Net pass (F8,
    :664/:666) and can be switched off; ROI painting takes the same dict list.
    """

    def __init__(self, spatial_dims, in_channels, out_channels, channels, strides, latent_spaces,
                 kernel_size=3, up_kernel_size=3, dropout=0, training=True, embeddings_out=False,
                 conditional=True, decoder_ds=False, **kwargs):
        super().__init__(spatial_dims, in_channels, out_channels, channels, strides, kernel_size,
                         up_kernel_size, dropout, conditional)
        self.training = training
        self.embeddings_out, self.decoder_ds = embeddings_out, decoder_ds
        self.depth = len(channels)
        vs = tuple(kwargs.get("volume_shape", (128, 128, 128)))
        self.volume_shape = vs
        self.double_forward = kwargs.get("double_forward", True)
        self.projection_heads = nn.ModuleList([ProjectionHead(channels[i]) for i in range(len(channels))])
        self.final_projection_head = nn.Sequential(
            nn.AdaptiveAvgPool3d(1), nn.Linear(out_channels, latent_spaces[-1]), nn.ReLU())
        self.pos_dynamic_prompt = nn.Parameter(torch.randn(1, 1, *vs))
        self.neg_dynamic_prompt = nn.Parameter(torch.randn(1, 1, *vs))
        self.fusion_layer = StackedFusionConvLayers(2, 8, 1, 3)
        self.modulator = Convolution(3, 2, 1, act="ReLU")        # unused in forward
        self.modulator_3c = Convolution(3, 3, 1, act="ReLU")     # unused in forward
        self.reweigh = nn.Parameter(torch.ones(vs))              # unused
        self.final_act = nn.ReLU()
        self.pos_reweigh = nn.Parameter(torch.ones((1, *vs)))    # unused
        self.neg_reweigh = nn.Parameter(torch.ones((1, *vs)))    # unused
        self.deep_modulator_3c = StackedFusionConvLayers(3, 16, 1, 3)
        self.final_pred_head = Convolution(3, 2, 1, kernel_size=1)
        self.roi_indices = list(ROI_INDICES)
        self.roi_ind_names_dict = dict(ROI_INDEX_TO_NAME)
        self.general_dynamic_prompt = nn.Parameter(torch.randn(1, 1, *vs))
        self.roi_wise_reweigh = nn.ParameterList([nn.Parameter(torch.ones(1)) for _ in self.roi_indices])
        self.save_attn = None

    def set_save_attn(self, v):
        self.save_attn = v

    def set_training(self, mode):
        self.training = mode

    def get_depth(self):
        return self.depth

    def forward_modulator_with_uq(self, x, out, covariate=None, roi_pred_dicts=None, sample_roi_mask=None):
        dyn, neutral = [], []
        suvr = torch.zeros_like(out)
        sal = torch.zeros_like(out)
        for b in range(x.size(0)):
            prompt_idx = covariate[b, ..., 0].item()
            dyn.append(self.pos_dynamic_prompt if prompt_idx == 1 else self.neg_dynamic_prompt)
            neutral.append(self.general_dynamic_prompt)
            for roi_idx in self.roi_indices:
                name = self.roi_ind_names_dict[roi_idx]
                m = sample_roi_mask[b] == roi_idx
                suvr[b][m] = float(np.nan_to_num(roi_pred_dicts[b][name]["loc"]))
                sal[b][m] = float(np.nan_to_num(roi_pred_dicts[b][name]["std"]))
        suvr = torch.where(x < 1e-04, torch.zeros_like(suvr), suvr)
        sal = torch.where(x < 1e-04, torch.zeros_like(sal), sal)
        dyn = torch.vstack(dyn)
        neutral = torch.vstack(neutral)
        modulated = neutral + self.deep_modulator_3c(torch.cat((dyn, sal, suvr), dim=1))
        final = self.final_pred_head(torch.cat((out, self.fusion_layer(torch.cat((modulated, out), dim=1))), dim=1))
        return self.final_act(final)

    def forward(self, x, covariate=None, roi_pred_dicts=None, sample_roi_mask=None):
        if self.double_forward:  # F8: first result discarded, BN running stats updated twice
            super().forward(x, covariate)
        out, enc, dec = super().forward(x, covariate)
        out = self.forward_modulator_with_uq(x, out, covariate, roi_pred_dicts, sample_roi_mask)
        if not self.training and not self.embeddings_out:
            return out
        projected = [self.projection_heads[i](enc[i]) for i in range(self.depth)]
        final_proj = self.final_projection_head(out)
        if self.embeddings_out:
            return out, projected, final_proj, enc
        if self.decoder_ds:
            return out, projected, final_proj, []
        return out, projected, final_proj


DEFAULT_MODEL_PARAMS = (3, 1, 1, [32, 64, 128, 256, 512], [2] * 5)  # validation.py:727


def build_reference_model(volume_shape=(128, 128, 128), **kw):
    """Constructor call of validation.py:98."""
    return ContrastiveAttentionUNET_DP(*DEFAULT_MODEL_PARAMS, latent_spaces=[2048] * 5, conditional=True,
                                       decoder_ds=False, volume_shape=volume_shape, **kw)
