"""MI355X-native (gfx950) implementation of the CoMA-UNet training hot path.

Importing this package loads ``libcoma_unet.so`` (hand-written HIP kernels behind the C ABI
of ``include/coma_unet.h``) and raises if it is missing: there is no CPU/PyTorch fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)
from .attn_unet_data_parallel import (ContrastiveAttentionUNET_DP, ObservableAttentionUnet, AttentionLayer,
                                      ObservableAttentionBlock, UpBlock, ProjectionHead, StackedFusionConvLayers)
from .criterions import RoiMSE, RnCLoss, GenerativeContrastiveLoss, VoxelL1, build_reference_criterion

DEFAULT_MODEL_PARAMS = (3, 1, 1, [32, 64, 128, 256, 512], [2] * 5)   # validation.py:727


def build_model(volume_shape=(128, 128, 128), compute_dtype=None, **kw):
    """The constructor call of validation.py:98."""
    import torch
    return ContrastiveAttentionUNET_DP(*DEFAULT_MODEL_PARAMS, latent_spaces=[2048] * 5, conditional=True,
                                       decoder_ds=kw.pop("decoder_ds", False), volume_shape=volume_shape,
                                       compute_dtype=compute_dtype or torch.float32, **kw)
