"""Stem / head seams of the hot path: the 3-D voxel-embed stem that turns
a volume into tokens, the LN + Linear(C,1) head with the trilinear upsample, and the 2-D X-ray CNN
stem.  Every function here is the single place where that stage is dispatched, so a stage moves
from its interim backend to its HIP kernel without touching the module classes.

Backend status is recorded in STAGE_BACKEND (reported by bench.py and DESIGN.md): "hip" = hand-written
gfx950 kernel through the C ABI; "miopen" = interim torch op on the GPU (MIOpen / ATen HIP), to be
replaced per the round plan in DESIGN.md.
"""
import torch
import torch.nn.functional as F

from . import functional as HF

STAGE_BACKEND = {
    "voxel_embed_conv3d": "miopen",
    "voxel_embed_groupnorm_silu": "miopen",
    "tokens_pos_embed": "aten",
    "head_layernorm": "hip",
    "head_proj": "hip",
    "trilinear_upsample": "aten",
    "xray_conv2d": "miopen",
    "xray_batchnorm_relu_pool": "miopen",
}


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("the HVC hot path runs on the MI355X HIP device only (got a CPU tensor)")


def voxel_tokens(voxel_embed, x, pos_embed):
    """(B,Cin,D,H,W) -> (B,N,C) fp32 tokens, n = (d*H'+h)*W'+w, + pos_embed
    (reference models/hybrid_vit_backbone.py:252-258)."""
    _require_gpu(x)
    with torch.autocast("cuda", enabled=False):     # interim MIOpen stage runs fp32 (its bf16 3-D conv is unreliable here)
        h = voxel_embed(x.float())
    return h.flatten(2).transpose(1, 2) + pos_embed


def token_head(tokens, norm, output_proj, grid):
    """LN -> Linear(C,1) -> (B,1,D',H',W')   (reference models/hybrid_vit_backbone.py:265-269)."""
    B, N, Cn = tokens.shape
    h = HF.layer_norm(tokens, norm.weight, norm.bias, out_dtype=torch.float32)
    y = HF.linear(h, output_proj.weight, output_proj.bias, torch.float32, torch.float32)    # (B,N,1)
    return y.transpose(1, 2).reshape(B, 1, *grid)


def upsample_trilinear(vol, size):
    """F.interpolate(trilinear, align_corners=True)   (reference models/hybrid_vit_backbone.py:272)."""
    return F.interpolate(vol, size=tuple(size), mode="trilinear", align_corners=True)


def xray_encoder(encoder, xrays_flat):
    """Conv/BN/ReLU/MaxPool stack on (B*V,1,H,W)   (reference models/diagnostic_losses.py:82-96)."""
    _require_gpu(xrays_flat)
    with torch.autocast("cuda", enabled=False):
        return encoder(xrays_flat.float())
