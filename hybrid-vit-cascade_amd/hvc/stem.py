"""Stem / head seams of the hot path: the 3-D voxel-embed stem that turns
a volume into tokens, the LN + Linear(C,1) head with the trilinear upsample, and the 2-D X-ray CNN
stem.  The module classes keep the reference's nn.Sequential containers (for parameter names and
initialisation); these functions walk the containers and run every layer on the HIP kernels, with
activations held CHANNELS-LAST so that each convolution is an (implicit) MFMA GEMM and the last stem
layer's output is already the (B, N, C) token matrix.

STAGE_BACKEND records what executes each stage (reported by bench.py): "hip" = hand-written gfx950
kernel through the C ABI; "aten" = a plain torch op on the GPU used as glue.
"""
import torch
import torch.nn as nn

from . import functional as HF
from . import ops

STAGE_BACKEND = {
    "voxel_embed_conv3d": "hip (implicit MFMA GEMM; 1-channel first layer: streaming MFMA kernel, no patch matrix)",
    "voxel_embed_groupnorm_silu": "hip",
    "tokens_pos_embed": "hip (GEMM epilogue)",
    "head_layernorm": "hip",
    "head_proj": "hip",
    "trilinear_upsample": "hip",
    "xray_conv2d": "hip (implicit MFMA GEMM; first layer im2col + GEMM)",
    "xray_batchnorm_relu_pool": "hip",
    "xray_view_mean_gap": "hip (fused)",
    "drr_projection_resize_l1_mse": "hip (fused resize + reduction)",
    "frequency_loss": "rocFFT + hip (fused magnitude / mask / L1)",
    "ssim_l1_loss": "hip",
    "cascade_glue_conv_gn_gelu_upsample": "hip (1 -> 32|64 and C -> 1 layers: streaming kernels; 64 -> 32 k3: LDS halo tile; others implicit MFMA GEMM)",
}


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("the HVC hot path runs on the MI355X HIP device only (got a CPU tensor; there is no CPU fallback)")


def _channels_last(x):
    """(B, C, *spatial) -> (B, *spatial, C); free when C == 1."""
    if x.shape[1] == 1:
        return x.reshape(x.shape[0], *x.shape[2:], 1)
    perm = (0, *range(2, x.dim()), 1)
    return x.permute(*perm).contiguous()


def conv_channels_last(h, layer, cdt, out_dtype=None, addvec=None):
    """nn.Conv3d / nn.Conv2d on a channels-last tensor (B, D, H, W, C) (2-D: D = 1) -> channels-last."""
    is3d = isinstance(layer, nn.Conv3d)
    ks = layer.kernel_size if is3d else (1, *layer.kernel_size)
    pad = layer.padding if is3d else (0, *layer.padding)
    if len(set(layer.stride)) != 1:
        raise RuntimeError("HVC conv: anisotropic strides are not supported")
    geom = ops.ConvGeometry(h.shape[0], layer.in_channels, h.shape[1:4], ks, layer.stride[0], pad)
    out_dtype = out_dtype or cdt
    return HF.ConvFn.apply(h, layer.weight, layer.bias, addvec, geom, cdt, out_dtype)


def glue_sequential(seq, h, cdt):
    """Walks an nn.Sequential of {Upsample(trilinear x2), Conv2d/Conv3d, GroupNorm + GELU/SiLU} on a
    channels-last tensor (B, D, H, W, C) -- the cascade glue of model_progressive.py:37-51,169-174,238-243,259-267."""
    layers = list(seq)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Upsample):
            if h.shape[-1] != 1 or layer.mode != "trilinear":
                raise RuntimeError("HVC glue: only single-channel trilinear upsampling is supported")
            sf = layer.scale_factor
            size = tuple(int(d * sf) for d in h.shape[1:4])
            v = HF.TrilinearFn.apply(h.reshape(h.shape[0], 1, *h.shape[1:4]).float(), size, bool(layer.align_corners))
            h = v.reshape(v.shape[0], *size, 1)
            i += 1
        elif isinstance(layer, (nn.Conv3d, nn.Conv2d)):
            h = conv_channels_last(h, layer, cdt)
            i += 1
        elif isinstance(layer, nn.GroupNorm):
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if isinstance(nxt, nn.GELU):
                act = ops.ACT_GELU_ERF
            elif isinstance(nxt, nn.SiLU):
                act = ops.ACT_SILU
            else:
                raise RuntimeError("HVC glue: GroupNorm must be followed by GELU or SiLU")
            if h.dtype != cdt:
                h = h.to(cdt)
            h = HF.GroupNormSiluFn.apply(h, layer.weight, layer.bias, layer.num_groups, layer.eps, act)
            i += 2
        else:
            raise RuntimeError(f"unexpected layer in cascade glue: {type(layer).__name__}")
    return h


def voxel_tokens(voxel_embed, x, pos_embed, channels_last=False):
    """(B,Cin,D,H,W) [or channels-last (B,D,H,W,Cin)] -> (B,N,C) fp32 tokens, n = (d*H'+h)*W'+w, + pos_embed
    (reference models/hybrid_vit_backbone.py:252-258)."""
    _require_gpu(x)
    cdt = HF.compute_dtype(x)
    h = x if channels_last else _channels_last(x)           # (B, D, H, W, Cin)
    layers = list(voxel_embed)
    i = 0
    fused_pos = False
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Conv3d):
            last = i == len(layers) - 1
            h = conv_channels_last(h, layer, cdt, torch.float32 if last else cdt, pos_embed if last else None)
            fused_pos = last
            i += 1
        elif isinstance(layer, nn.GroupNorm):
            assert isinstance(layers[i + 1], nn.SiLU)
            h = HF.GroupNormSiluFn.apply(h, layer.weight, layer.bias, layer.num_groups, layer.eps)
            i += 2
        else:
            raise RuntimeError(f"unexpected layer in voxel_embed: {type(layer).__name__}")
    tokens = h.reshape(h.shape[0], -1, h.shape[-1])
    if not fused_pos:                                       # stem that ends in GN+SiLU (256^3 geometry)
        tokens = tokens.float() + pos_embed
    return tokens


def token_head(tokens, norm, output_proj, grid):
    """LN -> Linear(C,1) -> (B,1,D',H',W')   (reference models/hybrid_vit_backbone.py:265-269)."""
    B, N, Cn = tokens.shape
    h = HF.layer_norm(tokens, norm.weight, norm.bias, out_dtype=torch.float32)
    y = HF.linear(h, output_proj.weight, output_proj.bias, torch.float32, torch.float32)    # (B,N,1)
    return y.reshape(B, 1, *grid)


def upsample_trilinear(vol, size, align_corners=True):
    """F.interpolate(trilinear): align_corners=True is the ViT head's upsample (reference models/hybrid_vit_backbone.py:272),
    False the trainers' CT resize and the cascade's nn.Upsample (train_progressive_4gpu.py:46-57, model_progressive.py:169)."""
    return HF.TrilinearFn.apply(vol, tuple(size), align_corners)


def xray_encoder(encoder, xrays_flat):
    """Conv / BN / ReLU / MaxPool stack on (B*V,1,H,W) (reference models/diagnostic_losses.py:82-96).
    Returns the feature map channels-last: (B*V, H', W', E)."""
    _require_gpu(xrays_flat)
    cdt = HF.compute_dtype(xrays_flat)
    h = _channels_last(xrays_flat)                          # (N, H, W, Cin)
    layers = list(encoder)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Conv2d):
            geom = ops.ConvGeometry(h.shape[0], layer.in_channels, (1, h.shape[1], h.shape[2]), (1, *layer.kernel_size),
                                    layer.stride[0], (0, *layer.padding))
            y = HF.ConvFn.apply(h.reshape(h.shape[0], 1, *h.shape[1:]), layer.weight, layer.bias, None, geom, cdt, cdt)
            h = y.reshape(y.shape[0], y.shape[2], y.shape[3], y.shape[4])
            i += 1
        elif isinstance(layer, nn.BatchNorm2d):
            assert isinstance(layers[i + 1], nn.ReLU)
            pool = None
            step = 2
            if i + 2 < len(layers) and isinstance(layers[i + 2], nn.MaxPool2d):
                mp = layers[i + 2]
                pool = (int(mp.kernel_size), int(mp.stride), int(mp.padding))
                step = 3
            training = layer.training
            h = HF.BnReluPoolFn.apply(h, layer.weight, layer.bias, layer.running_mean, layer.running_var, pool, training,
                                      layer.eps, layer.momentum if layer.momentum is not None else 0.1)
            if training:
                layer.num_batches_tracked += 1
            i += step
        else:
            raise RuntimeError(f"unexpected layer in xray encoder: {type(layer).__name__}")
    return h
