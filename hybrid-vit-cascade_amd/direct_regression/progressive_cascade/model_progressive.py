"""MI355X-native counterpart of the reference's direct_regression/progressive_cascade/model_progressive.py
(MultiScaleXrayEncoder :16, Stage1Base64 :86, Stage2Refiner128 :152, Stage3Refiner256 :219,
ProgressiveCascadeModel :319).  Same constructor kwargs, submodule / parameter names and forward contracts;
the ViT stages, the X-ray stem and the glue layers run on the HIP kernels with channels-last activations.

Token-grid note: the 128^3 refiner uses the 32^3 grid its stem really emits
(the reference's own 25^3 pos_embed makes its forward raise); pass token_grid=16 for the author's variant.
"""
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:          # the reference inserts '../..' relative to the cwd (model_progressive.py:10)
    sys.path.insert(0, _PKG)

from hvc import functional as HF  # noqa: E402
from hvc import stem as HS  # noqa: E402
from models.diagnostic_losses import XrayConditioningModule  # noqa: E402
from models.hybrid_vit_backbone import HybridViT3D  # noqa: E402


def _tokens_of(features_2d):
    """(B, C, H, W) feature map -> (B, H*W, C) context tokens (copy-free for the channels-last maps the stem emits)."""
    return features_2d.flatten(2).transpose(1, 2)


def _glue_2d(seq, feats, cdt):
    """Conv2d + GroupNorm + GELU stack on a (B, C, H, W) map held channels-last in memory."""
    h = feats.permute(0, 2, 3, 1)
    if not h.is_contiguous():
        h = h.contiguous()
    h = HS.glue_sequential(seq, h.unsqueeze(1), cdt)                 # (B, 1, H', W', C)
    return h.squeeze(1).permute(0, 3, 1, 2)                          # (B, C, H', W') view


class MultiScaleXrayEncoder(nn.Module):
    def __init__(self, img_size=512, in_channels=1, base_dim=512, num_views=2):
        super().__init__()
        self.xray_encoder = XrayConditioningModule(img_size=img_size, in_channels=in_channels, embed_dim=base_dim,
                                                   num_views=num_views, time_embed_dim=256, cond_dim=1024,
                                                   share_view_weights=False)
        self.to_stage1 = nn.Sequential(
            nn.Conv2d(base_dim, base_dim, 3, stride=2, padding=1), nn.GroupNorm(32, base_dim), nn.GELU(),
            nn.Conv2d(base_dim, base_dim, 3, stride=2, padding=1), nn.GroupNorm(32, base_dim), nn.GELU())
        self.to_stage2 = nn.Sequential(
            nn.Conv2d(base_dim, base_dim, 3, stride=2, padding=1), nn.GroupNorm(32, base_dim), nn.GELU())

    def forward(self, xrays, stage=1):
        B = xrays.shape[0]
        dummy_t = torch.zeros(B, 256, device=xrays.device)
        xray_context, time_xray_cond, feats = self.xray_encoder(xrays, dummy_t)
        cdt = HF.compute_dtype(feats)
        if stage == 1:
            feats = _glue_2d(self.to_stage1, feats, cdt)
        elif stage == 2:
            feats = _glue_2d(self.to_stage2, feats, cdt)
        return feats, time_xray_cond, xray_context


class Stage1Base64(nn.Module):
    def __init__(self, volume_size=(64, 64, 64), xray_img_size=512, voxel_dim=256, vit_depth=4, num_heads=4,
                 xray_feature_dim=512):
        super().__init__()
        self.volume_size = tuple(volume_size)
        self.xray_encoder = MultiScaleXrayEncoder(img_size=xray_img_size, in_channels=1, base_dim=xray_feature_dim, num_views=2)
        self.vit_backbone = HybridViT3D(volume_size=self.volume_size, in_channels=1, voxel_dim=voxel_dim, depth=vit_depth,
                                        num_heads=num_heads, context_dim=xray_feature_dim, cond_dim=1024, use_prev_stage=False)
        D, H, W = self.volume_size
        self.initial_volume = nn.Parameter(torch.randn(1, 1, D, H, W) * 0.01)

    def forward(self, xrays):
        B = xrays.shape[0]
        feats, cond, _ = self.xray_encoder(xrays, stage=1)
        x = self.initial_volume.expand(B, -1, -1, -1, -1)
        return self.vit_backbone(x=x, context=_tokens_of(feats), cond=cond, prev_stage_embed=None)


def _volume_channels_last(v):
    """(B, 1, D, H, W) -> (B, D, H, W, 1) view."""
    return v.reshape(v.shape[0], *v.shape[2:], 1)


def _resize(v, size):
    """F.interpolate(v, size, mode='trilinear', align_corners=False) on (B,1,d,h,w)."""
    return HF.TrilinearFn.apply(v.float(), tuple(size), False)


class Stage2Refiner128(nn.Module):
    def __init__(self, volume_size=(128, 128, 128), voxel_dim=256, vit_depth=6, num_heads=8, xray_feature_dim=512,
                 token_grid=None):
        super().__init__()
        self.volume_size = tuple(volume_size)
        self.upsample_from_64 = nn.Sequential(nn.Upsample(scale_factor=2, mode="trilinear", align_corners=False),
                                              nn.Conv3d(1, 32, 3, padding=1), nn.GroupNorm(8, 32), nn.GELU())
        self.vit_refiner = HybridViT3D(volume_size=self.volume_size, in_channels=32, voxel_dim=voxel_dim, depth=vit_depth,
                                       num_heads=num_heads, context_dim=xray_feature_dim, cond_dim=1024,
                                       use_prev_stage=False, token_grid=token_grid)
        self.residual_weight = nn.Parameter(torch.ones(1) * 0.5)

    def forward(self, volume_64, xray_features_2d, time_xray_cond):
        cdt = HF.compute_dtype(volume_64)
        x = HS.glue_sequential(self.upsample_from_64, _volume_channels_last(volume_64), cdt)      # (B,128,128,128,32)
        refinement = self.vit_refiner(x=x, context=_tokens_of(xray_features_2d), cond=time_xray_cond,
                                      prev_stage_embed=None, channels_last=True)
        return _resize(volume_64, self.volume_size) + self.residual_weight * refinement


class Stage3Refiner256(nn.Module):
    def __init__(self, volume_size=(256, 256, 256), voxel_dim=256, vit_depth=8, num_heads=8, xray_feature_dim=512,
                 use_gradient_checkpointing=True):
        super().__init__()
        self.volume_size = tuple(volume_size)
        self.use_gradient_checkpointing = use_gradient_checkpointing
        self.upsample_from_128 = nn.Sequential(nn.Upsample(scale_factor=2, mode="trilinear", align_corners=False),
                                               nn.Conv3d(1, 32, 3, padding=1), nn.GroupNorm(8, 32), nn.GELU())
        self.vit_refiner = HybridViT3D(volume_size=self.volume_size, in_channels=32, voxel_dim=voxel_dim, depth=vit_depth,
                                       num_heads=num_heads, context_dim=xray_feature_dim, cond_dim=1024, use_prev_stage=False)
        self.detail_enhancer = nn.Sequential(nn.Conv3d(1, 64, 3, padding=1), nn.GroupNorm(16, 64), nn.GELU(),
                                             nn.Conv3d(64, 32, 3, padding=1), nn.GroupNorm(8, 32), nn.GELU(),
                                             nn.Conv3d(32, 1, 1))
        self.residual_weight = nn.Parameter(torch.ones(1) * 0.5)
        self.detail_weight = nn.Parameter(torch.ones(1) * 0.3)

    def _vit_forward(self, x, xray_features_2d, time_xray_cond):
        return self.vit_refiner(x=x, context=_tokens_of(xray_features_2d), cond=time_xray_cond, prev_stage_embed=None,
                                channels_last=True)

    def forward(self, volume_128, xray_features_2d, time_xray_cond):
        cdt = HF.compute_dtype(volume_128)
        x = HS.glue_sequential(self.upsample_from_128, _volume_channels_last(volume_128), cdt)
        # per token and block the branches keep ~48 C bytes (bf16 LN outputs, qkv / q / o / z, pre- and post-GELU, fp32 residuals)
        vit = self.vit_refiner
        saved = x.shape[0] * vit.pos_embed.shape[1] * len(vit.blocks) * 48 * vit.pos_embed.shape[2]
        if self.training and HF.use_checkpoint(self.use_gradient_checkpointing, saved, x.device):
            refinement = torch.utils.checkpoint.checkpoint(self._vit_forward, x, xray_features_2d, time_xray_cond,
                                                           use_reentrant=False)
        else:
            refinement = self._vit_forward(x, xray_features_2d, time_xray_cond)
        up = _resize(volume_128, self.volume_size)
        details = HS.glue_sequential(self.detail_enhancer, _volume_channels_last(up), cdt)         # (B,256,256,256,1)
        details = details.reshape(up.shape).float()
        return up + self.residual_weight * refinement + self.detail_weight * details


class ProgressiveCascadeModel(nn.Module):
    def __init__(self, xray_img_size=512, xray_feature_dim=512, voxel_dim=256, use_gradient_checkpointing=True):
        super().__init__()
        self.xray_encoder = MultiScaleXrayEncoder(img_size=xray_img_size, in_channels=1, base_dim=xray_feature_dim, num_views=2)
        self.stage1 = Stage1Base64(volume_size=(64, 64, 64), xray_img_size=xray_img_size, voxel_dim=voxel_dim, vit_depth=4,
                                   num_heads=4, xray_feature_dim=xray_feature_dim)
        self.stage2 = Stage2Refiner128(volume_size=(128, 128, 128), voxel_dim=voxel_dim, vit_depth=6, num_heads=8,
                                       xray_feature_dim=xray_feature_dim)
        self.stage3 = Stage3Refiner256(volume_size=(256, 256, 256), voxel_dim=voxel_dim, vit_depth=8, num_heads=8,
                                       xray_feature_dim=xray_feature_dim,
                                       use_gradient_checkpointing=use_gradient_checkpointing)

    def forward(self, xrays, return_intermediate=False, max_stage=3):
        outputs = {}
        volume_64 = self.stage1(xrays)
        outputs["stage1"] = volume_64
        if max_stage == 1:
            return outputs if return_intermediate else volume_64
        feats2, cond, _ = self.xray_encoder(xrays, stage=2)
        volume_128 = self.stage2(volume_64, feats2, cond)
        outputs["stage2"] = volume_128
        if max_stage == 2:
            return outputs if return_intermediate else volume_128
        feats3, cond, _ = self.xray_encoder(xrays, stage=3)
        volume_256 = self.stage3(volume_128, feats3, cond)
        outputs["stage3"] = volume_256
        return outputs if return_intermediate else volume_256

    def _set_stage_grad(self, stage, flag):
        mod = {1: self.stage1, 2: self.stage2, 3: self.stage3}.get(stage)
        if mod is None:
            return
        for p in mod.parameters():
            p.requires_grad = flag
        res = {1: "64³", 2: "128³", 3: "256³"}[stage]
        print(f"Stage {stage} ({res}) {'unfrozen' if flag else 'frozen'}")

    def freeze_stage(self, stage):
        self._set_stage_grad(stage, False)

    def unfreeze_stage(self, stage):
        self._set_stage_grad(stage, True)
