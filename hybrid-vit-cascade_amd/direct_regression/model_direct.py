"""MI355X-native counterpart of the reference's direct_regression/model_direct.py:
DirectCTRegression (:15-85), compute_ssim_loss (:88-107), DirectRegressionLoss (:110-131)."""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:          # same cwd-relative trick as the reference (model_direct.py:9)
    sys.path.insert(0, _PKG)

from hvc import functional as HF  # noqa: E402
from models.diagnostic_losses import XrayConditioningModule  # noqa: E402
from models.hybrid_vit_backbone import HybridViT3D  # noqa: E402


class DirectCTRegression(nn.Module):
    """X-rays (B,2,1,H,W) -> CT volume (B,1,D,H,W); constructor kwargs = config['model'] keys."""

    def __init__(self, volume_size=(64, 64, 64), xray_img_size=512, voxel_dim=256, vit_depth=4, num_heads=4,
                 xray_feature_dim=512):
        super().__init__()
        self.volume_size = tuple(volume_size)
        self.xray_encoder = XrayConditioningModule(img_size=xray_img_size, in_channels=1, embed_dim=xray_feature_dim,
                                                   num_views=2, time_embed_dim=256, cond_dim=1024,
                                                   share_view_weights=False)
        self.vit_backbone = HybridViT3D(volume_size=self.volume_size, in_channels=1, voxel_dim=voxel_dim, depth=vit_depth,
                                        num_heads=num_heads, context_dim=xray_feature_dim, cond_dim=1024,
                                        use_prev_stage=False)
        D, H, W = self.volume_size
        self.initial_volume = nn.Parameter(torch.randn(1, 1, D, H, W) * 0.01)

    def forward(self, xrays):
        B = xrays.shape[0]
        dummy_t = torch.zeros(B, 256, device=xrays.device)
        _, cond, feats = self.xray_encoder(xrays, dummy_t)
        x = self.initial_volume.expand(B, -1, -1, -1, -1)
        return self.vit_backbone(x=x, context=feats.flatten(2).transpose(1, 2), cond=cond, prev_stage_embed=None)


def _check_window_fits(pred, window_size):
    # F.avg_pool3d (reference :92) rejects an input extent below the kernel size even when the padding would cover it
    if min(pred.shape[-3:]) < window_size:
        raise RuntimeError(f"input image (T: {pred.shape[-3]} H: {pred.shape[-2]} W: {pred.shape[-1]}) smaller than kernel size "
                           f"(kT: {window_size} kH: {window_size} kW: {window_size})")


def compute_ssim_loss(pred, target, window_size=11):
    """1 - mean SSIM with a window_size^3 zero-padded box window (reference :88-107), fused HIP kernel."""
    if not pred.is_cuda:
        raise RuntimeError("compute_ssim_loss runs on the MI355X HIP path only (no CPU fallback)")
    _check_window_fits(pred, int(window_size))
    return HF.SsimL1LossFn.apply(pred, target, 0.0, 1.0, int(window_size))[2]


class DirectRegressionLoss(nn.Module):
    def __init__(self, l1_weight=1.0, ssim_weight=0.5):
        super().__init__()
        self.l1_weight = l1_weight
        self.ssim_weight = ssim_weight

    def forward(self, pred, target):
        if not pred.is_cuda:
            raise RuntimeError("DirectRegressionLoss runs on the MI355X HIP path only (no CPU fallback)")
        _check_window_fits(pred, 11)
        out = HF.SsimL1LossFn.apply(pred, target, float(self.l1_weight), float(self.ssim_weight), 11)
        return {"total_loss": out[0], "l1_loss": out[1], "ssim_loss": out[2]}
