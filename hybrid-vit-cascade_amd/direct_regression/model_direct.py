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


def compute_ssim_loss(pred, target, window_size=11):
    """1 - mean SSIM with an 11^3 box window (reference :88-107)."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    pad = window_size // 2

    def box(z):
        return F.avg_pool3d(z, window_size, stride=1, padding=pad)

    mu_p, mu_t = box(pred), box(target)
    var_p = box(pred * pred) - mu_p * mu_p
    var_t = box(target * target) - mu_t * mu_t
    cov = box(pred * target) - mu_p * mu_t
    ssim = ((2 * mu_p * mu_t + C1) * (2 * cov + C2)) / ((mu_p * mu_p + mu_t * mu_t + C1) * (var_p + var_t + C2))
    return 1 - ssim.mean()


class DirectRegressionLoss(nn.Module):
    def __init__(self, l1_weight=1.0, ssim_weight=0.5):
        super().__init__()
        self.l1_weight = l1_weight
        self.ssim_weight = ssim_weight

    def forward(self, pred, target):
        l1 = F.l1_loss(pred, target)
        ssim = compute_ssim_loss(pred, target)
        return {"total_loss": self.l1_weight * l1 + self.ssim_weight * ssim, "l1_loss": l1, "ssim_loss": ssim}
