"""MI355X-native counterparts of the hot-path classes of the reference's models/diagnostic_losses.py:
DRRRenderer (:22-65), XrayConditioningModule (:68-138), ProjectionLoss (:141-169)."""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from hvc import functional as HF
from hvc import stem as HS


class DRRRenderer(nn.Module):
    """Beer-Lambert ray sum: exp(-0.3 (v + 1)) summed along D (frontal) or W (lateral, transposed to
    (B,H,D)), clamped at 1e-6.  HIP ray-sum kernel (wavefront reductions along W)."""

    def __init__(self, volume_shape: Tuple[int, int, int]):
        super().__init__()
        self.volume_shape = volume_shape

    def forward(self, volume: torch.Tensor, angle: float = 0) -> torch.Tensor:
        if angle == 90:
            return HF.drr_project(volume, 2, exp_mode=True, mu=0.3, clamp_min=1e-6, transpose_out=True)
        return HF.drr_project(volume, 0, exp_mode=True, mu=0.3, clamp_min=1e-6)


class XrayConditioningModule(nn.Module):
    """2-D CNN X-ray stem + global conditioning vector.  Same children / state_dict keys as the
    reference (encoder.{0,1,4,5,8,9}, time_mlp.{0,2}, to_cond).  `img_size` and
    `share_view_weights` are accepted and unused, as in the reference."""

    def __init__(self, img_size: int = 512, in_channels: int = 1, embed_dim: int = 256, num_views: int = 1,
                 time_embed_dim: int = 256, cond_dim: int = 1024, share_view_weights: bool = True):
        super().__init__()
        self.num_views = num_views
        self.embed_dim = embed_dim
        self.cond_dim = cond_dim
        self.encoder = nn.Sequential(
            nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            nn.Conv2d(64, 128, kernel_size=3, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Conv2d(128, embed_dim, kernel_size=3, padding=1), nn.BatchNorm2d(embed_dim), nn.ReLU(inplace=True),
        )
        self.time_mlp = nn.Sequential(nn.Linear(time_embed_dim, time_embed_dim * 2), nn.SiLU(),
                                      nn.Linear(time_embed_dim * 2, cond_dim))
        self.to_cond = nn.Linear(embed_dim, cond_dim)

    def forward(self, xrays: torch.Tensor, t: torch.Tensor):
        B, V = xrays.shape[0], xrays.shape[1]
        # channels-last feature map (B*V, H', W', E) straight out of the HIP stem
        # view mean (reference :126) and global average pool (:131) in one fused pass over the stem's output
        enc = HS.xray_encoder(self.encoder, xrays.reshape(B * V, *xrays.shape[2:]) if V > 1 else xrays[:, 0])
        hh, ww, E = enc.shape[1:]
        f, pooled = HF.ViewMeanGapFn.apply(enc.reshape(B * V if V > 1 else B, hh * ww, E), V if V > 1 else 1)
        f = f.view(B, hh, ww, E)
        feats = f.permute(0, 3, 1, 2)     # (B, E, H', W') view; .flatten(2).transpose(1, 2) of it is copy-free
        f32 = torch.float32
        xray_context = HF.linear(pooled, self.to_cond.weight, self.to_cond.bias, f32, f32)
        te = HF.linear(t.float(), self.time_mlp[0].weight, self.time_mlp[0].bias, f32, f32)
        te = HF.linear(F.silu(te), self.time_mlp[2].weight, self.time_mlp[2].bias, f32, f32)
        return xray_context, te + xray_context, feats


class ProjectionLoss(nn.Module):
    """MSE between the rendered DRR (bilinear align_corners=True resize when shapes differ) and the
    target X-ray (reference :149-169)."""

    def __init__(self, volume_shape: Tuple[int, int, int]):
        super().__init__()
        self.drr_renderer = DRRRenderer(volume_shape)

    def forward(self, volume: torch.Tensor, xray_target: torch.Tensor, angle: float = 0) -> torch.Tensor:
        drr = self.drr_renderer(volume.squeeze(1), angle=angle)
        # bilinear (align_corners=True) resize to the X-ray raster + MSE, fused: the resized image is never written
        # (a same-size "resize" is the identity at align_corners=True, so one path serves both branches of the reference)
        return HF.ResizeLossFn.apply(drr, xray_target.squeeze(1), True, 1)
