"""MI355X-native counterparts of the reference's models/hybrid_vit_backbone.py
(HybridViTBlock3D :21-143, HybridViT3D :146-274).

Each residual branch of a block is one fused HIP forward chain / backward chain
(hvc.functional.{SelfAttnBranchFn, CrossAttnBranchFn, MlpBranchFn}); the residual stream stays fp32.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from hvc import functional as HF
from hvc import stem as HS
from .vit_components import (AdaLNModulation, MultiHeadCrossAttention, MultiHeadSelfAttention,
                             SinusoidalTimeEmbedding)  # noqa: F401  (re-exported like the reference)


class HybridViTBlock3D(nn.Module):
    def __init__(self, voxel_dim: int, num_heads: int = 8, context_dim: int = 512, cond_dim: int = 1024,
                 mlp_ratio: int = 4, dropout: float = 0.1, use_prev_stage: bool = False,
                 return_attention: bool = False):
        super().__init__()
        self.voxel_dim = voxel_dim
        self.use_prev_stage = use_prev_stage
        self.return_attention = return_attention
        self.adaln = AdaLNModulation(embed_dim=voxel_dim, cond_dim=cond_dim + (256 if use_prev_stage else 0))
        self.self_attn = MultiHeadSelfAttention(embed_dim=voxel_dim, num_heads=num_heads, dropout=dropout)
        self.cross_attn = MultiHeadCrossAttention(embed_dim=voxel_dim, num_heads=num_heads, context_dim=context_dim,
                                                  dropout=dropout, store_attention=return_attention)
        hidden = int(voxel_dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(voxel_dim, hidden), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden, voxel_dim), nn.Dropout(dropout))
        self.norm1 = nn.LayerNorm(voxel_dim)
        self.norm2 = nn.LayerNorm(voxel_dim)
        self.norm3 = nn.LayerNorm(voxel_dim)

    def forward(self, voxel_features: torch.Tensor, xray_context: torch.Tensor, cond: torch.Tensor,
                prev_stage_embed: Optional[torch.Tensor] = None):
        x = voxel_features
        B = x.shape[0]
        if self.use_prev_stage:
            if prev_stage_embed is None:
                prev_stage_embed = torch.zeros(B, 256, device=x.device, dtype=cond.dtype)
            cond = torch.cat([cond, prev_stage_embed.to(cond.dtype)], dim=-1)
        shift_sa, scale_sa, gate_sa, shift_mlp, scale_mlp, gate_mlp = self.adaln(x, cond)
        cdt = HF.compute_dtype(x)
        sa, ca = self.self_attn, self.cross_attn
        training = self.training

        def p_of(mod):
            return mod.p if training else 0.0

        def seeds(p):
            return (HF.new_seed(), HF.new_seed()) if p > 0 else (0, 0)

        p = p_of(sa.attn_drop)
        x = HF.SelfAttnBranchFn.apply(x, self.norm1.weight, self.norm1.bias, scale_sa, shift_sa, gate_sa,
                                      sa.qkv.weight, sa.proj.weight, sa.proj.bias, sa.num_heads, cdt, p, seeds(p))
        if self.return_attention:
            with torch.no_grad():   # diagnostic side output only (reference :131-133)
                hn = HF.layer_norm(x.detach().float(), self.norm2.weight, self.norm2.bias, out_dtype=cdt)
                q = HF.linear(hn, ca.q.weight, None, cdt).view(B, -1, ca.num_heads, ca.head_dim)
                kv = HF.linear(xray_context, ca.kv.weight, None, cdt).view(B, -1, 2, ca.num_heads, ca.head_dim)
                ca._store_probs(q, kv)
        p = p_of(ca.attn_drop)
        x = HF.CrossAttnBranchFn.apply(x, xray_context, self.norm2.weight, self.norm2.bias, ca.q.weight, ca.kv.weight,
                                       ca.proj.weight, ca.proj.bias, ca.num_heads, cdt, p, seeds(p))
        p = p_of(self.mlp[2])
        x = HF.MlpBranchFn.apply(x, self.norm3.weight, self.norm3.bias, scale_mlp, shift_mlp, gate_mlp,
                                 self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias,
                                 cdt, p, seeds(p))
        if self.return_attention:
            return x, ca.attention_weights
        return x


def _stem_geometry(volume_size, token_grid):
    """Reference formula (models/hybrid_vit_backbone.py:174-188) for the downsample factor."""
    D, H, W = volume_size
    if token_grid is not None:
        target = token_grid
    elif D <= 64:
        target = 16
    elif D <= 128:
        target = 24
    else:
        target = 32
    return max(D // target, H // target, W // target, 1)


class HybridViT3D(nn.Module):
    """voxel-embed stem -> tokens (+pos_embed) -> blocks -> LN -> Linear(C,1) -> trilinear upsample.

    A2-fix: `downsampled_size` / `pos_embed` follow the grid the stem really
    emits (ceil(dim/2) per stride-2 conv).  This is byte-identical to the reference at 64^3 and
    256^3 and repairs its 128^3 shape error (25^3 pos_embed vs a 32^3 stem output).  `token_grid`
    (build-only kwarg) overrides the target grid, e.g. 16 reproduces the author's 128^3 variant."""

    def __init__(self, volume_size: Tuple[int, int, int] = (64, 64, 64), in_channels: int = 1, voxel_dim: int = 384,
                 depth: int = 6, num_heads: int = 6, context_dim: int = 512, cond_dim: int = 1024,
                 use_prev_stage: bool = False, dropout: float = 0.1, token_grid: Optional[int] = None):
        super().__init__()
        self.volume_size = tuple(volume_size)
        self.in_channels = in_channels
        self.voxel_dim = voxel_dim
        self.use_prev_stage = use_prev_stage
        remaining = _stem_geometry(self.volume_size, token_grid)
        layers, cur, grid = [], in_channels, list(self.volume_size)
        while remaining > 1:
            stride = min(remaining, 2)
            if cur == in_channels:
                out_dim = voxel_dim // 4
            elif len(layers) < 4:
                out_dim = voxel_dim // 2
            else:
                out_dim = voxel_dim
            layers += [nn.Conv3d(cur, out_dim, kernel_size=3, stride=stride, padding=1),
                       nn.GroupNorm(min(8, out_dim), out_dim), nn.SiLU()]
            grid = [(g - 1) // stride + 1 for g in grid]
            cur = out_dim
            remaining //= stride
        if cur != voxel_dim:
            layers.append(nn.Conv3d(cur, voxel_dim, kernel_size=3, padding=1))
        self.voxel_embed = nn.Sequential(*layers)
        self.downsampled_size = tuple(grid)
        n_tokens = grid[0] * grid[1] * grid[2]
        self.pos_embed = nn.Parameter(torch.randn(1, n_tokens, voxel_dim) * 0.02)
        self.blocks = nn.ModuleList([
            HybridViTBlock3D(voxel_dim=voxel_dim, num_heads=num_heads, context_dim=context_dim, cond_dim=cond_dim,
                             use_prev_stage=use_prev_stage, dropout=dropout) for _ in range(depth)])
        self.norm = nn.LayerNorm(voxel_dim)
        self.output_proj = nn.Linear(voxel_dim, 1)

    def forward(self, x: torch.Tensor, context: torch.Tensor, cond: torch.Tensor,
                prev_stage_embed: Optional[torch.Tensor] = None, channels_last: bool = False) -> torch.Tensor:
        """x: (B, Cin, D, H, W); with channels_last=True (build-only kwarg) x is (B, D, H, W, Cin), which
        spares the cascade stages a layout copy."""
        Dd, Hd, Wd = self.downsampled_size
        tokens = HS.voxel_tokens(self.voxel_embed, x, self.pos_embed, channels_last)   # (B, N, C) fp32, n = (d*H'+h)*W'+w
        for block in self.blocks:
            tokens = block(tokens, context, cond, prev_stage_embed)
        vol = HS.token_head(tokens, self.norm, self.output_proj, (Dd, Hd, Wd))  # (B,1,D',H',W') fp32
        return HS.upsample_trilinear(vol, self.volume_size)
