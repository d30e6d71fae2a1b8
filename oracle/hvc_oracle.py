"""CPU oracle for the Hybrid-ViT-Cascade hot path.  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32/fp64 restatement, in functional form (explicit parameter dicts keyed by the
reference's state_dict names), of the reference algorithm for every row of SURVEY.md §8(a).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (hybrid-vit-cascade_amd/) never does.

Pinning: every function below is checked against golden vectors produced by importing the
reference itself in the build container (tests/golden/*.npz, generator
tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.

Each function cites the reference lines (relative to the reference repo root) it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
import torch.utils.checkpoint

Params = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------------
# attention  (models/vit_components.py)
# ------------------------------------------------------------------------------------------------
CHUNK_CHECKPOINT = False      # attention_core: checkpoint every q_chunk slab (set by the 128^3 fixture generator only)


def dropout(x: torch.Tensor, p: float, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Dropout(p) in train mode (torch/nn/functional.py dropout: x * mask / (1 - p)).  keep = None draws torch's own mask as
    the reference does; a GIVEN keep mask (0/1, shape of x) is used by the dropout-on parity tests, which feed the oracle the
    mask the HIP kernels drew (recovered from kernel outputs: tests/test_gpu_dropout_parity.py) - same formula, that mask
    (tests/test_oracle_golden.py checks it reproduces F.dropout bit for bit on torch's own mask)."""
    if keep is None:
        return F.dropout(x, p, p > 0)
    return x * (keep.to(x.dtype) / (1.0 - p))       # ATen's order: the mask is divided by (1 - p), then multiplied in


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
                   q_chunk: Optional[int] = None, p_drop: float = 0.0, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v on (B, h, N, d) operands -- models/vit_components.py:46-51 and
    :103-113.  p_drop > 0 draws the reference's attn_drop mask on the probabilities (:48, :110; torch's
    generator, so the masks are torch's, as in the reference's train mode), or applies the given (B, h, Nq, Nk)
    keep mask (see dropout()).
    q_chunk bounds the materialised score slab so N = 32768 stays tractable on the CPU (identical
    arithmetic per row)."""
    if keep is not None:
        attn = (q @ k.transpose(-2, -1)) * scale
        return dropout(attn.softmax(dim=-1), p_drop, keep) @ v
    if q_chunk is None or q.shape[-2] <= q_chunk:
        attn = (q @ k.transpose(-2, -1)) * scale
        return F.dropout(attn.softmax(dim=-1), p_drop, p_drop > 0) @ v

    def rows(qs, k_, v_):
        return F.dropout(((qs @ k_.transpose(-2, -1)) * scale).softmax(dim=-1), p_drop, p_drop > 0) @ v_
    outs = []
    for s in range(0, q.shape[-2], q_chunk):
        qs = q[..., s:s + q_chunk, :]
        if CHUNK_CHECKPOINT and p_drop == 0 and torch.is_grad_enabled():
            # same arithmetic, but the (chunk x N) probabilities are rebuilt in the backward pass instead of being kept:
            # a fp32 backward at N = 32768 (128^3) then fits in tens of GB (tests/golden/make_direct128_probes.py)
            outs.append(torch.utils.checkpoint.checkpoint(rows, qs, k, v, use_reentrant=False))
        else:
            outs.append(rows(qs, k, v))
    return torch.cat(outs, dim=-2)


def self_attention(x: torch.Tensor, P: Params, pre: str, num_heads: int,
                   q_chunk: Optional[int] = None, p_drop: float = 0.0, keeps=(None, None)) -> torch.Tensor:
    """MultiHeadSelfAttention.forward, models/vit_components.py:31-57 (p_drop = 0: eval; > 0: attn_drop :48
    and proj_drop :55).  qkv columns are laid out [q(h,d) | k(h,d) | v(h,d)] (:41-43).
    keeps = (attention-probability keep mask, proj-output keep mask) or Nones (torch's masks)."""
    B, N, Cn = x.shape
    d = Cn // num_heads
    qkv = F.linear(x, P[pre + "qkv.weight"]).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    o = attention_core(qkv[0], qkv[1], qkv[2], d ** -0.5, q_chunk, p_drop, keeps[0])
    o = o.transpose(1, 2).reshape(B, N, Cn)
    return dropout(F.linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"]), p_drop, keeps[1])


def cross_attention(x: torch.Tensor, ctx: torch.Tensor, P: Params, pre: str, num_heads: int,
                    q_chunk: Optional[int] = None, p_drop: float = 0.0, keeps=(None, None)) -> torch.Tensor:
    """MultiHeadCrossAttention.forward, models/vit_components.py:84-119 (p_drop / keeps as in self_attention: :110, :117)."""
    B, N, Cn = x.shape
    M = ctx.shape[1]
    d = Cn // num_heads
    q = F.linear(x, P[pre + "q.weight"]).reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    kv = F.linear(ctx, P[pre + "kv.weight"]).reshape(B, M, 2, num_heads, d).permute(2, 0, 3, 1, 4)
    o = attention_core(q, kv[0], kv[1], d ** -0.5, q_chunk, p_drop, keeps[0])
    o = o.transpose(1, 2).reshape(B, N, Cn)
    return dropout(F.linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"]), p_drop, keeps[1])


def adaln_params(cond: torch.Tensor, P: Params, pre: str) -> Tuple[torch.Tensor, ...]:
    """AdaLNModulation.forward, models/vit_components.py:135-149: Linear(cond_dim, 6C),
    unsqueeze(1), chunk order shift_sa, scale_sa, gate_sa, shift_mlp, scale_mlp, gate_mlp."""
    p = F.linear(cond, P[pre + "linear.weight"], P[pre + "linear.bias"]).unsqueeze(1)
    return tuple(p.chunk(6, dim=-1))


def sinusoidal_time_embedding(t: torch.Tensor, embed_dim: int) -> torch.Tensor:
    """SinusoidalTimeEmbedding.forward, models/vit_components.py:161-174."""
    half = embed_dim // 2
    freq = torch.exp(torch.arange(half, device=t.device) * -(math.log(10000) / (half - 1)))
    e = t[:, None] * freq[None, :]
    return torch.cat([e.sin(), e.cos()], dim=-1)


# ------------------------------------------------------------------------------------------------
# ViT block / backbone  (models/hybrid_vit_backbone.py)
# ------------------------------------------------------------------------------------------------
def vit_block(x: torch.Tensor, ctx: torch.Tensor, cond: torch.Tensor, P: Params, pre: str,
              num_heads: int, q_chunk: Optional[int] = None, p_drop: float = 0.0, keeps: Optional[dict] = None) -> torch.Tensor:
    """HybridViTBlock3D.forward with use_prev_stage=False, return_attention=False,
    models/hybrid_vit_backbone.py:88-143 (p_drop > 0: the train-mode nn.Dropout draws of :77, :79 and of the
    attention modules).  keeps: optional given keep masks {"sa_attn", "sa_proj", "ca_attn", "ca_proj", "fc1", "fc2"}
    in place of torch's draws (see dropout())."""
    Cn = x.shape[-1]
    kp = keeps or {}
    shift_sa, scale_sa, gate_sa, shift_mlp, scale_mlp, gate_mlp = adaln_params(cond, P, pre + "adaln.")
    h = F.layer_norm(x, (Cn,), P[pre + "norm1.weight"], P[pre + "norm1.bias"], 1e-5)          # :120
    h = (1 + scale_sa) * h + shift_sa                                                          # :121
    x = x + gate_sa * self_attention(h, P, pre + "self_attn.", num_heads, q_chunk, p_drop,
                                     (kp.get("sa_attn"), kp.get("sa_proj")))                   # :122-123
    h = F.layer_norm(x, (Cn,), P[pre + "norm2.weight"], P[pre + "norm2.bias"], 1e-5)          # :126
    x = x + cross_attention(h, ctx, P, pre + "cross_attn.", num_heads, q_chunk, p_drop,
                            (kp.get("ca_attn"), kp.get("ca_proj")))                             # :127-128
    h = F.layer_norm(x, (Cn,), P[pre + "norm3.weight"], P[pre + "norm3.bias"], 1e-5)          # :136
    h = (1 + scale_mlp) * h + shift_mlp                                                        # :137
    h = F.linear(h, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"])                            # :75
    h = dropout(F.gelu(h), p_drop, kp.get("fc1"))                                              # :76 exact erf, :77
    h = F.linear(h, P[pre + "mlp.3.weight"], P[pre + "mlp.3.bias"])                            # :78
    return x + gate_mlp * dropout(h, p_drop, kp.get("fc2"))                                    # :79, :139


def voxel_embed_plan(volume_size: Sequence[int], in_channels: int, voxel_dim: int,
                     token_grid: Optional[int] = None):
    """Geometry of HybridViT3D.__init__, models/hybrid_vit_backbone.py:174-210.

    Returns (layers, ref_downsampled_size, actual_grid) where layers is a list of
    ('conv', cin, cout, stride) | ('gn', groups, ch) | ('silu',) in nn.Sequential index order,
    ref_downsampled_size is the reference's formula (:186) and actual_grid is what the stem
    really emits (ceil(dim/2) per stride-2 conv, k3 p1).  They differ only for 64 < D <= 128
    (SURVEY.md §0.4); the build uses actual_grid (row A2-fix)."""
    D, H, W = volume_size
    if token_grid is not None:
        target = token_grid
    elif D <= 64:
        target = 16
    elif D <= 128:
        target = 24
    else:
        target = 32
    factor = max(D // target, H // target, W // target, 1)
    ref_ds = tuple(d // factor for d in volume_size)
    layers: List[tuple] = []
    cur, rem = in_channels, factor
    grid = list(volume_size)
    while rem > 1:
        stride = min(rem, 2)
        if cur == in_channels:
            out = voxel_dim // 4
        elif len(layers) < 4:
            out = voxel_dim // 2
        else:
            out = voxel_dim
        layers += [("conv", cur, out, stride), ("gn", min(8, out), out), ("silu",)]
        grid = [(g + 2 - 3) // stride + 1 for g in grid]
        cur = out
        rem //= stride
    if cur != voxel_dim:
        layers.append(("conv", cur, voxel_dim, 1))
    return layers, ref_ds, tuple(grid)


def voxel_embed(x: torch.Tensor, P: Params, pre: str, layers) -> torch.Tensor:
    """self.voxel_embed(x), models/hybrid_vit_backbone.py:195-210, :252."""
    for i, spec in enumerate(layers):
        if spec[0] == "conv":
            x = F.conv3d(x, P[f"{pre}{i}.weight"], P[f"{pre}{i}.bias"], stride=spec[3], padding=1)
        elif spec[0] == "gn":
            x = F.group_norm(x, spec[1], P[f"{pre}{i}.weight"], P[f"{pre}{i}.bias"], 1e-5)
        else:
            x = F.silu(x)
    return x


def hybrid_vit3d(x: torch.Tensor, ctx: torch.Tensor, cond: torch.Tensor, P: Params, pre: str,
                 volume_size: Sequence[int], in_channels: int, voxel_dim: int, depth: int,
                 num_heads: int, token_grid: Optional[int] = None,
                 q_chunk: Optional[int] = None, p_drop: float = 0.0) -> torch.Tensor:
    """HybridViT3D.forward, models/hybrid_vit_backbone.py:233-274 (token n = (d*H'+h)*W'+w, :255)."""
    layers, _, grid = voxel_embed_plan(volume_size, in_channels, voxel_dim, token_grid)
    B = x.shape[0]
    h = voxel_embed(x, P, pre + "voxel_embed.", layers)
    assert tuple(h.shape[2:]) == grid
    h = h.flatten(2).transpose(1, 2) + P[pre + "pos_embed"]                                    # :255-258
    for i in range(depth):
        h = vit_block(h, ctx, cond, P, f"{pre}blocks.{i}.", num_heads, q_chunk, p_drop)        # :261-262
    h = F.layer_norm(h, (voxel_dim,), P[pre + "norm.weight"], P[pre + "norm.bias"], 1e-5)     # :265
    h = F.linear(h, P[pre + "output_proj.weight"], P[pre + "output_proj.bias"])               # :266
    h = h.transpose(1, 2).reshape(B, 1, *grid)                                                 # :269
    return F.interpolate(h, size=tuple(volume_size), mode="trilinear", align_corners=True)     # :272


# ------------------------------------------------------------------------------------------------
# X-ray conditioning stem  (models/diagnostic_losses.py:68-138)
# ------------------------------------------------------------------------------------------------
def _bn(x, P, pre, training, new_stats):
    """nn.BatchNorm2d (momentum 0.1, eps 1e-5).  training=True uses batch statistics and records
    the updated running stats (unbiased variance) in new_stats, as torch does in place."""
    rm, rv = P[pre + "running_mean"], P[pre + "running_var"]
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, P[pre + "weight"], P[pre + "bias"], True, 0.1, 1e-5)
        if new_stats is not None:
            new_stats[pre + "running_mean"], new_stats[pre + "running_var"] = rm2, rv2
        return y
    return F.batch_norm(x, rm, rv, P[pre + "weight"], P[pre + "bias"], False, 0.1, 1e-5)


def _relu_pool(b: torch.Tensor, pool, tag: str, route: Optional[dict]) -> torch.Tensor:
    """ReLU (+ MaxPool2d(k, s, p)) with the ROUTING made explicit.  route = None: plain F.relu / F.max_pool2d.
    Otherwise route["own"][tag] records this evaluation's routing -- relu mask (b > 0) and, per pooled window, the
    flat h*W+w index of its arg-max -- and, when route["use"] holds an entry for tag, the output is formed with THAT
    routing instead (mask multiply + gather): the same function wherever the two routings agree, and a smooth function
    of its inputs for a fixed routing, which is what a gradient comparison across implementations needs (a near-tie
    that rounds the other way re-routes a whole window's gradient)."""
    if route is None:
        r = F.relu(b)
        return r if pool is None else F.max_pool2d(r, *pool)
    own = route.setdefault("own", {})
    mask = b > 0
    rec = {"relu": mask}
    r = F.relu(b)
    if pool is not None:
        pooled, idx = F.max_pool2d(r, *pool, return_indices=True)
        rec["argmax"], rec["max"] = idx, pooled.detach()
    own[tag] = rec
    use = route.get("use", {}).get(tag)
    if use is None:
        return r if pool is None else pooled
    r = b * use["relu"].to(b.dtype)
    if pool is None:
        return r
    idx = use["argmax"]
    return r.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)


def xray_encoder(x: torch.Tensor, P: Params, pre: str, training: bool = False,
                 new_stats: Optional[Params] = None, route: Optional[dict] = None) -> torch.Tensor:
    """self.encoder, models/diagnostic_losses.py:82-96: Conv7x7 s2 p3 -> BN -> ReLU -> MaxPool(3,2,1)
    -> Conv3x3 p1 -> BN -> ReLU -> MaxPool(2,2) -> Conv3x3 p1 -> BN -> ReLU.  route: see _relu_pool."""
    x = F.conv2d(x, P[pre + "0.weight"], P[pre + "0.bias"], stride=2, padding=3)
    x = _relu_pool(_bn(x, P, pre + "1.", training, new_stats), (3, 2, 1), "1", route)
    x = F.conv2d(x, P[pre + "4.weight"], P[pre + "4.bias"], padding=1)
    x = _relu_pool(_bn(x, P, pre + "5.", training, new_stats), (2, 2, 0), "5", route)
    x = F.conv2d(x, P[pre + "8.weight"], P[pre + "8.bias"], padding=1)
    return _relu_pool(_bn(x, P, pre + "9.", training, new_stats), None, "9", route)


def xray_conditioning(xrays: torch.Tensor, t: torch.Tensor, P: Params, pre: str,
                      training: bool = False, new_stats: Optional[Params] = None, route: Optional[dict] = None):
    """XrayConditioningModule.forward, models/diagnostic_losses.py:108-138.
    Returns (xray_context (B,cond), time_xray_cond (B,cond), features (B,E,H',W'))."""
    B, V = xrays.shape[0], xrays.shape[1]
    if V > 1:
        f = xray_encoder(xrays.reshape(B * V, *xrays.shape[2:]), P, pre + "encoder.", training, new_stats, route)  # :123-124
        f = f.view(B, V, *f.shape[1:]).mean(dim=1)                                                          # :126
    else:
        f = xray_encoder(xrays[:, 0], P, pre + "encoder.", training, new_stats, route)                      # :128
    xc = F.linear(f.mean(dim=[-2, -1]), P[pre + "to_cond.weight"], P[pre + "to_cond.bias"])                # :131-132
    te = F.linear(t, P[pre + "time_mlp.0.weight"], P[pre + "time_mlp.0.bias"])                             # :99-103
    te = F.linear(F.silu(te), P[pre + "time_mlp.2.weight"], P[pre + "time_mlp.2.bias"])
    return xc, te + xc, f                                                                                   # :135-138


# ------------------------------------------------------------------------------------------------
# DRR ops
# ------------------------------------------------------------------------------------------------
def drr_render(volume: torch.Tensor, angle: float = 0) -> torch.Tensor:
    """DRRRenderer.forward, models/diagnostic_losses.py:31-65. volume (B,D,H,W)."""
    att = torch.exp(-0.3 * (volume + 1.0))                     # :45-51
    if angle == 90:
        drr = att.sum(dim=-1).transpose(1, 2)                  # :53-56  (B,H,D)
    else:
        drr = att.sum(dim=1)                                   # :58-59  (B,H,W)
    return torch.clamp(drr, min=1e-6)                          # :63


def projection_loss(volume: torch.Tensor, xray_target: torch.Tensor, angle: float = 0) -> torch.Tensor:
    """ProjectionLoss.forward, models/diagnostic_losses.py:149-169. volume (B,1,D,H,W), target (B,1,h,w)."""
    drr = drr_render(volume.squeeze(1), angle)
    if drr.shape != xray_target.squeeze(1).shape:
        drr = F.interpolate(drr.unsqueeze(1), size=xray_target.shape[2:], mode="bilinear",
                            align_corners=True).squeeze(1)     # :161-165
    return F.mse_loss(drr, xray_target.squeeze(1))             # :167-169


def mean_projection(ct: torch.Tensor, view_angle: int, img_size: int) -> torch.Tensor:
    """DRRReprojectionLoss.generate_drr, direct_regression/progressive_cascade/loss_multiscale.py:250-273."""
    drr = ct.mean(dim=2) if view_angle == 0 else ct.mean(dim=4)
    return F.interpolate(drr, size=(img_size, img_size), mode="bilinear", align_corners=False)


def drr_reprojection_loss(pred: torch.Tensor, xrays: torch.Tensor, img_size: int = 512) -> torch.Tensor:
    """DRRReprojectionLoss.forward, loss_multiscale.py:275-293."""
    ap = mean_projection(pred, 0, img_size)
    lat = mean_projection(pred, 90, img_size)
    return (F.l1_loss(ap, xrays[:, 0]) + F.l1_loss(lat, xrays[:, 1])) / 2


# ------------------------------------------------------------------------------------------------
# Direct regression model + loss + metric
# ------------------------------------------------------------------------------------------------
def direct_ct_regression(xrays: torch.Tensor, P: Params, volume_size=(64, 64, 64), voxel_dim=256,
                         vit_depth=4, num_heads=4, training: bool = False,
                         new_stats: Optional[Params] = None, token_grid: Optional[int] = None,
                         q_chunk: Optional[int] = None, p_drop: float = 0.0, route: Optional[dict] = None) -> torch.Tensor:
    """DirectCTRegression.forward, direct_regression/model_direct.py:59-85 (p_drop = 0.1 with training=True is the
    reference's train mode: hybrid_vit_backbone.py:38, :166 hard-code dropout 0.1)."""
    B = xrays.shape[0]
    t = torch.zeros(B, 256, dtype=xrays.dtype, device=xrays.device)                            # :69
    _, cond, feats = xray_conditioning(xrays, t, P, "xray_encoder.", training, new_stats, route)   # :72 (route: see _relu_pool)
    x = P["initial_volume"].expand(B, -1, -1, -1, -1)                                          # :75
    ctx = feats.flatten(2).transpose(1, 2)                                                     # :80
    return hybrid_vit3d(x, ctx, cond, P, "vit_backbone.", volume_size, 1, voxel_dim, vit_depth,
                        num_heads, token_grid, q_chunk, p_drop)


def ssim_loss_3d(pred: torch.Tensor, target: torch.Tensor, window: int = 11) -> torch.Tensor:
    """compute_ssim_loss, direct_regression/model_direct.py:88-107."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    pool = lambda z: F.avg_pool3d(z, window, stride=1, padding=window // 2)
    mp, mt = pool(pred), pool(target)
    sp = pool(pred * pred) - mp * mp
    st = pool(target * target) - mt * mt
    spt = pool(pred * target) - mp * mt
    ssim = ((2 * mp * mt + C1) * (2 * spt + C2)) / ((mp * mp + mt * mt + C1) * (sp + st + C2))
    return 1 - ssim.mean()


def direct_regression_loss(pred, target, l1_weight=1.0, ssim_weight=0.5):
    """DirectRegressionLoss.forward, direct_regression/model_direct.py:118-131."""
    l1 = F.l1_loss(pred, target)
    ss = ssim_loss_3d(pred, target)
    return {"total_loss": l1_weight * l1 + ssim_weight * ss, "l1_loss": l1, "ssim_loss": ss}


def total_variation_loss(pred: torch.Tensor, target: torch.Tensor = None, eps: float = 1e-8) -> torch.Tensor:
    """TotalVariationLoss.forward, direct_regression/progressive_cascade/loss_multiscale.py:150-188: mean over the three
    axes of mean sqrt(forward-difference^2 + eps), clamped to [0, 100]; with a target, |tv(pred) - tv(target)|."""
    def tv(v):
        v = v.float()
        terms = [torch.sqrt((v[:, :, 1:] - v[:, :, :-1]).abs().pow(2) + eps).mean(),
                 torch.sqrt((v[:, :, :, 1:] - v[:, :, :, :-1]).abs().pow(2) + eps).mean(),
                 torch.sqrt((v[..., 1:] - v[..., :-1]).abs().pow(2) + eps).mean()]
        return torch.clamp(sum(terms) / 3, 0, 100)
    return tv(pred) if target is None else F.l1_loss(tv(pred), tv(target))


def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """compute_psnr, direct_regression/train_direct_4gpu.py:40-46 (data range 2)."""
    mse = torch.mean((pred - target) ** 2)
    if mse == 0:
        return float("inf")
    return (20 * torch.log10(2.0 / torch.sqrt(mse))).item()


# ------------------------------------------------------------------------------------------------
# Progressive cascade glue  (direct_regression/progressive_cascade/model_progressive.py)
# ------------------------------------------------------------------------------------------------
def _conv_gn_gelu_2d(x, P, pre, idx, stride):
    x = F.conv2d(x, P[f"{pre}{idx}.weight"], P[f"{pre}{idx}.bias"], stride=stride, padding=1)
    x = F.group_norm(x, 32, P[f"{pre}{idx + 1}.weight"], P[f"{pre}{idx + 1}.bias"], 1e-5)
    return F.gelu(x)


def multiscale_xray_encoder(xrays: torch.Tensor, P: Params, pre: str, stage: int, training: bool = False):
    """MultiScaleXrayEncoder.forward, model_progressive.py:54-83.  Returns (features_2d, time_xray_cond, xray_context)."""
    B = xrays.shape[0]
    t = torch.zeros(B, 256, dtype=xrays.dtype)
    ctx, cond, f = xray_conditioning(xrays, t, P, pre + "xray_encoder.", training)
    if stage == 1:                                                    # :74-76  two stride-2 conv+GN(32)+GELU
        f = _conv_gn_gelu_2d(f, P, pre + "to_stage1.", 0, 2)
        f = _conv_gn_gelu_2d(f, P, pre + "to_stage1.", 3, 2)
    elif stage == 2:                                                  # :77-79
        f = _conv_gn_gelu_2d(f, P, pre + "to_stage2.", 0, 2)
    return f, cond, ctx


def _upsample_stem(v, P, pre):
    """nn.Upsample(x2, trilinear, align_corners=False) -> Conv3d(1,32,3,p1) -> GroupNorm(8,32) -> GELU (:169-174, :238-243)."""
    x = F.interpolate(v, scale_factor=2, mode="trilinear", align_corners=False)
    x = F.conv3d(x, P[pre + "1.weight"], P[pre + "1.bias"], padding=1)
    return F.gelu(F.group_norm(x, 8, P[pre + "2.weight"], P[pre + "2.bias"], 1e-5))


def stage2_refiner(v64, feats, cond, P: Params, pre: str, volume_size, voxel_dim, depth, heads, token_grid=None):
    """Stage2Refiner128.forward, model_progressive.py:191-216."""
    x = _upsample_stem(v64, P, pre + "upsample_from_64.")
    ref = hybrid_vit3d(x, feats.flatten(2).transpose(1, 2), cond, P, pre + "vit_refiner.", volume_size, 32, voxel_dim,
                       depth, heads, token_grid)
    up = F.interpolate(v64, size=tuple(volume_size), mode="trilinear", align_corners=False)
    return up + P[pre + "residual_weight"] * ref


def stage3_refiner(v128, feats, cond, P: Params, pre: str, volume_size, voxel_dim, depth, heads):
    """Stage3Refiner256.forward (no checkpointing), model_progressive.py:273-307."""
    x = _upsample_stem(v128, P, pre + "upsample_from_128.")
    ref = hybrid_vit3d(x, feats.flatten(2).transpose(1, 2), cond, P, pre + "vit_refiner.", volume_size, 32, voxel_dim,
                       depth, heads)
    up = F.interpolate(v128, size=tuple(volume_size), mode="trilinear", align_corners=False)
    d = F.conv3d(up, P[pre + "detail_enhancer.0.weight"], P[pre + "detail_enhancer.0.bias"], padding=1)       # :259-267
    d = F.gelu(F.group_norm(d, 16, P[pre + "detail_enhancer.1.weight"], P[pre + "detail_enhancer.1.bias"], 1e-5))
    d = F.conv3d(d, P[pre + "detail_enhancer.3.weight"], P[pre + "detail_enhancer.3.bias"], padding=1)
    d = F.gelu(F.group_norm(d, 8, P[pre + "detail_enhancer.4.weight"], P[pre + "detail_enhancer.4.bias"], 1e-5))
    d = F.conv3d(d, P[pre + "detail_enhancer.6.weight"], P[pre + "detail_enhancer.6.bias"])
    return up + P[pre + "residual_weight"] * ref + P[pre + "detail_weight"] * d                                 # :303-305
