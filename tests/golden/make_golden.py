"""Generates the golden vectors under tests/golden/ by importing the REFERENCE implementation.

Run only in the build container, where the reference is mounted read-only:
    HVC_REFERENCE=/root/reference python tests/golden/make_golden.py
The reference never travels to the GPU box; only the .npz files written here do.  Each file holds
inputs, the reference's parameters (state_dict), its outputs and its gradients, all fp32, from
seeded generators.  AdaLN linears are re-initialised ~N(0, 0.02^2) before capture because the
reference zero-initialises them (models/vit_components.py:132-133), which would hide the
self-attention and MLP branches; dropout is disabled (eval(), or p=0 in train mode) because its
mask is not reproducible outside the reference's RNG stream.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("HVC_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    sys.exit(f"reference not found at {REF}; golden vectors can only be regenerated in the build container")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "direct_regression"))
sys.path.insert(0, os.path.join(REF, "direct_regression", "progressive_cascade"))

from models.vit_components import MultiHeadSelfAttention, MultiHeadCrossAttention, AdaLNModulation, SinusoidalTimeEmbedding  # noqa: E402
from models.hybrid_vit_backbone import HybridViTBlock3D, HybridViT3D  # noqa: E402
from models.diagnostic_losses import XrayConditioningModule, DRRRenderer, ProjectionLoss  # noqa: E402
from model_direct import DirectCTRegression, DirectRegressionLoss  # noqa: E402
from loss_multiscale import DRRReprojectionLoss, TotalVariationLoss, FrequencyLoss, compute_psnr  # noqa: E402
from model_progressive import MultiScaleXrayEncoder, Stage2Refiner128, Stage3Refiner256  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def randn(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def reinit_adaln(module, g):
    for name, m in module.named_modules():
        if isinstance(m, AdaLNModulation):
            with torch.no_grad():
                m.linear.weight.copy_(randn(g, *m.linear.weight.shape, scale=0.02))
                m.linear.bias.copy_(randn(g, *m.linear.bias.shape, scale=0.02))


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


PROBE_LIMIT = 32768   # gradient / state arrays larger than this are stored as probes + sums
PROBE_N = 512


def probe_indices(n):
    """Fixed pseudo-random flat indices used to sample large arrays (shared with the tests)."""
    return np.random.default_rng(12345).integers(0, n, size=PROBE_N)


def save(name, arrays, compact=("pgrad", "stats_after")):
    flat = {}

    def put(key, val, may_compact):
        a = val.detach().cpu().numpy() if torch.is_tensor(val) else np.asarray(val)
        if may_compact and a.size > PROBE_LIMIT:
            f = a.reshape(-1).astype(np.float64)
            flat[key + "#probe"] = a.reshape(-1)[probe_indices(a.size)]
            flat[key + "#stats"] = np.array([f.sum(), np.abs(f).sum(), a.size])
        else:
            flat[key] = a

    for k, v in arrays.items():
        mc = any(c in k for c in compact)
        if isinstance(v, dict):
            for kk, vv in v.items():
                put(f"{k}/{kk}", vv, mc)
        else:
            put(k, v, mc)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(flat)} arrays")


def grads_of(module, out, weight, inputs):
    module.zero_grad()
    for t in inputs.values():
        if t.grad is not None:
            t.grad = None
    (out * weight).sum().backward()
    pg = {k: p.grad.clone() for k, p in module.named_parameters() if p.grad is not None}
    ig = {k: t.grad.clone() for k, t in inputs.items() if t.grad is not None}
    return pg, ig


def attention_fixtures():
    g = gen(101)
    B, N, M, Cn, Cc, heads = 2, 24, 10, 64, 48, 2
    torch.manual_seed(11)
    sa = MultiHeadSelfAttention(Cn, heads, dropout=0.1).eval()
    ca = MultiHeadCrossAttention(Cn, Cc, heads, dropout=0.1).eval()
    x = randn(g, B, N, Cn).requires_grad_(True)
    ctx = randn(g, B, M, Cc).requires_grad_(True)
    w1, w2 = randn(g, B, N, Cn), randn(g, B, N, Cn)
    y1 = sa(x)
    pg1, ig1 = grads_of(sa, y1, w1, {"x": x})
    y2 = ca(x, ctx)
    pg2, ig2 = grads_of(ca, y2, w2, {"x": x, "ctx": ctx})
    save("attention", dict(meta=np.array([B, N, M, Cn, Cc, heads]), x=x, ctx=ctx, w_sa=w1, w_ca=w2,
                           sa_params=dict(sa.state_dict()), ca_params=dict(ca.state_dict()),
                           sa_out=y1, ca_out=y2, sa_pgrad=pg1, sa_igrad=ig1, ca_pgrad=pg2, ca_igrad=ig2))


def block_fixture():
    g = gen(202)
    B, N, M, Cn, Cc, cond_dim, heads = 2, 40, 12, 64, 48, 40, 2
    torch.manual_seed(12)
    blk = HybridViTBlock3D(Cn, num_heads=heads, context_dim=Cc, cond_dim=cond_dim).eval()
    reinit_adaln(blk, g)
    x = randn(g, B, N, Cn).requires_grad_(True)
    ctx = randn(g, B, M, Cc).requires_grad_(True)
    cond = randn(g, B, cond_dim).requires_grad_(True)
    w = randn(g, B, N, Cn)
    y = blk(x, ctx, cond)
    pg, ig = grads_of(blk, y, w, {"x": x, "ctx": ctx, "cond": cond})
    save("block", dict(meta=np.array([B, N, M, Cn, Cc, cond_dim, heads]), x=x, ctx=ctx, cond=cond, w=w,
                       params=dict(blk.state_dict()), out=y, pgrad=pg, igrad=ig))


def vit3d_fixtures():
    for tag, vs, in_ch in (("8", (8, 8, 8), 1), ("32", (32, 32, 32), 2), ("64x32x32", (64, 32, 32), 1)):
        g = gen(300 + sum(vs))
        B, M, Cn, Cc, cond_dim, heads, depth = (2 if tag == "8" else 1), 9, 64, 48, 40, 2, 2
        torch.manual_seed(13)
        m = HybridViT3D(volume_size=vs, in_channels=in_ch, voxel_dim=Cn, depth=depth, num_heads=heads,
                        context_dim=Cc, cond_dim=cond_dim).eval()
        reinit_adaln(m, g)
        x = randn(g, B, in_ch, *vs).requires_grad_(True)
        ctx = randn(g, B, M, Cc).requires_grad_(True)
        cond = randn(g, B, cond_dim).requires_grad_(True)
        w = randn(g, B, 1, *vs)
        y = m(x, ctx, cond)
        pg, ig = grads_of(m, y, w, {"x": x, "ctx": ctx, "cond": cond})
        save(f"vit3d_{tag}", dict(meta=np.array([B, M, Cn, Cc, cond_dim, heads, depth, in_ch, *vs, *m.downsampled_size]),
                                  x=x, ctx=ctx, cond=cond, w=w, params=dict(m.state_dict()), out=y, pgrad=pg, igrad=ig))


def xray_fixtures():
    g = gen(404)
    B, V, S, E, T, cond_dim = 2, 2, 64, 32, 16, 24
    torch.manual_seed(14)
    m = XrayConditioningModule(img_size=S, in_channels=1, embed_dim=E, num_views=V, time_embed_dim=T, cond_dim=cond_dim)
    with torch.no_grad():   # non-trivial running stats so eval mode is a real test
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(randn(g, *mod.running_mean.shape, scale=0.1))
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    xr = randn(g, B, V, 1, S, S).requires_grad_(True)
    t = randn(g, B, T).requires_grad_(True)
    w_ctx, w_cond, w_f = randn(g, B, cond_dim), randn(g, B, cond_dim), randn(g, B, E, S // 8, S // 8)
    out = {}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        params_before = {k: v.clone() for k, v in m.state_dict().items()}
        ctx, cond, feats = m(xr, t)
        m.zero_grad(); xr.grad = None; t.grad = None
        ((ctx * w_ctx).sum() + (cond * w_cond).sum() + (feats * w_f).sum()).backward()
        out[mode] = dict(params=params_before, ctx=ctx, cond=cond, feats=feats,
                         pgrad={k: p.grad.clone() for k, p in m.named_parameters()},
                         dxr=xr.grad.clone(), dt=t.grad.clone(),
                         stats_after={k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    arrays = dict(meta=np.array([B, V, S, E, T, cond_dim]), xrays=xr, t=t, w_ctx=w_ctx, w_cond=w_cond, w_f=w_f)
    for mode, d in out.items():
        for k, v in d.items():
            arrays[f"{mode}_{k}"] = v
    save("xray_cond", arrays)


def drr_fixtures():
    # SURVEY.md §9 known-answer recipe
    g = gen(1234)
    vol = torch.rand(2, 1, 8, 6, 4, generator=g) * 2 - 1
    r = DRRRenderer((8, 6, 4))
    ap, lat = r(vol.squeeze(1), 0), r(vol.squeeze(1), 90)
    xr = torch.rand(2, 1, 16, 16, generator=g) * 2 - 1
    pl0 = ProjectionLoss((8, 6, 4))(vol, xr, 0)
    pl90 = ProjectionLoss((8, 6, 4))(vol, xr, 90)
    xr2 = torch.rand(2, 2, 1, 16, 16, generator=g) * 2 - 1
    rl = DRRReprojectionLoss(img_size=16)
    volg = vol.clone().requires_grad_(True)
    l = rl(volg, xr2)
    l.backward()
    volp = vol.clone().requires_grad_(True)
    lp = ProjectionLoss((8, 6, 4))(volp, xr, 0) + ProjectionLoss((8, 6, 4))(volp, xr, 90)
    lp.backward()
    save("drr", dict(vol=vol, ap=ap, lat=lat, xr=xr, proj_loss_0=pl0, proj_loss_90=pl90, xr2=xr2,
                     reproj_loss=l, reproj_dvol=volg.grad, proj_dvol=volp.grad,
                     reproj_ap=rl.generate_drr(vol, 0), reproj_lat=rl.generate_drr(vol, 90)))


def tv_fixture():
    # TotalVariationLoss (loss_multiscale.py:140-188): prediction-only and match-the-target modes, with gradients
    g = gen(4321)
    pred = torch.rand(2, 1, 9, 7, 12, generator=g) * 2 - 1
    target = torch.rand(2, 1, 9, 7, 12, generator=g) * 0.5
    tv = TotalVariationLoss()
    p1 = pred.clone().requires_grad_(True)
    l1 = tv(p1)
    l1.backward()
    p2 = pred.clone().requires_grad_(True)
    l2 = tv(p2, target)
    l2.backward()
    flat = torch.full((1, 1, 4, 5, 6), 0.25)                 # all differences zero: sqrt(eps) plateau
    save("tv", dict(pred=pred, target=target, tv_pred=l1, tv_pred_grad=p1.grad, tv_match=l2, tv_match_grad=p2.grad, tv_flat=tv(flat)))


def direct_fixture():
    g = gen(505)
    cfg = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=32, vit_depth=2, num_heads=1, xray_feature_dim=32)
    torch.manual_seed(15)
    m = DirectCTRegression(**cfg)
    reinit_adaln(m, g)
    zero_dropout(m)
    B = 2
    xr = randn(g, B, 2, 1, 64, 64).requires_grad_(True)
    target = torch.rand(B, 1, 16, 16, 16, generator=g) * 2 - 1
    crit = DirectRegressionLoss(1.0, 0.5)
    arrays = dict(cfg=np.array([16, 16, 16, 64, 32, 2, 1, 32]), xrays=xr, target=target)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        params_before = {k: v.clone() for k, v in m.state_dict().items()}
        pred = m(xr)
        losses = crit(pred, target)
        m.zero_grad(); xr.grad = None
        losses["total_loss"].backward()
        arrays[f"{mode}_pred"] = pred
        arrays[f"{mode}_loss"] = torch.stack([losses["total_loss"], losses["l1_loss"], losses["ssim_loss"]])
        arrays[f"{mode}_psnr"] = np.array(compute_psnr(pred.detach(), target))
        arrays[f"{mode}_dxr"] = xr.grad.clone()
        arrays[f"{mode}_pgrad"] = {k: p.grad.clone() for k, p in m.named_parameters()}
        if mode == "eval":
            arrays["params"] = params_before
        else:
            arrays["train_stats_after"] = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    save("direct_small", arrays)


def cascade_fixture():
    """Stage 2 + stage 3 refiners and the shared multi-scale X-ray encoder at reduced sizes (16^3 -> 32^3 -> 64^3),
    chained exactly as ProgressiveCascadeModel.forward does (model_progressive.py:388-400)."""
    g = gen(606)
    torch.manual_seed(16)
    enc = MultiScaleXrayEncoder(img_size=64, in_channels=1, base_dim=32, num_views=2)
    s2 = Stage2Refiner128(volume_size=(32, 32, 32), voxel_dim=32, vit_depth=1, num_heads=1, xray_feature_dim=32)
    s3 = Stage3Refiner256(volume_size=(64, 64, 64), voxel_dim=32, vit_depth=1, num_heads=1, xray_feature_dim=32,
                          use_gradient_checkpointing=False)
    for m in (enc, s2, s3):
        reinit_adaln(m, g)
        m.eval()
    xr = randn(g, 1, 2, 1, 64, 64)
    v16 = (randn(g, 1, 1, 16, 16, 16) * 0.5).requires_grad_(True)
    w2, w3 = randn(g, 1, 1, 32, 32, 32), randn(g, 1, 1, 64, 64, 64)
    f1, c1, _ = enc(xr, stage=1)
    f2, cond2, _ = enc(xr, stage=2)
    v32 = s2(v16, f2, cond2)
    f3, cond3, _ = enc(xr, stage=3)
    v64 = s3(v32, f3, cond3)
    for m in (enc, s2, s3):
        m.zero_grad()
    ((v32 * w2).sum() + (v64 * w3).sum() + f1.sum() * 0.1).backward()
    save("cascade_small", dict(xrays=xr, v16=v16, w2=w2, w3=w3, feats1=f1, feats2=f2, v32=v32, v64=v64, dv16=v16.grad,
                               enc_params=dict(enc.state_dict()), s2_params=dict(s2.state_dict()), s3_params=dict(s3.state_dict()),
                               enc_pgrad={k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                               s2_pgrad={k: p.grad.clone() for k, p in s2.named_parameters() if p.grad is not None},
                               s3_pgrad={k: p.grad.clone() for k, p in s3.named_parameters() if p.grad is not None}))


def time_embedding_fixture():
    # SinusoidalTimeEmbedding (models/vit_components.py:152-174): integer-like and fractional timesteps, two widths
    g = gen(707)
    t = torch.cat([torch.tensor([0.0, 1.0, 17.0, 999.0]), torch.rand(4, generator=g) * 1000])
    save("time_embedding", dict(t=t, emb32=SinusoidalTimeEmbedding(32)(t), emb256=SinusoidalTimeEmbedding(256)(t)))


def frequency_fixture():
    # FrequencyLoss (loss_multiscale.py:191-236) on a non-cubic volume (the mask is centred on D//2, H//2, W//2 of the
    # UNSHIFTED spectrum, radius min(D,H,W)//4), loss and gradient, default and custom high-frequency weight
    g = gen(808)
    pred = torch.rand(2, 1, 12, 10, 16, generator=g) * 2 - 1
    target = torch.rand(2, 1, 12, 10, 16, generator=g) * 2 - 1
    arrays = dict(pred=pred, target=target)
    for tag, w in (("w2", 2.0), ("w05", 0.5)):
        p = pred.clone().requires_grad_(True)
        loss = FrequencyLoss(high_freq_weight=w)(p, target)
        loss.backward()
        arrays[f"loss_{tag}"], arrays[f"grad_{tag}"] = loss, p.grad
    save("frequency", arrays)


def train_step_fixture():
    """One optimisation step of the reference's direct trainer (direct_regression/train_direct_4gpu.py:59-75; the script
    itself cannot be imported here - nibabel - so its loop body is restated on the imported reference model):
    zero_grad -> forward -> DirectRegressionLoss(1.0, 0.5) -> backward -> clip_grad_norm_(1.0) -> AdamW(1e-4, wd 0.01).step(),
    train mode (BatchNorm batch statistics), dropout p = 0 (masks are not reproducible), fp32 on the CPU (autocast and
    GradScaler are no-ops there, train_direct.py:51,174).  Two consecutive steps, so the second one sees Adam state."""
    g = gen(909)
    cfg = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=32, vit_depth=2, num_heads=1, xray_feature_dim=32)
    torch.manual_seed(19)
    m = DirectCTRegression(**cfg)
    reinit_adaln(m, g)
    zero_dropout(m)
    m.train()
    xr = randn(g, 2, 2, 1, 64, 64)
    target = torch.rand(2, 1, 16, 16, 16, generator=g) * 2 - 1
    crit = DirectRegressionLoss(1.0, 0.5)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
    arrays = dict(cfg=np.array([16, 16, 16, 64, 32, 2, 1, 32]), xrays=xr, target=target,
                  params={k: v.clone() for k, v in m.state_dict().items()})
    for step in (1, 2):
        opt.zero_grad()
        pred = m(xr)
        loss = crit(pred, target)["total_loss"]
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        arrays[f"step{step}_loss"], arrays[f"step{step}_gradnorm"] = loss.detach(), norm
        arrays[f"step{step}_clipped"] = {k: p.grad.clone() for k, p in m.named_parameters()}
        opt.step()
        arrays[f"step{step}_after"] = {k: v.clone() for k, v in m.state_dict().items()}
    save("train_step", arrays, compact=("clipped", "after"))


def direct_kat():
    """Full-size known answers: values only, no tensors."""
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(64, 64, 64)).eval()
    x = torch.randn(1, 2, 1, 512, 512, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = m(x)
    idx = torch.tensor([[0, 0, 0, 0, 0], [0, 0, 31, 17, 5], [0, 0, 63, 63, 63], [0, 0, 12, 40, 7]])
    save("direct_kat64", dict(mean=y.mean(), std=y.std(), abssum=y.abs().sum(), idx=idx,
                              vals=torch.stack([y[tuple(i)] for i in idx]),
                              nparams=np.array(sum(p.numel() for p in m.parameters())),
                              keys=np.array(list(m.state_dict().keys())),
                              shapes=np.array([str(tuple(v.shape)) for v in m.state_dict().values()])))


if __name__ == "__main__":
    attention_fixtures()
    block_fixture()
    vit3d_fixtures()
    xray_fixtures()
    drr_fixtures()
    tv_fixture()
    direct_fixture()
    direct_kat()
    cascade_fixture()
    time_embedding_fixture()
    frequency_fixture()
    train_step_fixture()
