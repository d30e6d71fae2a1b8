"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI by
the host-side module mirrors, against (a) the golden vectors captured from the reference and (b) the
CPU oracle on the same seeded inputs.

Tolerances (north_star: "within 1e-3 rel fp32"):
  fp32 mode  : max|got-ref| <= 1e-3 max|ref| per tensor (split-bf16 MFMA products are ~1e-5)
  bf16 mode  : ||got-ref|| <= 4e-2 ||ref|| per tensor (relative Frobenius error; bf16 has 8 mantissa
               bits and the reference is fp32, so element-wise bounds on gradients formed by
               cancellation are meaningless: the CPU oracle itself under autocast(bf16) misses the
               fp32 initial_volume gradient by 22 % of its max while matching in norm).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3
BF16_TOL = 4e-2


def dev():
    return torch.device("cuda:0")


def _load(module, params):
    missing = module.load_state_dict(params, strict=True)
    return module.to(dev())


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def _zero_grad_biases(module):
    """Names of conv biases that feed a GroupNorm with one channel per group (or a train-mode BatchNorm): the
    normalisation removes the per-channel mean, so their gradient is exactly zero and both sides hold noise."""
    out = set()
    for name, seq in module.named_modules():
        if not isinstance(seq, torch.nn.Sequential):
            continue
        layers = list(seq)
        for i in range(len(layers) - 1):
            a, b = layers[i], layers[i + 1]
            if isinstance(a, (torch.nn.Conv2d, torch.nn.Conv3d)) and isinstance(b, torch.nn.GroupNorm) and b.num_groups == b.num_channels:
                out.add(f"{name}.{i}.bias" if name else f"{i}.bias")
    return out


def _run(mode, fn):
    if mode == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return fn()
    return fn()


def _tol(mode):
    return F32_TOL if mode == "f32" else BF16_TOL


def _metric(mode):
    return "max" if mode == "f32" else "l2"


def _native_loaded():
    from hvc import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libhvc_hip.so" in maps, "native library is not mapped into this process"


def test_native_library_is_loaded_and_reports_gfx950():
    import ctypes as C
    from hvc import _lib
    lib = _lib.load()
    _native_loaded()
    cu, wave = C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    assert lib.hvc_device_info(C.byref(cu), C.byref(wave), arch, 64) == 0
    assert wave.value == 64 and arch.value.decode().startswith("gfx950"), (wave.value, arch.value)
    assert cu.value == 256


def test_cpu_tensors_are_rejected_loudly():
    from hvc import ops
    x = torch.randn(4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(x, x)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_attention_modules_vs_golden(golden, mode):
    from models.vit_components import MultiHeadCrossAttention, MultiHeadSelfAttention
    g = golden("attention")
    B, N, M, Cn, Cc, heads = (int(v) for v in g.z["meta"])
    sa = _load(MultiHeadSelfAttention(Cn, heads).eval(), g.group("sa_params"))
    ca = _load(MultiHeadCrossAttention(Cn, Cc, heads).eval(), g.group("ca_params"))
    x = g.t("x").to(dev()).requires_grad_(True)
    ctx = g.t("ctx").to(dev()).requires_grad_(True)
    y = _run(mode, lambda: sa(x))
    g.check("", "sa_out", y, _tol(mode), metric=_metric(mode))
    (y.float() * g.t("w_sa").to(dev())).sum().backward()
    g.check("sa_igrad", "x", x.grad, _tol(mode), metric=_metric(mode))
    for k, p in sa.named_parameters():
        g.check("sa_pgrad", k, p.grad, _tol(mode), metric=_metric(mode))
    x.grad = None
    y = _run(mode, lambda: ca(x, ctx))
    g.check("", "ca_out", y, _tol(mode), metric=_metric(mode))
    (y.float() * g.t("w_ca").to(dev())).sum().backward()
    g.check("ca_igrad", "x", x.grad, _tol(mode), metric=_metric(mode))
    g.check("ca_igrad", "ctx", ctx.grad, _tol(mode), metric=_metric(mode))
    for k, p in ca.named_parameters():
        g.check("ca_pgrad", k, p.grad, _tol(mode), metric=_metric(mode))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_block_vs_golden(golden, mode):
    from models.hybrid_vit_backbone import HybridViTBlock3D
    g = golden("block")
    B, N, M, Cn, Cc, cond_dim, heads = (int(v) for v in g.z["meta"])
    blk = _load(HybridViTBlock3D(Cn, num_heads=heads, context_dim=Cc, cond_dim=cond_dim).eval(), g.group("params"))
    x, ctx, cond = (g.t(k).to(dev()).requires_grad_(True) for k in ("x", "ctx", "cond"))
    y = _run(mode, lambda: blk(x, ctx, cond))
    assert y.dtype == torch.float32          # residual stream stays fp32
    g.check("", "out", y, _tol(mode), metric=_metric(mode))
    (y * g.t("w").to(dev())).sum().backward()
    for k, t in (("x", x), ("ctx", ctx), ("cond", cond)):
        g.check("igrad", k, t.grad, _tol(mode), metric=_metric(mode))
    for k, p in blk.named_parameters():
        g.check("pgrad", k, p.grad, _tol(mode), metric=_metric(mode))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("tag", ["8", "32", "64x32x32"])
def test_hybrid_vit3d_vs_golden(golden, tag, mode):
    from models.hybrid_vit_backbone import HybridViT3D
    g = golden("vit3d_" + tag)
    meta = [int(v) for v in g.z["meta"]]
    B, M, Cn, Cc, cond_dim, heads, depth, in_ch = meta[:8]
    vs, ds = tuple(meta[8:11]), tuple(meta[11:14])
    m = HybridViT3D(volume_size=vs, in_channels=in_ch, voxel_dim=Cn, depth=depth, num_heads=heads, context_dim=Cc,
                    cond_dim=cond_dim).eval()
    assert m.downsampled_size == ds                 # bit-exact token-grid geometry
    _load(m, g.group("params"))
    x, ctx, cond = (g.t(k).to(dev()).requires_grad_(True) for k in ("x", "ctx", "cond"))
    y = _run(mode, lambda: m(x, ctx, cond))
    g.check("", "out", y, _tol(mode), metric=_metric(mode))
    (y.float() * g.t("w").to(dev())).sum().backward()
    for k, t in (("x", x), ("ctx", ctx), ("cond", cond)):
        g.check("igrad", k, t.grad, _tol(mode), metric=_metric(mode))
    for k, p in m.named_parameters():
        g.check("pgrad", k, p.grad, _tol(mode), metric=_metric(mode))


def _hip_routing(records):
    """ReLU masks and max-pool arg-max indices the HIP X-ray stem used, in the oracle's form (NCHW mask, flat h*W+w
    index per pooled window), from the operands captured at each hvc_bn_relu_pool_fwd call."""
    route = {}
    for tag, (x, amax, stats, gamma, beta, pool) in zip(("1", "5", "9"), records):
        x, stats = x.float().cpu(), stats.cpu()
        a = stats[:, 1] * gamma.cpu()
        b = x * a + (beta.cpu() - stats[:, 0] * a)                       # bn(x), channels-last (N,H,W,C): the kernel's own formula
        rec = {"relu": (b > 0).permute(0, 3, 1, 2).contiguous()}
        if pool is not None:
            k, s_, p_ = pool
            W = x.shape[2]
            am = amax.cpu().long()                                        # (N,HP,WP,C), window-local kh*k+kw
            assert int(am.max()) < k * k
            hp = torch.arange(am.shape[1]).view(1, -1, 1, 1)
            wp = torch.arange(am.shape[2]).view(1, 1, -1, 1)
            flat = (hp * s_ + am // k - p_) * W + (wp * s_ + am % k - p_)
            rec["argmax"] = flat.permute(0, 3, 1, 2).contiguous()
        route[tag] = rec
    return route


def _capture_stem_routing(monkeypatch):
    """Records the operands of every hvc_bn_relu_pool_fwd call (-> _hip_routing): the ReLU gates and max-pool arg-max
    choices the HIP X-ray stem really used."""
    from hvc import ops
    records = []
    real_fwd = ops.bn_relu_pool_fwd

    def capture(x, gamma, beta, running_mean, running_var, pool, training, *a, **kw):
        y, amax, stats = real_fwd(x, gamma, beta, running_mean, running_var, pool, training, *a, **kw)
        records.append((x.detach().clone(), None if amax is None else amax.clone(), stats.clone(), gamma.detach().clone(),
                        beta.detach().clone(), pool))
        return y, amax, stats
    monkeypatch.setattr(ops, "bn_relu_pool_fwd", capture)
    return records


def _routing_agreement(route, hip_route):
    """Fraction check of the routing-aware method: the HIP stem's routing differs from the oracle's own in < 1e-3 of the live
    pool windows and ReLU gates (route["own"] is filled by the oracle evaluation that USED hip_route)."""
    live = flipped = acts = act_flips = 0
    for tag, own in route["own"].items():
        acts += own["relu"].numel()
        act_flips += int((own["relu"] != hip_route[tag]["relu"]).sum())
        if "argmax" in own:
            alive = own["max"] > 0                      # windows whose maximum survives the ReLU carry gradient
            live += int(alive.sum())
            flipped += int(((own["argmax"] != hip_route[tag]["argmax"]) & alive).sum())
    assert flipped < 1e-3 * live and act_flips < 1e-3 * acts, (flipped, live, act_flips, acts)
    return flipped, live, act_flips, acts


def _maxrel(got, ref):
    return ((got.detach().double().cpu() - ref.detach().double().cpu()).abs().max() / ref.detach().abs().max().clamp_min(1e-6)).item()


def _note(key, err, tol, metric="max"):
    """Same report line as conftest.Golden.check writes (HVC_TEST_REPORT), for comparisons made against the oracle directly."""
    import os
    path = os.environ.get("HVC_TEST_REPORT")
    if path:
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]
        with open(path, "a") as f:
            f.write(f"{test}\t{key}\t{metric}\t{err:.3e}\t{tol:.1e}\n")


@pytest.mark.parametrize("train", [False, True])
def test_xray_conditioning_vs_golden(golden, train, monkeypatch):
    """Outputs vs the reference's golden vectors at 1e-3; gradients that do NOT pass the ReLU / max-pool routing at 1e-3.
    Gradients behind the routing (X-ray pixels, conv / BN parameters) are discontinuous in the forward values: a near-tie
    that rounds the other way re-routes a window.  They are therefore checked in two exact steps instead of a loose norm:
      (1) the routing the HIP stem used (captured from its kernels' operands) differs from the oracle's own routing in
          < 1e-3 of the live windows / activations, and
      (2) the oracle evaluated WITH the HIP routing (oracle._relu_pool) reproduces every HIP gradient to 1e-3 max-rel."""
    from models.diagnostic_losses import XrayConditioningModule
    from hvc import ops
    from oracle import hvc_oracle as O
    g = golden("xray_cond")
    B, V, S, E, T, cond_dim = (int(v) for v in g.z["meta"])
    mode = "train" if train else "eval"
    m = XrayConditioningModule(img_size=S, in_channels=1, embed_dim=E, num_views=V, time_embed_dim=T, cond_dim=cond_dim)
    _load(m, g.group(f"{mode}_params")).train(train)
    records = _capture_stem_routing(monkeypatch)
    xr, t = g.t("xrays").to(dev()).requires_grad_(True), g.t("t").to(dev()).requires_grad_(True)
    ctx, cond, feats = m(xr, t)
    assert len(records) == 3
    g.check("", f"{mode}_ctx", ctx, F32_TOL)
    g.check("", f"{mode}_cond", cond, F32_TOL)
    g.check("", f"{mode}_feats", feats, F32_TOL)
    w_ctx, w_cond, w_f = g.t("w_ctx"), g.t("w_cond"), g.t("w_f")
    ((ctx * w_ctx.to(dev())).sum() + (cond * w_cond.to(dev())).sum() + (feats * w_f.to(dev())).sum()).backward()
    g.check("", f"{mode}_dt", t.grad, F32_TOL)
    for k, p in m.named_parameters():
        if not k.startswith("encoder."):
            g.check(f"{mode}_pgrad", k, p.grad, F32_TOL)
    if train:
        for k, v in m.state_dict().items():
            if "running" in k:
                g.check("train_stats_after", k, v, F32_TOL)
    # (1) + (2): routing-aware gradient comparison
    hip_route = _hip_routing(records)
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k) for k, v in g.group(f"{mode}_params").items()}
    xr_c = g.t("xrays").requires_grad_(True)
    route = {"use": hip_route}
    c2, d2, f2 = O.xray_conditioning(xr_c, g.t("t"), P, "", train, {}, route)
    ((c2 * w_ctx).sum() + (d2 * w_cond).sum() + (f2 * w_f).sum()).backward()
    flipped, live, act_flips, acts = _routing_agreement(route, hip_route)
    _note(f"{mode}_dxr(routed)", _maxrel(xr.grad, xr_c.grad), F32_TOL)
    assert _maxrel(xr.grad, xr_c.grad) < F32_TOL, _maxrel(xr.grad, xr_c.grad)
    for k, p in m.named_parameters():
        if k.startswith("encoder."):
            if train and k.endswith(("encoder.0.bias", "encoder.4.bias", "encoder.8.bias")):
                continue   # exactly-zero gradients (bias ahead of train-mode BN): rounding noise only
            _note(f"{mode}_pgrad(routed)/{k}", _maxrel(p.grad, P[k].grad), F32_TOL)
            assert _maxrel(p.grad, P[k].grad) < F32_TOL, (k, _maxrel(p.grad, P[k].grad))
    print(f"xray stem routing [{mode}]: {flipped}/{live} live pool windows and {act_flips}/{acts} ReLU gates resolved differently")


def test_drr_vs_golden_and_known_answers(golden):
    from models.diagnostic_losses import DRRRenderer, ProjectionLoss
    g = golden("drr")
    vol = g.t("vol").to(dev())
    r = DRRRenderer((8, 6, 4))
    ap, lat = r(vol.squeeze(1), 0), r(vol.squeeze(1), 90)
    g.check("", "ap", ap, 1e-5)
    g.check("", "lat", lat, 1e-5)
    assert ap.shape == (2, 6, 4) and lat.shape == (2, 6, 8)
    assert abs(ap.sum().item() - 285.885590) < 2e-3                 # SURVEY.md §9 known answers
    v = vol.clone().requires_grad_(True)
    pl = ProjectionLoss((8, 6, 4))
    l0, l90 = pl(v, g.t("xr").to(dev()), 0), pl(v, g.t("xr").to(dev()), 90)
    assert abs(l0.item() - 35.666653) < 1e-3
    g.check("", "proj_loss_0", l0, 1e-5)
    g.check("", "proj_loss_90", l90, 1e-5)
    (l0 + l90).backward()
    g.check("", "proj_dvol", v.grad, 1e-4)


# Tolerance policy of the model-level tests below.  fp32 mode: max|got - ref| <= 1e-3 max|ref| for EVERY tensor, no multipliers.
# Tensors behind the X-ray stem's ReLU / max-pool routing (X-ray pixel gradients, encoder.* parameters) are compared with the
# oracle evaluated WITH the routing the HIP stem used (a near-tie that rounds the other way re-routes a whole window, which no
# tolerance can absorb): same 1e-3, plus < 1e-3 of the routing decisions may differ.  bf16 mode (the trainers' autocast): relative
# Frobenius error <= 4e-2; routed tensors in bf16 are only held to a norm bound of 0.4 - the bf16 stem resolves ~1 % of its
# near-ties differently and fp32 mode is the parity statement for them.
_ROUTED_BF16_TOL = 0.4
# bf16 mode, parameter / input gradients of whole models: gradients formed by cancellation over thousands of bf16 products
# (pos_embed, first-layer conv weights, GroupNorm affine parameters) sit at 5e-2 .. 1.2e-1 relative Frobenius error against the
# fp32 reference - the CPU oracle under autocast(bf16) misses them by as much; outputs keep 4e-2.
_BF16_GRAD_TOL = 1.5e-1


def _gtol(mode):
    return F32_TOL if mode == "f32" else _BF16_GRAD_TOL


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("train", [False, True])
def test_direct_regression_small_vs_golden(golden, train, mode, monkeypatch):
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    g = golden("direct_small")
    cfg = [int(v) for v in g.z["cfg"]]
    m = DirectCTRegression(volume_size=tuple(cfg[:3]), xray_img_size=cfg[3], voxel_dim=cfg[4], vit_depth=cfg[5],
                           num_heads=cfg[6], xray_feature_dim=cfg[7])
    _load(m, g.group("params"))
    _zero_dropout(m)
    m.train(train)
    tag = "train" if train else "eval"
    xr = g.t("xrays").to(dev()).requires_grad_(True)
    target = g.t("target").to(dev())
    crit = DirectRegressionLoss(1.0, 0.5)
    records = _capture_stem_routing(monkeypatch)

    def step():
        pred = m(xr)
        return pred, crit(pred.float(), target)
    pred, losses = _run(mode, step)
    tol = _tol(mode)
    g.check("", f"{tag}_pred", pred, tol, metric=_metric(mode))
    ref = g.z[f"{tag}_loss"]
    got = [losses[k].item() for k in ("total_loss", "l1_loss", "ssim_loss")]
    assert np.allclose(got, ref, rtol=tol), (got, ref)
    from oracle import hvc_oracle as O
    assert abs(O.psnr(pred.detach().float().cpu(), target.cpu()) - float(g.z[f"{tag}_psnr"])) < 0.1   # dB
    losses["total_loss"].backward()
    routed = lambda k: k.startswith("xray_encoder.encoder.")
    for k, p in m.named_parameters():
        if train and k.endswith(("encoder.0.bias", "encoder.4.bias", "encoder.8.bias")):
            continue                       # exactly-zero gradients (bias ahead of train-mode BN): rounding noise only
        if not routed(k):
            g.check(f"{tag}_pgrad", k, p.grad, _gtol(mode), metric=_metric(mode))
        elif mode == "bf16":
            g.check(f"{tag}_pgrad", k, p.grad, _ROUTED_BF16_TOL, metric="l2")
    if mode == "bf16":
        g.check("", f"{tag}_dxr", xr.grad, _ROUTED_BF16_TOL, metric="l2")
        return
    # fp32 mode, routed tensors: the oracle evaluated with the routing the HIP stem used
    hip_route = _hip_routing(records)
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k) for k, v in g.group("params").items()}
    xr_c = g.t("xrays").requires_grad_(True)
    route = {"use": hip_route}
    pred_c = O.direct_ct_regression(xr_c, P, tuple(cfg[:3]), cfg[4], cfg[5], cfg[6], training=train, new_stats={}, route=route)
    O.direct_regression_loss(pred_c, g.t("target"))["total_loss"].backward()
    _routing_agreement(route, hip_route)
    _note(f"{tag}_dxr(routed)", _maxrel(xr.grad, xr_c.grad), tol)
    assert _maxrel(xr.grad, xr_c.grad) < tol, _maxrel(xr.grad, xr_c.grad)
    for k, p in m.named_parameters():
        if routed(k) and not (train and k.endswith(("encoder.0.bias", "encoder.4.bias", "encoder.8.bias"))):
            _note(f"{tag}_pgrad(routed)/{k}", _maxrel(p.grad, P[k].grad), tol)
            assert _maxrel(p.grad, P[k].grad) < tol, (k, _maxrel(p.grad, P[k].grad))


# ----------------------------------------------------------------------------------------------
# kernel-level checks through the C ABI against the oracle, incl. ragged / edge shapes
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 1, 1, 1, 32), (2, 3, 65, 1, 64), (1, 2, 129, 257, 32), (2, 4, 128, 64, 64),
                                   (1, 1, 31, 63, 64),
                                   (1, 2, 2048, 64, 32), (1, 1, 1100, 130, 64)])   # few key blocks: query-range-sliced dK/dV
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_kernel_edge_shapes(shape, dtype):
    from hvc import ops
    from oracle import hvc_oracle as O
    B, H, Nq, Nk, D = shape
    g = torch.Generator().manual_seed(sum(shape))
    q, k, v = (torch.randn(B, n, H, D, generator=g) for n in (Nq, Nk, Nk))
    qd, kd, vd = (t.to(dev(), dtype) for t in (q, k, v))
    qr, kr, vr = (t.float().cpu().permute(0, 2, 1, 3).requires_grad_(True) for t in (qd, kd, vd))
    o_ref = O.attention_core(qr, kr, vr, D ** -0.5)
    do = torch.randn(B, Nq, H, D, generator=g)
    o_ref.backward(do.permute(0, 2, 1, 3))
    o, lse = ops.attention_fwd(qd, kd, vd, D ** -0.5)
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, o, do.to(dev(), dtype), lse, D ** -0.5)
    tol = F32_TOL if dtype == torch.float32 else 3e-2

    def rel(a, b):   # floor 1.0 = natural scale of |dO||V||K|/sqrt(D) for unit-variance operands: with a
        # single key the softmax is constant, dq and dk are exactly zero and only rounding noise remains
        return ((a.float().cpu() - b).abs().max() / max(b.abs().max().item(), 1.0)).item()
    assert rel(o, o_ref.permute(0, 2, 1, 3)) < tol
    assert rel(dq, qr.grad.permute(0, 2, 1, 3)) < tol
    assert rel(dk, kr.grad.permute(0, 2, 1, 3)) < tol
    assert rel(dv, vr.grad.permute(0, 2, 1, 3)) < tol


@pytest.mark.parametrize("seed", range(8))
def test_attention_random_shapes_and_packed_strides(seed):
    """Randomised ragged shapes, with q/k/v read in place from a packed (B, N, 3, H, D) projection as the modules do
    (vit_components.py:41-43) and gradients written into the packed buffer, fp32 and bf16, with dropout consistency
    checked through linearity in V; odd sizes exercise the peeled ragged tile, the clamped loads and the tile parity logic."""
    from hvc import ops
    from oracle import hvc_oracle as O
    rng = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    B, H, D = ri(1, 2), ri(1, 3), (32, 64)[seed % 2]
    N = ri(1, 330)
    dtype = (torch.float32, torch.bfloat16)[(seed // 2) % 2]
    qkv = torch.randn(B, N, 3, H, D, generator=rng).to(dev(), dtype)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    qr, kr, vr = (t.float().cpu().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))
    o_ref = O.attention_core(qr, kr, vr, D ** -0.5)
    do = torch.randn(B, N, H, D, generator=rng)
    o_ref.backward(do.permute(0, 2, 1, 3))
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(q, k, v, o, do.to(dev(), dtype), lse, D ** -0.5, dq=dqkv[:, :, 0], dk=dqkv[:, :, 1], dv=dqkv[:, :, 2])
    tol = F32_TOL if dtype == torch.float32 else 3e-2
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / max(b.abs().max().item(), 1.0)).item()
    assert rel(o, o_ref.permute(0, 2, 1, 3)) < tol, (B, H, N, D, dtype)
    for i, ref in enumerate((qr.grad, kr.grad, vr.grad)):
        assert rel(dqkv[:, :, i], ref.permute(0, 2, 1, 3)) < tol, (i, B, H, N, D, dtype)
    # dropout on the same ragged shape: O is linear in V for a fixed seed, and <dO, O> = <dV, V>
    p, sd = 0.3, 555 + seed
    o1, lse1 = ops.attention_fwd(q, k, v, D ** -0.5, p, sd)
    o2, _ = ops.attention_fwd(q, k, (2 * v.float()).to(dtype), D ** -0.5, p, sd)
    assert rel(o2, 2 * o1.float().cpu()) < (1e-5 if dtype == torch.float32 else 2e-2)
    _, _, dv = ops.attention_bwd(q, k, v, o1, do.to(dev(), dtype), lse1, D ** -0.5, p, sd)
    lhs, rhs = (do.double() * o1.double().cpu()).sum().item(), (dv.double() * v.double()).sum().item()
    assert abs(lhs - rhs) < (2e-3 if dtype == torch.float32 else 5e-2) * (abs(lhs) + abs(rhs) + 1.0)


def test_attention_rejects_unsupported_head_dim():
    from hvc import ops
    q = torch.randn(1, 8, 1, 48, device=dev())
    with pytest.raises(RuntimeError, match="head dim"):
        ops.attention_fwd(q, q, q, 1.0)


def test_attention_online_softmax_rescale_branch():
    """Forces the running-max rescale: one key far above the rest appears in a late tile."""
    from hvc import ops
    from oracle import hvc_oracle as O
    B, H, N, D = 1, 1, 256, 64
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(B, N, H, D, generator=g) for _ in range(3))
    k[0, 200, 0] = q[0, 17, 0] * 6.0        # spike at tile 3 for query 17
    o, _ = ops.attention_fwd(q.to(dev()), k.to(dev()), v.to(dev()), D ** -0.5)
    ref = O.attention_core(*(t.permute(0, 2, 1, 3) for t in (q, k, v)), D ** -0.5).permute(0, 2, 1, 3)
    assert ((o.cpu() - ref).abs().max() / ref.abs().max()).item() < F32_TOL


def test_attention_dropout_statistics_determinism_and_gradient_consistency():
    from hvc import ops
    B, H, N, D = 2, 2, 192, 32
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, N, H, D, generator=g).to(dev()) for _ in range(3))
    p = 0.1
    # with v = 1 the output of row i is sum_j keep_ij P_ij / (1-p): mean over rows -> 1
    ones = torch.ones_like(v)
    o1, lse = ops.attention_fwd(q, k, ones, D ** -0.5, p, 1234)
    o2, _ = ops.attention_fwd(q, k, ones, D ** -0.5, p, 1234)
    o3, _ = ops.attention_fwd(q, k, ones, D ** -0.5, p, 99)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)          # seeded, reproducible
    assert abs(o1.mean().item() - 1.0) < 0.02
    # O is linear in V for a fixed mask: <dO, O(V + E) - O(V)> == <dV, E> ; ties fwd mask to bwd mask
    do = torch.randn(B, N, H, D, generator=g).to(dev())
    E = torch.randn(B, N, H, D, generator=g).to(dev())
    oa, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 77)
    ob, _ = ops.attention_fwd(q, k, v + E, D ** -0.5, p, 77)
    dq, dk, dv = ops.attention_bwd(q, k, v, oa, do, lse, D ** -0.5, p, 77)
    lhs, rhs = (do * (ob - oa)).sum().item(), (dv * E).sum().item()
    assert abs(lhs - rhs) < 2e-3 * (abs(lhs) + abs(rhs) + 1)
    # directional finite difference for q and k (mask fixed by the seed)
    for name, grad in (("q", dq), ("k", dk)):
        Eq = torch.randn(B, N, H, D, generator=g).to(dev()) * 1e-2
        args_p = dict(q=q, k=k)
        args_m = dict(q=q, k=k)
        args_p[name] = args_p[name] + Eq
        args_m[name] = args_m[name] - Eq
        op_, _ = ops.attention_fwd(args_p["q"], args_p["k"], v, D ** -0.5, p, 77)
        om_, _ = ops.attention_fwd(args_m["q"], args_m["k"], v, D ** -0.5, p, 77)
        fd = (do * (op_ - om_)).sum().item() / 2
        an = (grad * Eq).sum().item()
        assert abs(fd - an) < 2e-2 * (abs(fd) + abs(an)) + 1e-3, (name, fd, an)


@pytest.mark.parametrize("shape", [(2, 4, 4096, 4096 - 37, 64), (1, 8, 32768, 1024, 32), (2, 2, 4096, 64, 32)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_is_bit_reproducible_with_dropout(shape, dtype):
    """Same inputs, same seed -> the same bits, run after run, for every output of the forward and the backward (no atomics on
    the path; the reference's SDPA + nn.Dropout under a fixed generator is reproducible as well).  Round 3 found dQ differing
    between runs on exactly these shapes: an asm select read MFMA accumulators ahead of the hazard window."""
    from hvc import ops
    B, H, N, M, D = shape
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, N, H, D, generator=g).to(dev(), dtype)
    k = torch.randn(B, M, H, D, generator=g).to(dev(), dtype)
    v = torch.randn(B, M, H, D, generator=g).to(dev(), dtype)
    first = None
    for _ in range(4):
        o, lse = ops.attention_fwd(q, k, v, D ** -0.5, 0.1, 7)
        do = torch.full_like(o, 0.5)
        cur = (o, lse) + tuple(ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, 0.1, 7))
        if first is None:
            first = [t.clone() for t in cur]
        else:
            for name, a, b in zip(("o", "lse", "dq", "dk", "dv"), first, cur):
                assert torch.equal(a, b), name


@pytest.mark.parametrize("kind", ["plain", "bias", "dx", "fc1", "fc2_dx", "proj", "cross_proj"])
def test_gemm_persistent_static_epilogues_match_the_generic_kernel(kind, hvc_option):
    """Token-matrix GEMMs (>= 1024 interior 128 x 128 tiles) run as persistent workgroups with straight-line epilogues - one per fused
    projection of the block (round 3).  Same k loop, same epilogue arithmetic in the same order as the generic one-tile-per-workgroup kernel
    (HVC_GEMM_PERSISTENT=0), so every output (C, saved pre-activation, zsave) must agree bit for bit (up to per-instantiation FMA
    contraction: one-ulp differences behind dropout); the generic kernel is the one the oracle tests cover on small shapes."""
    from hvc import ops
    M = 65536 + 128
    g = torch.Generator().manual_seed(11)
    rnd = lambda *sh, dt=torch.bfloat16, s=1.0: (torch.randn(*sh, generator=g) * s).to(dev(), dt)
    kw, extra = {}, []
    if kind in ("plain", "bias"):
        a, b = rnd(M, 256), rnd(768, 256, s=0.06)
        if kind == "bias":
            kw["bias"] = rnd(768, dt=torch.float32)
    elif kind == "dx":
        a, b = rnd(M, 768), rnd(768, 256, s=0.06)
        kw["b_kmajor"] = True
    elif kind == "fc1":
        a, b = rnd(M, 256), rnd(1024, 256, s=0.06)
        kw.update(bias=rnd(1024, dt=torch.float32), act=ops.ACT_GELU, p_drop=0.1, seed=5)
        extra = ["aux"]
    elif kind == "fc2_dx":
        a, b = rnd(M, 256), rnd(256, 1024, s=0.06)
        kw.update(b_kmajor=True, act=ops.ACT_GELU_GRAD, p_drop=0.1, seed=5)
        extra = ["aux_in"]
    else:
        a, b = rnd(M, 1024 if kind == "proj" else 256), rnd(256, 1024 if kind == "proj" else 256, s=0.06)
        kw.update(bias=rnd(256, dt=torch.float32), residual=rnd(M, 256, dt=torch.float32), p_drop=0.1, seed=9, out_dtype=torch.float32)
        if kind == "proj":
            kw.update(gate=rnd((M + 2047) // 2048, 256, dt=torch.float32), rows_per_batch=2048)
            extra = ["zsave"]
    N = b.shape[1] if kw.get("b_kmajor") else b.shape[0]
    pre_in = rnd(M, N) if "aux_in" in extra else None
    outs = {}
    for form in ("0", "1"):
        hvc_option("HVC_GEMM_PERSISTENT", form)
        call = dict(kw)
        side = None
        if "aux" in extra:
            side = call["aux"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
        if "aux_in" in extra:
            call["aux"] = pre_in
        if "zsave" in extra:
            side = call["zsave"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
        outs[form] = (ops.gemm(a, b, **call), side)
    assert torch.isfinite(outs["1"][0].float()).all()

    def same(x, y):
        # hipcc contracts multiply-adds per instantiation (x * Phi * keep_scale behind GELU + dropout; v * gate + residual): elements may
        # land an ulp of the output dtype apart (measured: 67 of 67 M bf16 elements for fc1, a quarter of the fp32 elements of proj, all
        # by one ulp); anything beyond two ulps is a different computation
        if torch.equal(x, y):
            return True
        eps = 2.0 ** -7 if x.dtype == torch.bfloat16 else 2.0 ** -22
        return torch.allclose(x.float(), y.float(), rtol=eps, atol=eps)

    assert same(outs["0"][0], outs["1"][0])
    if outs["0"][1] is not None:
        assert same(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("shape", [(2, 4, 700, 515, 0.1), (1, 2, 1500, 97, 0.0), (2, 2, 513, 4096, 0.25)])
def test_attention_forward_eight_wavefront_workgroups_match_four(shape, hvc_option):
    """The 64-row forward kernel (d = 64) has a 512-row form (eight wavefronts, one workgroup per CU; picked by problem size, here pinned
    through HVC_ATTN_FWD_WAVES): same arithmetic per wavefront, so output and log-sum-exp must be bit-identical on ragged shapes,
    with and without dropout."""
    from hvc import ops
    B, H, Nq, Nk, p = shape
    D = 64
    g = torch.Generator().manual_seed(Nq * 3 + Nk)
    q = torch.randn(B, Nq, H, D, generator=g).to(dev(), torch.bfloat16)
    k = torch.randn(B, Nk, H, D, generator=g).to(dev(), torch.bfloat16)
    v = torch.randn(B, Nk, H, D, generator=g).to(dev(), torch.bfloat16)
    hvc_option("HVC_ATTN_FWD_ROWS", "64")
    outs = {}
    for waves in ("4", "8"):
        hvc_option("HVC_ATTN_FWD_WAVES", waves)
        outs[waves] = ops.attention_fwd(q, k, v, D ** -0.5, p, 33)
    assert torch.isfinite(outs["8"][0].float()).all()
    assert torch.equal(outs["4"][0], outs["8"][0]) and torch.equal(outs["4"][1], outs["8"][1])


@pytest.mark.parametrize("shape", [(2, 4, 700, 515, 64, 0.1), (1, 8, 1030, 1100, 32, 0.1), (2, 2, 256, 4096, 64, 0.0), (1, 2, 65, 257, 32, 0.25)])
def test_attention_backward_eight_wavefront_workgroups_match_four(shape, hvc_option):
    """The dQ and dK/dV kernels have a 256-row / 256-key form (eight wavefronts, one workgroup per CU; picked by problem size, here
    pinned through HVC_ATTN_BWD_WAVES).  A wavefront does the same arithmetic in the same order in both forms, so the gradients
    must be bit-identical - ragged sizes, both head dims, with and without dropout (the 4-wavefront form is the one the oracle
    tests cover on small shapes)."""
    from hvc import ops
    B, H, Nq, Nk, D, p = shape
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(B, Nq, H, D, generator=g).to(dev(), torch.bfloat16)
    k = torch.randn(B, Nk, H, D, generator=g).to(dev(), torch.bfloat16)
    v = torch.randn(B, Nk, H, D, generator=g).to(dev(), torch.bfloat16)
    do = torch.randn(B, Nq, H, D, generator=g).to(dev(), torch.bfloat16)
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 21)
    grads = {}
    for waves in ("4", "8"):
        hvc_option("HVC_ATTN_BWD_WAVES", waves)
        grads[waves] = ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 21)
    for name, a, b in zip(("dq", "dk", "dv"), grads["4"], grads["8"]):
        assert torch.isfinite(b.float()).all(), name
        assert torch.equal(a, b), name


@pytest.mark.parametrize("p,N", [(0.1, 256), (0.25, 256), (0.1, 1024)])
def test_attention_dropout_mask_is_bernoulli_like(p, N):
    """Recover the full keep mask of the counter-based dropout (reference: nn.Dropout on the probabilities,
    vit_components.py:48 / :105) and check it behaves like iid Bernoulli(1-p): global / per-row / per-column
    rates and no serial correlation along keys (inside and across the 4-key hash groups), queries or heads."""
    from hvc import ops
    B, H, D = 1, 2, 32                                   # N = 1024: 2 M mask bits, in-group correlations resolved to 0.007
    q = torch.zeros(B, N, H, D, device=dev())           # uniform probabilities 1/N, exactly representable
    k = torch.zeros(B, N, H, D, device=dev())
    cols = []
    for c in range(N // D):                              # V = one-hot over the 32 keys of chunk c -> O[i, d] = keep(i, 32c + d) / (N (1-p))
        v = torch.zeros(B, N, H, D, device=dev())
        idx = torch.arange(D, device=dev())
        v[:, 32 * c + idx, :, idx] = 1.0
        o, _ = ops.attention_fwd(q, k, v, D ** -0.5, p, 4321)
        cols.append(o * N * (1 - p))
    m = torch.cat(cols, dim=-1).permute(0, 2, 1, 3).reshape(B * H, N, N)      # [bh][query][key]
    assert ((m - m.round()).abs() < 1e-3).all() and set(m.round().unique().tolist()) <= {0.0, 1.0}
    m = m.round().double().cpu()
    n = m.numel()
    sd = (p * (1 - p)) ** 0.5
    assert abs(m.mean().item() - (1 - p)) < 4 * sd / n ** 0.5
    assert (m.mean(dim=2) - (1 - p)).abs().max().item() < 5 * sd / N ** 0.5     # per query row
    assert (m.mean(dim=1) - (1 - p)).abs().max().item() < 5 * sd / N ** 0.5     # per key column
    z = (m - (1 - p)) / sd
    def corr(a, b):
        return (a * b).mean().item()
    lim = 5 / (n / 2) ** 0.5
    for lag in (1, 2, 3, 4, 5, 8, 64):                    # along keys: lags 1-3 mix in-group and cross-group pairs
        assert abs(corr(z[:, :, :-lag], z[:, :, lag:])) < lim, ("key lag", lag)
    for lag in (1, 2, 32):                                # along queries
        assert abs(corr(z[:, :-lag, :], z[:, lag:, :])) < lim, ("query lag", lag)
    assert abs(corr(z[0], z[1])) < lim                    # between heads
    assert abs(corr(z[:, :-1, :-1], z[:, 1:, 1:])) < lim  # diagonal neighbours
    # keys 4g .. 4g+3 share one hash evaluation: check every in-group pair separately
    g = z.reshape(B * H, N, N // 4, 4)
    for a in range(4):
        for b in range(a + 1, 4):
            assert abs(corr(g[..., a], g[..., b])) < 5 / (n / 4) ** 0.5, ("in-group", a, b)


def test_seed_counter_is_seen_by_the_autograd_thread():
    """hvc_set_seed_counter is process-wide: the backward of an autograd Function runs on torch's device worker
    thread and must fold the SAME device counter into its dropout seed as the forward did on the caller's thread.  The
    GEMM-epilogue dropout of hvc.functional.linear: recover the forward's mask from its zeros, and require the input gradient
    to be the one of exactly that mask."""
    from hvc import _lib, functional as HF
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 96, generator=g).to(dev()).requires_grad_(True)
    w = torch.randn(128, 96, generator=g).to(dev())
    counter = torch.full((1,), 5, dtype=torch.int32, device=dev())
    _lib.check(lib.hvc_set_seed_counter(counter.data_ptr()), "hvc_set_seed_counter")
    try:
        y = HF.linear(x, w, None, cdt=torch.float32, out_dtype=torch.float32, p_drop=0.5, seed=31)
        y.sum().backward()                                                        # backward: autograd worker thread
        keep = (y.detach() != 0).float()
        assert 0.4 < keep.mean().item() < 0.6
        want = (keep / 0.5) @ w
        assert ((x.grad - want).abs().max() / want.abs().max()).item() < F32_TOL
        y0 = HF.linear(x.detach(), w, None, cdt=torch.float32, out_dtype=torch.float32, p_drop=0.5, seed=31)
        _lib.check(lib.hvc_set_seed_counter(None), "hvc_set_seed_counter")
        y1 = HF.linear(x.detach(), w, None, cdt=torch.float32, out_dtype=torch.float32, p_drop=0.5, seed=31)
        assert torch.equal(y0, y.detach()) and not torch.equal(y1 != 0, y0 != 0)          # the counter really moved the mask
    finally:
        _lib.check(lib.hvc_set_seed_counter(None), "hvc_set_seed_counter")


@pytest.mark.parametrize("M,N,K", [(1, 1, 8), (5, 7, 24), (130, 257, 72), (256, 128, 64), (4100, 24, 40), (8192, 64, 32), (5000, 33, 100)])
@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_layouts_and_ragged_shapes(M, N, K, akm, bkm, dtype):
    from hvc import ops
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    Bm = torch.randn(N, K, generator=g).to(dtype)
    ref = A.double() @ Bm.double().t()
    a_in = (A.t().contiguous() if akm else A).to(dev())
    b_in = (Bm.t().contiguous() if bkm else Bm).to(dev())
    C = ops.gemm(a_in, b_in, a_kmajor=akm, b_kmajor=bkm, out_dtype=torch.float32)
    tol = 1e-4 if dtype == torch.float32 else 3e-3
    assert ((C.cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-12)).item() < tol


@pytest.mark.parametrize("M,N,K", [(768, 256, 65536), (256, 3456, 16384), (384, 200, 16448), (128, 128, 640), (1024, 256, 4100), (64, 1728, 40000)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_weight_gradient_shapes(M, N, K, dtype):
    """dW = dy^T x (both operands k-major, fp32 out) on shapes whose split-K slicing is uneven: tile counts that do not divide the
    512 resident workgroup slots (768 x 256: 12 tiles -> 42 slices of 25 k-tiles would leave an EMPTY 42nd slice - the launcher
    trims it; an unguarded fast path once read beyond K there), ragged K, ragged N, and the few-row tiles.  Twice: bit-equal
    (the slice order is fixed)."""
    from hvc import ops
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = torch.randn(K, M, generator=g).to(dtype)
    Bm = torch.randn(K, N, generator=g).to(dtype)
    ref = A.double().t() @ Bm.double()
    a_in, b_in = A.to(dev()), Bm.to(dev())
    C1 = ops.gemm(a_in, b_in, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    C2 = ops.gemm(a_in, b_in, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    assert torch.equal(C1, C2)
    tol = 1e-4 if dtype == torch.float32 else 3e-3
    assert ((C1.cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-12)).item() < tol


@pytest.mark.parametrize("M,N", [(100000, 1), (5000, 3), (777, 33), (4096, 200), (64, 300), (70000, 256), (33333, 32), (1, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_bias_gradient_shapes(M, N, dtype):
    """Column sums (bias gradients: hvc_colsum) incl. the ragged column counts of the cascade's 1-channel output layer
    (model_progressive.py:123: N = 1, a column of 16.7 M values at 256^3) and of odd channel counts; twice: bit-equal."""
    from hvc import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g).to(dtype).to(dev())
    a, b = ops.colsum(x), ops.colsum(x)
    assert torch.equal(a, b)
    ref = x.double().sum(dim=0).cpu()
    scale = x.double().abs().sum(dim=0).cpu() + 1e-9
    assert ((a.cpu().double() - ref).abs() / scale).max().item() < 1e-5


@pytest.mark.parametrize("M,N,K", [(2, 1536, 1024), (1, 7, 260), (8, 130, 512), (4, 512, 256)])
def test_gemm_few_rows_fp32_conditioning_linears(M, N, K):
    """M <= 8 fp32 rows (AdaLNModulation.linear, vit_components.py:131-147; time / context MLPs of diagnostic_losses.py:99-103,
    132): the streaming row-vector path of hvc_gemm, with bias and alpha, against fp64."""
    from hvc import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ref = 0.5 * (x.double() @ w.double().t()) + b.double()
    y = ops.gemm(x.to(dev()), w.to(dev()), bias=b.to(dev()), alpha=0.5)
    assert y.shape == (M, N) and ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


@pytest.mark.parametrize("seed", range(10))
def test_gemm_fused_epilogues_random(seed):
    """hvc_gemm's fused epilogue against fp64: alpha, bias, exact-erf GELU with the pre-activation saved (block MLP fc1,
    hybrid_vit_backbone.py:74-81), GELU' multiply (its backward), z-save + AdaLN gate + fp32 residual (the gated residual
    adds of :123 / :139), row-broadcast residual (pos_embed, :258), on ragged shapes and both operand dtypes."""
    import torch.nn.functional as F
    from hvc import ops
    rng = torch.Generator().manual_seed(4000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    nb = ri(1, 3)
    rows = ri(1, 150)
    M, N, K = nb * rows, ri(1, 40) * 8 if seed % 2 else ri(1, 300), ri(1, 40) * 8
    dtype = (torch.float32, torch.bfloat16)[(seed // 2) % 2]
    A = torch.randn(M, K, generator=rng).to(dtype)
    W = torch.randn(N, K, generator=rng).to(dtype) / K ** 0.5
    bias = torch.randn(N, generator=rng)
    gate = torch.randn(nb, N, generator=rng)
    res = torch.randn(M, N, generator=rng)
    pos = torch.randn(rows, N, generator=rng)
    pre_ref = 0.5 * (A.double() @ W.double().t()) + bias.double()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / max(b.abs().max().item(), 1e-6)).item()
    d = lambda t: t.to(dev())
    mode = seed % 5
    if mode == 0:        # bias + GELU, pre-activation saved
        aux = torch.empty(M, N, dtype=dtype, device=dev())
        y = ops.gemm(d(A), d(W), alpha=0.5, bias=d(bias), act=ops.ACT_GELU, aux=aux)
        assert rel(aux, pre_ref) < tol and rel(y, F.gelu(pre_ref)) < tol
    elif mode == 1:      # GELU' multiply with a saved pre-activation
        pre = torch.randn(M, N, generator=rng).to(dtype)
        xr = pre.double().requires_grad_(True)
        F.gelu(xr).sum().backward()
        y = ops.gemm(d(A), d(W), alpha=0.5, act=ops.ACT_GELU_GRAD, aux=d(pre))
        assert rel(y, (0.5 * (A.double() @ W.double().t())) * xr.grad) < tol
    elif mode == 2:      # z-save, gate, residual -> fp32 residual stream
        z = torch.empty(M, N, dtype=dtype, device=dev())
        y = ops.gemm(d(A), d(W), alpha=0.5, bias=d(bias), zsave=z, gate=d(gate), residual=d(res), rows_per_batch=rows,
                     out_dtype=torch.float32)
        assert rel(z, pre_ref) < tol
        assert rel(y, res.double() + pre_ref * gate.double().repeat_interleave(rows, 0)) < tol
    elif mode == 3:      # row-broadcast residual (pos_embed added to every sample's tokens)
        y = ops.gemm(d(A), d(W), alpha=0.5, bias=d(bias), residual=d(pos), residual_rows=rows, out_dtype=torch.float32)
        assert rel(y, pre_ref + pos.double().repeat(nb, 1)) < tol
    else:                # k-major operands with bias (dx / dW forms) into fp32
        y = ops.gemm(d(A.t().contiguous()), d(W.t().contiguous()), a_kmajor=True, b_kmajor=True, alpha=0.5, bias=d(bias),
                     out_dtype=torch.float32)
        assert rel(y, pre_ref) < tol


def test_gemm_output_dropout_is_seeded_and_consistent_with_branch_bwd():
    from hvc import ops
    M, N, K = 256, 96, 64
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(M, K, generator=g).to(dev()), torch.randn(N, K, generator=g).to(dev())
    y0 = ops.gemm(a, b)
    y1 = ops.gemm(a, b, p_drop=0.25, seed=42)
    y2 = ops.gemm(a, b, p_drop=0.25, seed=42)
    assert torch.equal(y1, y2)
    kept = y1 != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.02
    assert torch.allclose(y1[kept], y0[kept] / 0.75, rtol=1e-5, atol=1e-6)
    dy = torch.randn(M, N, generator=g).to(dev())
    dz, _, db = ops.branch_bwd(dy, None, None, out_dtype=torch.float32, p_drop=0.25, seed=42)
    assert torch.allclose(dz, torch.where(kept, dy / 0.75, torch.zeros_like(dy)), rtol=1e-6, atol=1e-7)
    assert torch.allclose(db, dz.sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C", [32, 30, 256, 1024])
def test_layernorm_kernel_vs_oracle(C):
    import torch.nn.functional as F
    from hvc import ops
    B, N = 2, 19
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B * N, C, generator=g) * 3 + 1
    gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
    sc, sh = torch.randn(B, C, generator=g) * 0.2, torch.randn(B, C, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (C,), gam, bet, 1e-5).view(B, N, C) * (1 + sc[:, None]) + sh[:, None]
    dy = torch.randn(B * N, C, generator=g)
    y_ref.reshape(B * N, C).backward(dy)
    d = dev()
    y, mean, rstd = ops.layernorm_fwd(x.to(d), gam.to(d), bet.to(d), sc.to(d), sh.to(d), rows_per_batch=N)
    assert torch.allclose(y.cpu(), y_ref.reshape(B * N, C).detach(), rtol=1e-4, atol=1e-4)
    dx, dg, db, dsc, dsh = ops.layernorm_bwd(dy.to(d), x.to(d), gam.to(d), bet.to(d), sc.to(d), mean, rstd, rows_per_batch=N)
    assert torch.allclose(dx.cpu(), xr.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("seed", range(10))
def test_layernorm_and_row_kernels_random(seed):
    """LayerNorm (+AdaLN modulate) forward / backward with every output (dx incl. the residual-stream gradient, dgamma,
    dbeta, dscale, dshift), the residual-branch backward (dz, dgate, dbias) and the column sum, on random ragged shapes:
    both the 4-values-per-lane (C <= 256) and the wide instantiation, vector and scalar column paths."""
    import torch.nn.functional as F
    from hvc import ops
    rng = torch.Generator().manual_seed(7000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    nb, rows = ri(1, 3), ri(1, 200)
    C = (ri(1, 64) * 4, ri(1, 256), ri(65, 256) * 4, 256, 1024)[seed % 5]
    mod = seed % 2 == 0
    dy_dtype = (torch.float32, torch.bfloat16)[(seed // 5) % 2]
    M = nb * rows
    x = (torch.randn(M, C, generator=rng) * 2 + 0.5).double().requires_grad_(True)
    gam = torch.randn(C, generator=rng).double().requires_grad_(True)
    bet = torch.randn(C, generator=rng).double().requires_grad_(True)
    sc = (torch.randn(nb, C, generator=rng) * 0.3).double().requires_grad_(True)
    sh = torch.randn(nb, C, generator=rng).double().requires_grad_(True)
    y = F.layer_norm(x, (C,), gam, bet, 1e-5)
    if mod:
        y = y.view(nb, rows, C) * (1 + sc[:, None]) + sh[:, None]
    dy = torch.randn(M, C, generator=rng).to(dy_dtype)
    dres = torch.randn(M, C, generator=rng)
    y.reshape(M, C).backward(dy.double())
    d = lambda t: None if t is None else t.detach().float().to(dev())
    yk, mean, rstd = ops.layernorm_fwd(d(x), d(gam), d(bet), d(sc) if mod else None, d(sh) if mod else None, rows_per_batch=rows)
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / max(b.abs().max().item(), 1e-3)).item()
    assert rel(yk, y.reshape(M, C).detach()) < 1e-5
    outs = ops.layernorm_bwd(dy.to(dev()), d(x), d(gam), d(bet), d(sc) if mod else None, mean, rstd, dres=dres.to(dev()), rows_per_batch=rows)
    tol = 2e-5 if dy_dtype == torch.float32 else 2e-5      # dy is the same rounded tensor on both sides
    assert rel(outs[0], x.grad + dres.double()) < tol
    assert rel(outs[1], gam.grad) < 1e-4 and rel(outs[2], bet.grad) < 1e-4
    if mod:
        assert rel(outs[3], sc.grad) < 1e-4 and rel(outs[4], sh.grad) < 1e-4
    else:
        assert outs[3] is None and outs[4] is None
    # residual-branch backward and column sum
    N = C
    g2 = torch.randn(M, N, generator=rng)
    z = torch.randn(M, N, generator=rng).to(dy_dtype)
    gate = torch.randn(nb, N, generator=rng)
    dz, dgate, dbias = ops.branch_bwd(g2.to(dev()), z.to(dev()), gate.to(dev()), rows_per_batch=rows, out_dtype=dy_dtype)
    v = g2.double() * gate.double().repeat_interleave(rows, 0)
    assert rel(dz, v) < (1e-6 if dy_dtype == torch.float32 else 1e-2)
    assert rel(dgate, (g2.double() * z.double()).view(nb, rows, N).sum(1)) < 1e-4
    assert rel(dbias, v.sum(0)) < 1e-4
    assert rel(ops.colsum(z.to(dev())), z.double().sum(0)) < 1e-4


@pytest.mark.parametrize("seed", range(8))
def test_groupnorm_and_batchnorm_pool_kernels_random(seed):
    """GroupNorm + SiLU / GELU (voxel stem hybrid_vit_backbone.py:195-210, cascade glue model_progressive.py:37-51, 169-174) and
    BatchNorm2d + ReLU + MaxPool2d (X-ray stem diagnostic_losses.py:82-96) on channels-last tensors of random shape
    against ATen: outputs, running statistics, input and parameter gradients, train and eval mode."""
    import torch.nn.functional as F
    from hvc import functional as HF, ops
    rng = torch.Generator().manual_seed(9000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / max(b.abs().max().item(), 1e-3)).item()
    # ---- GroupNorm + activation on (B, D, H, W, C)
    C = (8, 16, 32, 64, 128, 256, 64, 32)[seed]
    G = (8, 8, 8, 8, 16, 32, 16, 32)[seed]            # incl. one channel per group
    B, sp = ri(1, 3), (ri(1, 5), ri(1, 6), ri(1, 7))
    act = seed % 2
    x = torch.randn(B, *sp, C, generator=rng) * 1.5 + 0.3
    gam, bet = torch.randn(C, generator=rng), torch.randn(C, generator=rng)
    dy = torch.randn(B, *sp, C, generator=rng)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gam, bet))
    yn = F.group_norm(xr.permute(0, 4, 1, 2, 3), G, gr, br, 1e-5)
    yr = (F.silu(yn) if act == 0 else F.gelu(yn)).permute(0, 2, 3, 4, 1)
    yr.backward(dy.double())
    xg, gg, bg = (t.to(dev()).requires_grad_(True) for t in (x, gam, bet))
    y = HF.GroupNormSiluFn.apply(xg, gg, bg, G, 1e-5, act)
    y.backward(dy.to(dev()))
    assert rel(y.detach(), yr.detach()) < 2e-5 and rel(xg.grad, xr.grad) < 1e-4
    assert rel(gg.grad, gr.grad) < 1e-4 and rel(bg.grad, br.grad) < 1e-4
    # ---- BatchNorm + ReLU + MaxPool on (N, H, W, C)
    Cb = (8, 64, 128, 16, 64, 512, 8, 128)[seed]
    pool = (None, (3, 2, 1), (2, 2, 0), (3, 2, 1), None, None, (2, 2, 0), (3, 2, 1))[seed]
    N, Hh, Ww = ri(1, 4), ri(2, 17), ri(2, 19)
    training = seed % 3 != 0
    x = torch.randn(N, Hh, Ww, Cb, generator=rng) * 2 - 0.4
    gam, bet = torch.rand(Cb, generator=rng) + 0.5, torch.randn(Cb, generator=rng) * 0.5
    rm, rv = torch.randn(Cb, generator=rng) * 0.1, torch.rand(Cb, generator=rng) + 0.5
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gam, bet))
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    yb = F.relu(F.batch_norm(xr.permute(0, 3, 1, 2), rm_r, rv_r, gr, br, training, 0.1, 1e-5))
    if pool:
        yb = F.max_pool2d(yb, pool[0], pool[1], pool[2])
    yb = yb.permute(0, 2, 3, 1)
    dy = torch.randn(*yb.shape, generator=rng)
    yb.backward(dy.double())
    xg, gg, bg = (t.to(dev()).requires_grad_(True) for t in (x, gam, bet))
    rm_g, rv_g = rm.to(dev()), rv.to(dev())
    y = HF.BnReluPoolFn.apply(xg, gg, bg, rm_g, rv_g, pool, training, 1e-5, 0.1)
    y.backward(dy.to(dev()))
    assert y.shape == yb.shape and rel(y.detach(), yb.detach()) < 2e-5
    assert rel(rm_g, rm_r) < 1e-5 and rel(rv_g, rv_r) < 1e-5
    # ReLU / max ties are measure-zero for random data: gradients must agree element-wise
    assert rel(xg.grad, xr.grad) < 1e-4 and rel(gg.grad, gr.grad) < 1e-4 and rel(bg.grad, br.grad) < 1e-4


def test_layernorm_rejects_wide_rows():
    from hvc import ops
    x = torch.randn(4, 2048, device=dev())
    with pytest.raises(RuntimeError, match="C <= 1024"):
        ops.layernorm_fwd(x, torch.ones(2048, device=dev()), torch.zeros(2048, device=dev()))


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (2, 8, 6, 4), (1, 9, 7, 5), (2, 64, 64, 64), (1, 3, 5, 130)])
def test_drr_kernels_vs_oracle(shape):
    from hvc import functional as HF
    from oracle import hvc_oracle as O
    g = torch.Generator().manual_seed(sum(shape))
    vol = torch.rand(*shape, generator=g) * 2 - 1
    for angle in (0, 90):
        vr = vol.clone().requires_grad_(True)
        ref = O.drr_render(vr, angle)
        w = torch.randn(ref.shape, generator=g)
        (ref * w).sum().backward()
        vd = vol.to(dev()).requires_grad_(True)
        out = HF.drr_project(vd, 2 if angle == 90 else 0, exp_mode=True, mu=0.3, clamp_min=1e-6, transpose_out=angle == 90)
        (out * w.to(dev())).sum().backward()
        assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(vd.grad.cpu(), vr.grad, rtol=1e-4, atol=1e-6)
    # mean projections of DRRReprojectionLoss (no exp, no clamp, no transpose)
    v5 = vol.unsqueeze(1)
    ap = HF.drr_project(vol.to(dev()), 0, exp_mode=False, out_scale=1.0 / shape[1])
    lat = HF.drr_project(vol.to(dev()), 2, exp_mode=False, out_scale=1.0 / shape[3])
    assert torch.allclose(ap.cpu(), v5.mean(dim=2).squeeze(1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(lat.cpu(), v5.mean(dim=4).squeeze(1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 256, 256, 256), (2, 96, 384, 512)])
def test_drr_lateral_tiled_kernel_vs_oracle(shape, dtype):
    """The ray sum along W at full size (>= 65536 rows of >= 256 voxels: drr_fwd_w_tiled_kernel - eight rows in flight per
    wavefront, one butterfly reduction per eight rows, stores through LDS) in both output layouts: DRRRenderer's lateral view
    (exp, clamp, transposed: diagnostic_losses.py:45-63) and DRRReprojectionLoss's mean (loss_multiscale.py:260-267), against the
    oracle evaluated on the same (bf16-rounded) volume."""
    from hvc import ops
    from oracle import hvc_oracle as O
    g = torch.Generator().manual_seed(sum(shape))
    vol = (torch.rand(*shape, generator=g) * 2 - 1).to(dtype)
    vd = vol.to(dev())
    ref = O.drr_render(vol.double(), 90)                                     # (B, H, D)
    out = ops.drr_fwd(vd, 2, exp_mode=True, mu=0.3, clamp_min=1e-6, transpose_out=True)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert out.shape == ref.shape
    assert ((out.double().cpu() - ref).abs().max() / ref.abs().max()).item() < tol
    lat = ops.drr_fwd(vd, 2, exp_mode=False, out_scale=1.0 / shape[3])
    refm = vol.double().mean(dim=3)
    assert ((lat.double().cpu() - refm).abs().max() / refm.abs().max().clamp_min(1e-3)).item() < (1e-5 if dtype == torch.float32 else 2e-2)


def test_drr_full_size_linearity_and_checksum():
    """256^3 (BASELINE full size): mean projection is linear, and sum over the projection equals the
    scaled sum over the volume (size-independent properties; no CPU oracle run needed)."""
    from hvc import functional as HF
    g = torch.Generator().manual_seed(11)
    a = torch.rand(1, 256, 256, 256, generator=g).to(dev())
    b = torch.rand(1, 256, 256, 256, generator=g).to(dev())
    for axis in (0, 2):
        pa = HF.drr_project(a, axis, exp_mode=False, out_scale=1 / 256)
        pb = HF.drr_project(b, axis, exp_mode=False, out_scale=1 / 256)
        pab = HF.drr_project(a + 2 * b, axis, exp_mode=False, out_scale=1 / 256)
        assert torch.allclose(pab, pa + 2 * pb, rtol=1e-5, atol=1e-5)
        assert abs(pa.double().sum().item() * 256 - a.double().sum().item()) < 1e-3 * a.numel() ** 0.5 + 1.0


def test_direct_model_full_size_64_vs_oracle_and_psnr():
    """BASELINE config #1/#2 geometry (64^3, 2-view 512^2): identical weights and synthetic inputs
    through the CPU oracle and the HIP path; fp32 within 1e-3 rel, PSNR within 0.1 dB (fp32 and bf16)."""
    from direct_regression.model_direct import DirectCTRegression
    from hvc import synthetic
    from oracle import hvc_oracle as O
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(64, 64, 64)).eval()
    gen = torch.Generator().manual_seed(21)
    with torch.no_grad():      # AdaLN is zero-initialised in the reference: make the gated branches visible
        for blk in m.vit_backbone.blocks:
            blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)
            blk.adaln.linear.bias.copy_(torch.randn(blk.adaln.linear.bias.shape, generator=gen) * 0.02)
    xr, ct = synthetic.sample(0, (64, 64, 64), 512)
    xr, ct = xr[None], ct[None]
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = O.direct_ct_regression(xr, P)
    m.to(dev())
    with torch.no_grad():
        y32 = m(xr.to(dev())).cpu()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = m(xr.to(dev())).float().cpu()
    assert ((y32 - ref).abs().max() / ref.abs().max()).item() < F32_TOL
    p_ref, p32, p16 = O.psnr(ref, ct), O.psnr(y32, ct), O.psnr(y16, ct)
    assert abs(p32 - p_ref) < 0.1 and abs(p16 - p_ref) < 0.1, (p_ref, p32, p16)


def test_attention_full_size_128_properties():
    """BASELINE config #3 attention shape (A2-fix: 32^3 = 32768 tokens, h x d = 4 x 64), where no N x N oracle fits
    in memory: size-independent properties of softmax attention and of its backward.
      * rows of P sum to one: V = 1 gives O = 1, and then dQ = dK = 0 exactly in exact arithmetic;
      * O is linear in V; <dO, O> = <dV, V> (adjoint of the V -> O map);
      * probed query rows (first, last, tile seams) of O, LSE and dQ agree with a direct fp64 evaluation."""
    from hvc import ops
    B, H, N, D = 1, 4, 32768, 64
    g = torch.Generator().manual_seed(128)
    q, k, v = (torch.randn(B, N, H, D, generator=g).to(dev()) for _ in range(3))
    do = torch.randn(B, N, H, D, generator=g).to(dev())
    scale = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, scale, 0.0, 0)
    ones = torch.ones_like(v)
    o1, lse1 = ops.attention_fwd(q, k, ones, scale, 0.0, 0)
    assert (o1 - 1).abs().max().item() < 2e-5 and torch.equal(lse, lse1)
    dq1, dk1, dv1 = ops.attention_bwd(q, k, ones, o1, do, lse1, scale, 0.0, 0)
    assert dq1.abs().max().item() < 1e-4 * do.abs().max().item() and dk1.abs().max().item() < 1e-3 * do.abs().max().item()
    v2 = torch.randn(B, N, H, D, generator=g).to(dev())
    o2, _ = ops.attention_fwd(q, k, v2, scale, 0.0, 0)
    o12, _ = ops.attention_fwd(q, k, 0.5 * v - 2.0 * v2, scale, 0.0, 0)
    assert (o12 - (0.5 * o - 2.0 * o2)).abs().max().item() < 1e-4 * max(o.abs().max().item(), o2.abs().max().item())
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, scale, 0.0, 0)
    lhs, rhs = (do.double() * o.double()).sum().item(), (dv.double() * v.double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * (abs(lhs) + abs(rhs) + 1.0)
    # probed rows against fp64: 8 query rows of one head for O / dQ, 8 key rows for dK / dV need all queries -> use O / dQ rows
    b, hh = 0, 2
    rows = torch.tensor([0, 1, 31, 32, 4095, 16384, 32766, 32767])
    kd, vd = k[b, :, hh].double(), v[b, :, hh].double()
    s_ = (q[b, rows, hh].double() @ kd.t()) * scale
    p_ = torch.softmax(s_, dim=-1)
    o_ref = p_ @ vd
    assert (o[b, rows, hh].double() - o_ref).abs().max().item() < 1e-3 * o_ref.abs().max().item()
    lse_ref = torch.logsumexp(s_, dim=-1)
    assert (lse[b, hh, rows].double() - lse_ref).abs().max().item() < 1e-4
    dp_ = do[b, rows, hh].double() @ vd.t()
    ds_ = p_ * (dp_ - (p_ * dp_).sum(-1, keepdim=True))
    dq_ref = ds_ @ kd * scale
    assert (dq[b, rows, hh].double() - dq_ref).abs().max().item() < 1e-3 * dq_ref.abs().max().item()


def test_direct_model_full_size_128_a2fix_geometry_and_bf16_psnr():
    """BASELINE config #3 geometry (128^3; the reference raises here): the A2-fix sizes the token grid from the
    stem's real output (32^3 = 32768 tokens); fp32 (split-bf16 MFMA) and bf16 runs of the same weights agree, and
    their PSNR against the synthetic target differs by < 0.1 dB; the run is reproducible bit for bit."""
    from direct_regression.model_direct import DirectCTRegression
    from hvc import synthetic
    from oracle import hvc_oracle as O
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(128, 128, 128)).eval()
    assert tuple(m.vit_backbone.downsampled_size) == (32, 32, 32) and m.vit_backbone.pos_embed.shape == (1, 32768, 256)
    gen = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for blk in m.vit_backbone.blocks:
            blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)
    xr, ct = synthetic.sample(1, (128, 128, 128), 512)
    m.to(dev())
    with torch.no_grad():
        y32 = m(xr[None].to(dev()))
        y32b = m(xr[None].to(dev()))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = m(xr[None].to(dev())).float()
    assert y32.shape == (1, 1, 128, 128, 128) and torch.isfinite(y32).all() and torch.equal(y32, y32b)
    assert ((y16 - y32).norm() / y32.norm()).item() < 2e-2
    assert abs(O.psnr(y32.cpu(), ct[None]) - O.psnr(y16.cpu(), ct[None])) < 0.1


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_cascade_refiners_vs_golden(golden, mode):
    """Shared multi-scale X-ray encoder + stage-2 / stage-3 refiners (reduced sizes) chained as
    ProgressiveCascadeModel.forward does, against the reference's outputs and gradients."""
    from direct_regression.progressive_cascade.model_progressive import (MultiScaleXrayEncoder, Stage2Refiner128,
                                                                         Stage3Refiner256)
    g = golden("cascade_small")
    enc = _load(MultiScaleXrayEncoder(img_size=64, in_channels=1, base_dim=32, num_views=2).eval(), g.group("enc_params"))
    s2 = _load(Stage2Refiner128(volume_size=(32, 32, 32), voxel_dim=32, vit_depth=1, num_heads=1, xray_feature_dim=32).eval(),
               g.group("s2_params"))
    s3 = _load(Stage3Refiner256(volume_size=(64, 64, 64), voxel_dim=32, vit_depth=1, num_heads=1, xray_feature_dim=32,
                                use_gradient_checkpointing=False).eval(), g.group("s3_params"))
    xr = g.t("xrays").to(dev())
    v16 = g.t("v16").to(dev()).requires_grad_(True)

    def run():
        f1, _, _ = enc(xr, stage=1)
        f2, cond2, _ = enc(xr, stage=2)
        v32 = s2(v16, f2, cond2)
        f3, cond3, _ = enc(xr, stage=3)
        return f1, f2, v32, s3(v32, f3, cond3)
    f1, f2, v32, v64 = _run(mode, run)
    tol, met = _tol(mode), _metric(mode)
    g.check("", "feats1", f1, tol, metric=met)
    g.check("", "feats2", f2, tol, metric=met)
    g.check("", "v32", v32, tol, metric=met)
    g.check("", "v64", v64, tol, metric=met)
    ((v32.float() * g.t("w2").to(dev())).sum() + (v64.float() * g.t("w3").to(dev())).sum() + f1.float().sum() * 0.1).backward()
    g.check("", "dv16", v16.grad, _gtol(mode), metric=met)
    for pre, m in (("enc", enc), ("s2", s2), ("s3", s3)):
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            if k in _zero_grad_biases(m):
                continue   # conv bias ahead of a one-channel-per-group GroupNorm: exactly-zero gradient, rounding noise only
            # encoder parameters sit behind the stem's ReLU / max-pool routing; on this fixture (eval mode) the HIP and the
            # reference routings agree on every live window in fp32 mode, so they hold the flat 1e-3 like everything else
            g.check(f"{pre}_pgrad", k, p.grad, _gtol(mode), metric=met)


def test_stage3_gradient_checkpointing_replays_dropout_masks(monkeypatch):
    """BASELINE config #5 trains stage 3 with gradient checkpointing (model_progressive.py:296-305).  The recomputation
    must redraw the same counter-based dropout seeds (torch.utils.checkpoint restores the CPU generator they come
    from): gradients with and without checkpointing are then bitwise equal in train mode with dropout 0.1."""
    from direct_regression.progressive_cascade.model_progressive import Stage3Refiner256
    from hvc import functional as HF
    monkeypatch.setattr(HF, "CHECKPOINT_POLICY", "on")      # the default "auto" skips the recomputation while HBM has room: force the path under test
    torch.manual_seed(5)
    ref = Stage3Refiner256(volume_size=(32, 32, 32), voxel_dim=64, vit_depth=2, num_heads=2, xray_feature_dim=32,
                           use_gradient_checkpointing=False).to(dev()).train()
    for blk in ref.vit_refiner.blocks:                      # AdaLN is zero-initialised: give the gated branches some weight
        torch.nn.init.normal_(blk.adaln.linear.weight, std=0.02)
    ck = Stage3Refiner256(volume_size=(32, 32, 32), voxel_dim=64, vit_depth=2, num_heads=2, xray_feature_dim=32,
                          use_gradient_checkpointing=True).to(dev()).train()
    ck.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(6)
    v = torch.randn(1, 1, 16, 16, 16, generator=g).to(dev())
    feats = torch.randn(1, 32, 8, 8, generator=g).to(dev())
    cond = torch.randn(1, 1024, generator=g).to(dev())
    outs = []
    for m in (ref, ck):
        torch.manual_seed(77)                               # same dropout seed stream for both runs
        vin = v.clone().requires_grad_(True)
        out = m(vin, feats, cond)
        out.square().mean().backward()
        outs.append((out.detach(), vin.grad, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2].keys() == outs[1][2].keys() and len(outs[0][2]) > 20
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


def test_drr_reprojection_loss_vs_golden(golden):
    from direct_regression.progressive_cascade.loss_multiscale import DRRReprojectionLoss, compute_psnr
    g = golden("drr")
    vol = g.t("vol").to(dev()).requires_grad_(True)
    rl = DRRReprojectionLoss(img_size=16)
    loss = rl(vol, g.t("xr2").to(dev()))
    assert abs(loss.item() - 0.516497) < 1e-5                    # SURVEY.md §9 known answer
    loss.backward()
    g.check("", "reproj_dvol", vol.grad, 1e-4)
    g.check("", "reproj_ap", rl.generate_drr(vol.detach(), 0), 1e-5)
    g.check("", "reproj_lat", rl.generate_drr(vol.detach(), 90), 1e-5)
    assert compute_psnr(vol.detach(), vol.detach()) == float("inf")


def test_multiscale_loss_stage1_matches_direct_loss():
    from direct_regression.model_direct import DirectRegressionLoss
    from direct_regression.progressive_cascade.loss_multiscale import MultiScaleLoss, compute_ssim_metric
    from oracle import hvc_oracle as O
    g = torch.Generator().manual_seed(9)
    p = (torch.rand(2, 1, 16, 16, 16, generator=g) * 2 - 1)
    t = (torch.rand(2, 1, 16, 16, 16, generator=g) * 2 - 1)
    ref = O.direct_regression_loss(p, t)
    a = MultiScaleLoss()(p.to(dev()), t.to(dev()), stage=1)
    b = DirectRegressionLoss()(p.to(dev()), t.to(dev()))
    for k in ("total_loss", "l1_loss", "ssim_loss"):
        assert abs(a[k].item() - ref[k].item()) < 1e-5 and abs(b[k].item() - ref[k].item()) < 1e-5
    assert abs(compute_ssim_metric(p.to(dev()), t.to(dev())) - (1 - ref["ssim_loss"].item())) < 1e-5
    d = MultiScaleLoss()(p.to(dev()).requires_grad_(True), t.to(dev()), stage=3,
                         input_xrays=torch.rand(2, 2, 1, 512, 512, generator=g).to(dev()))
    assert set(d) == {"total_loss", "l1_loss", "ssim_loss", "vgg_loss", "tv_loss", "freq_loss", "drr_loss"}
    d["total_loss"].backward()


def test_ssim_l1_loss_zero_crossing_numerators():
    """SSIM numerators N1 = 2 mu_p mu_t + C1 and N2 = 2 cov + C2 cross zero for anti-correlated windows (early
    training does this at 128^3); dS/dN must stay finite there, as in the reference's autograd of
    model_direct.py:88-116 which never divides by a numerator."""
    from direct_regression.model_direct import DirectRegressionLoss
    from oracle import hvc_oracle as O
    gen = torch.Generator().manual_seed(11)
    t = torch.rand(1, 1, 24, 24, 24, generator=gen) * 2 - 1
    ramp = torch.linspace(0.0, 0.01, 24).view(1, 1, 1, 1, 24)
    p = -ramp * t                                   # covariance sweeps through -C2/2 along W
    p[..., :12, :, :] = 0.01                        # constant window means: 2 mu_p mu_t + C1 == 0 in the interior
    t[..., :12, :, :] = -0.005
    pr = p.clone().requires_grad_(True)
    ref = O.direct_regression_loss(pr, t)
    ref["total_loss"].backward()
    pg = p.to(dev()).requires_grad_(True)
    got = DirectRegressionLoss(1.0, 0.5)(pg, t.to(dev()))
    got["total_loss"].backward()
    assert torch.isfinite(pg.grad).all()
    for k in ("total_loss", "l1_loss", "ssim_loss"):
        assert abs(got[k].item() - ref[k].item()) < 1e-5
    err = (pg.grad.cpu() - pr.grad).norm() / pr.grad.norm()
    assert err < 1e-3, err                          # fp32, north_star tolerance


@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("shape", [((1, 1, 1), (4, 4, 4)), ((4, 3, 5), (16, 9, 20)), ((16, 16, 16), (64, 64, 64)), ((7, 6, 5), (7, 6, 5)),
                                   ((8, 8, 8), (4, 4, 4)), ((3, 3, 3), (10, 7, 1))])
def test_trilinear_resize_kernels_vs_aten_semantics(shape, align_corners):
    """F.interpolate(mode='trilinear') in both corner conventions (hybrid_vit_backbone.py:272 uses True;
    model_progressive.py:169 / train_progressive_4gpu.py:46-57 use False), up- and down-sampling, forward and adjoint."""
    import torch.nn.functional as F
    from hvc import ops
    (d, h, w), out = shape
    g = torch.Generator().manual_seed(d * 100 + h * 10 + w)
    x = torch.randn(2, d, h, w, generator=g)
    dy = torch.randn(2, *out, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr[:, None], size=out, mode="trilinear", align_corners=align_corners)[:, 0]
    ref.backward(dy)
    y = ops.trilinear_fwd(x.to(dev()), out, align_corners)
    assert (y.cpu() - ref.detach()).abs().max().item() < 1e-5
    for separable in (True, False):         # three 1-D adjoint passes / single-pass 3-D gather
        dx = ops.trilinear_bwd(dy.to(dev()), (d, h, w), align_corners, separable=separable)
        assert (dx.cpu() - xr.grad).abs().max().item() < 1e-4 * max(1.0, xr.grad.abs().max().item()), separable


@pytest.mark.timeout(600)
def test_progressive_trainer_stage_steps(tmp_path):
    """One optimisation step of stage 1 and of stage 2 through train_progressive_4gpu.build_stage / train_step: stage 2
    loads the stage-1 checkpoint, freezes stage 1 (no gradients, weights untouched) and updates stage 2 + its X-ray heads."""
    import copy, json, os, sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "hybrid-vit-cascade_amd", "direct_regression"))
    from progressive_cascade import train_progressive_4gpu as T
    from hvc import synthetic
    cfg = json.load(open(os.path.join(here, "hybrid-vit-cascade_amd", "direct_regression", "progressive_cascade", "config_progressive.json")))
    torch.manual_seed(0)
    xr, ct = synthetic.batch(0, 1, (128, 128, 128), 512)
    xr, ct = xr.to(dev()), ct.to(dev())
    model, crit, opt, _ = T.build_stage(cfg, 1, tmp_path, dev())
    model.train()
    before = copy.deepcopy(model.stage1.state_dict())
    losses = T.train_step(model, crit, opt, None, xr, ct, 1, cfg["training"]["gradient_clip"])
    assert set(losses) == {"total_loss", "l1_loss", "ssim_loss"} and torch.isfinite(losses["total_loss"])
    assert any(not torch.equal(v, before[k]) for k, v in model.stage1.state_dict().items() if v.dtype.is_floating_point)
    torch.save({"model_state_dict": model.state_dict()}, tmp_path / "stage1_best.pth")
    s1 = copy.deepcopy(model.stage1.state_dict())
    model2, crit2, opt2, _ = T.build_stage(cfg, 2, tmp_path, dev())
    model2.train()
    for k, v in model2.stage1.state_dict().items():
        assert torch.equal(v, s1[k]), k                                    # stage-1 checkpoint loaded
    assert all(not p.requires_grad for p in model2.stage1.parameters())
    w0 = model2.stage2.vit_refiner.blocks[0].mlp[0].weight.detach().clone()
    losses = T.train_step(model2, crit2, opt2, None, xr, ct, 2, cfg["training"]["gradient_clip"])
    assert {"total_loss", "l1_loss", "ssim_loss", "tv_loss", "freq_loss"} <= set(losses) and torch.isfinite(losses["total_loss"])
    assert all(p.grad is None for p in model2.stage1.parameters())
    bn_free = {k: v for k, v in model2.stage1.state_dict().items() if "running_" not in k and "num_batches" not in k}
    for k, v in bn_free.items():
        assert torch.equal(v, s1[k]), k                                    # frozen weights untouched
    assert not torch.equal(model2.stage2.vit_refiner.blocks[0].mlp[0].weight.detach(), w0)
    val = T.validate(model2, [{"drr_stacked": xr.cpu(), "ct_volume": ct.cpu()}], crit2, 0, 2, max_stage=2)
    assert {"total", "psnr", "ssim"} <= set(val) and val["psnr"] == val["psnr"]


def _ddp_gpu_worker(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "hybrid-vit-cascade_amd"))
    from direct_regression import train_direct_4gpu as T
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import synthetic
    T.setup_ddp(rank, world, backend="gloo", port=str(port))       # both ranks share cuda:0; gloo moves the buckets
    torch.cuda.set_device(0)
    cfg = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=64, vit_depth=1, num_heads=2, xray_feature_dim=32)
    torch.manual_seed(0)
    model = DirectCTRegression(**cfg).cuda(0).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    crit = DirectRegressionLoss()
    xr, ct = synthetic.batch(10 * rank, 2, cfg["volume_size"], cfg["xray_img_size"], device="cuda:0")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(xr), ct)["total_loss"]
    loss.backward()
    local = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    for m in model.modules():                     # undo the running-stat update of the probe pass
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    ddp = T.wrap_ddp(model, [0])
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3, weight_decay=0.01)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(ddp(xr), ct)["total_loss"]
    loss.backward()
    reduced = {k: p.grad.clone() for k, p in model.named_parameters()}
    T.train_step(ddp, crit, opt, None, xr, ct, gradient_clip=1.0)
    torch.save({"local": {k: v.cpu() for k, v in local.items()}, "reduced": {k: v.cpu() for k, v in reduced.items()},
                "params": {k: v.detach().cpu() for k, v in model.state_dict().items()}}, os.path.join(out_dir, f"rank{rank}.pt"))
    T.cleanup_ddp()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_on_the_hip_path(tmp_path):
    """Two processes, both on cuda:0, DDP over gloo: checks that the HIP autograd Functions cooperate with DDP's
    bucket hooks (gradient-as-bucket-view) -- reduced gradients are the mean of the ranks' local ones, parameters
    stay identical after an optimizer step.  (The production launch is one rank per GPU over RCCL.)"""
    import torch.multiprocessing as mp
    world, port = 2, 29533
    mp.spawn(_ddp_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    for k in r0["reduced"]:
        mean_local = (r0["local"][k] + r1["local"][k]) / 2
        assert torch.allclose(r0["reduced"][k], mean_local, rtol=2e-3, atol=1e-6), k
        assert torch.equal(r0["reduced"][k], r1["reduced"][k]), k
    for k in r0["params"]:
        if "running_" in k or "num_batches" in k:
            continue   # BatchNorm statistics are per rank, as in the reference (no SyncBN); DDP re-broadcasts them at the next forward
        assert torch.equal(r0["params"][k], r1["params"][k]), k


def _reducer_gpu_worker(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "hybrid-vit-cascade_amd"))
    from direct_regression import train_direct_4gpu as T
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import synthetic
    from hvc.reducer import BucketedGradReducer, broadcast_module_state
    T.setup_ddp(rank, world, backend="gloo", port=str(port))       # both ranks share cuda:0; gloo moves the buckets
    torch.cuda.set_device(0)
    cfg = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=64, vit_depth=2, num_heads=2, xray_feature_dim=32)
    torch.manual_seed(rank)                        # different initial weights per rank: the broadcast must align them
    model = DirectCTRegression(**cfg).cuda(0).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    broadcast_module_state(model)
    crit = DirectRegressionLoss()
    xr, ct = synthetic.batch(10 * rank, 2, cfg["volume_size"], cfg["xray_img_size"], device="cuda:0")
    params = [p for p in model.parameters() if p.requires_grad]
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    red = BucketedGradReducer(params, bucket_bytes=256 * 1024, check=True)
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01)
    local, reduced = [], []
    for it in range(3):
        # the rank's own gradient of this step, for the mean check (no reduction: plain autograd.grad)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(xr), ct)["total_loss"]
        g = torch.autograd.grad(loss, params, allow_unused=True)
        local.append({k: (None if t is None else t.detach().float().cpu()) for k, t in zip(names, g)})
        for m in model.modules():                 # the probe pass moved the BatchNorm running statistics: irrelevant to the gradients
            pass
        red.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(xr), ct)["total_loss"]
        loss.backward()
        red.finish()
        reduced.append({k: p.grad.detach().float().cpu().clone() for k, p in zip(names, params)})
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
    torch.save({"local": local, "reduced": reduced, "params": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "desc": red.describe(), "order_head": [names[i] for i in red.order[:3]]}, os.path.join(out_dir, f"rank{rank}.pt"))
    T.cleanup_ddp()


@pytest.mark.timeout(600)
def test_bucketed_reducer_two_ranks_on_the_hip_path(tmp_path):
    """hvc.reducer.BucketedGradReducer (what `bench.py --gpus N` runs eagerly: one autograd hook per bucket instead of DDP's one per
    parameter) with the HIP autograd Functions: two processes on cuda:0 over gloo, three optimizer steps in check mode - every step's
    reduced gradients are the mean of the ranks' local ones (discovery step and bucketed steps alike), replicas stay identical."""
    import torch.multiprocessing as mp
    world, port = 2, 29537
    mp.spawn(_reducer_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    assert r0["desc"]["buckets"] >= 2, r0["desc"]
    for step in range(3):
        for k in r0["reduced"][step]:
            a, b = r0["local"][step][k], r1["local"][step][k]
            if a is None:
                continue
            mean_local = (a + b) / 2
            assert torch.allclose(r0["reduced"][step][k], mean_local, rtol=2e-3, atol=1e-6), (step, k)
            assert torch.equal(r0["reduced"][step][k], r1["reduced"][step][k]), (step, k)
    for k in r0["params"]:
        if "running_" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["params"][k], r1["params"][k]), k


@pytest.fixture(params=["implicit", "im2col"])
def conv_path(request):
    """Runs a convolution test on both data paths: patches gathered inside the GEMM (C % 8 == 0) and im2col + GEMM + col2im."""
    from hvc import functional as HF
    old = HF.CONV_IMPLICIT
    HF.CONV_IMPLICIT = request.param == "implicit"
    yield request.param
    HF.CONV_IMPLICIT = old


@pytest.mark.parametrize("cfg", [(3, 2, 16, 24, 3, 1, 1, (10, 9, 8)), (3, 1, 8, 16, 3, 2, 1, (13, 8, 8)), (3, 2, 1, 8, 3, 2, 1, (12, 6, 6)),
                                 (3, 1, 32, 8, 1, 1, 0, (9, 4, 4))])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_slabbed_equals_whole_and_oracle(cfg, dtype, conv_path):
    """Convolution run slab by slab along D (large-volume path) == the single-shot path == F.conv3d of the oracle.
    (On the implicit path only the few-channel layers and the strided input gradients still have a matrix to slab.)"""
    import torch.nn.functional as F
    from hvc import functional as HF
    from hvc import ops
    dims, B, Cin, Cout, k, stride, pad, sp = cfg
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.to(dtype).double().requires_grad_(True) for t in (x, w, b))
    y_ref = F.conv3d(xr, wr, br, stride=stride, padding=pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.to(dtype).double())
    geom = ops.ConvGeometry(B, Cin, sp, (k,) * 3, stride, (pad,) * 3)
    results = []
    for cap in (HF.CONV_SLAB_BYTES, 4096):
        old = HF.CONV_SLAB_BYTES
        HF.CONV_SLAB_BYTES = cap
        try:
            xd = x.to(dev()).to(dtype).permute(0, 2, 3, 4, 1).contiguous().requires_grad_(True)
            wd = w.to(dev()).to(dtype).float().requires_grad_(True)
            bd = b.to(dev()).to(dtype).float().requires_grad_(True)
            y = HF.ConvFn.apply(xd, wd, bd, None, geom, dtype, dtype)
            (y.float() * dy.to(dev()).to(dtype).permute(0, 2, 3, 4, 1).float()).sum().backward()
            results.append((y.detach().permute(0, 4, 1, 2, 3).float().cpu(), xd.grad.permute(0, 4, 1, 2, 3).float().cpu(),
                            wd.grad.cpu(), bd.grad.cpu()))
        finally:
            HF.CONV_SLAB_BYTES = old
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    refs = (y_ref.detach(), xr.grad, wr.grad, br.grad)
    for got in results:
        for a, r in zip(got, refs):
            assert ((a.double() - r).abs().max() / (r.abs().max() + 1e-12)).item() < tol
    for a, c in zip(*results):       # slab-wise vs single shot: same products, only the dW summation order differs
        assert torch.allclose(a, c, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("seed", range(10))
def test_conv_layers_random_geometry(seed, conv_path):
    """nn.Conv2d / nn.Conv3d through hvc.stem.conv_channels_last (implicit GEMM, or im2col + MFMA GEMM, col2im; split-K wgrad) against ATen on
    random geometry: kernel 1 / 3 / 5 / 7, stride 1 / 2, asymmetric extents, channel counts on both im2col paths (C % 8 == 0 and
    few-channel), fp32 and bf16 - the stems of diagnostic_losses.py:82-96, hybrid_vit_backbone.py:195-210 and
    model_progressive.py:259-267 (incl. its 1x1x1 convolution)."""
    import torch.nn as nn
    from hvc import stem
    rng = torch.Generator().manual_seed(12000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    is3d = seed % 3 != 0
    k = (1, 3, 3, 7, 5, 3, 1, 3, 7, 3)[seed]
    stride = (1, 2, 1, 2, 1, 2, 1, 1, 2, 2)[seed]
    pad = k // 2 if seed % 4 else 0
    cin = (1, 8, 3, 1, 16, 32, 64, 2, 5, 24)[seed]
    cout = (7, 16, 32, 64, 9, 24, 1, 33, 8, 40)[seed]
    B = ri(1, 2)
    sp = tuple(ri(max(k, 2), 13) for _ in range(3 if is3d else 2))
    dtype = (torch.float32, torch.bfloat16)[(seed // 5) % 2]
    torch.manual_seed(seed)
    layer = (nn.Conv3d if is3d else nn.Conv2d)(cin, cout, k, stride=stride, padding=pad)
    x = torch.randn(B, cin, *sp, generator=rng)
    ref_layer = (nn.Conv3d if is3d else nn.Conv2d)(cin, cout, k, stride=stride, padding=pad).double()
    ref_layer.load_state_dict({n: (t.to(dtype).double() if dtype == torch.bfloat16 else t.double()) for n, t in layer.state_dict().items()})
    xr = x.to(dtype).double().requires_grad_(True)
    yr = ref_layer(xr)
    dy = torch.randn(yr.shape, generator=rng)
    yr.backward(dy.to(dtype).double())
    layer.to(dev())
    perm_in = (0, 2, 3, 4, 1) if is3d else (0, 2, 3, 1)
    h = x.to(dev()).to(dtype).permute(*perm_in).contiguous()
    if not is3d:
        h = h.unsqueeze(1)                                          # (B, 1, H, W, C): the 2-D convolutions run as depth-1 volumes
    h.requires_grad_(True)
    y = stem.conv_channels_last(h, layer, dtype)
    dyc = dy.to(dev()).to(dtype).permute(*perm_in)
    if not is3d:
        dyc = dyc.unsqueeze(1)
    (y.float() * dyc.float()).sum().backward()
    back = (0, 4, 1, 2, 3)
    got_y = y.detach().permute(*back).float().cpu()
    got_dx = h.grad.permute(*back).float().cpu()
    if not is3d:
        got_y, got_dx = got_y.squeeze(2), got_dx.squeeze(2)
    tol = 1e-4 if dtype == torch.float32 else 2.5e-2
    rel = lambda a, b: ((a.double() - b).abs().max() / (b.abs().max() + 1e-9)).item()
    assert got_y.shape == yr.shape and rel(got_y, yr.detach()) < tol
    assert rel(got_dx, xr.grad) < tol
    assert rel(layer.weight.grad.cpu(), ref_layer.weight.grad) < tol and rel(layer.bias.grad.cpu(), ref_layer.bias.grad) < tol


@pytest.mark.parametrize("cfg", [
    # is3d, B, Cin, Cout, k, stride, pad, spatial
    (True, 2, 64, 32, 3, 1, 1, (20, 18, 22)),      # the cascade's detail-enhancer layer shape (model_progressive.py:122), many row tiles
    (True, 1, 32, 64, 3, 1, 1, (17, 33, 9)),
    (True, 2, 96, 192, 3, 2, 1, (16, 16, 16)),     # stride-2 stem layer (hybrid_vit_backbone.py:195-210): dx stays on col2im
    (True, 1, 8, 40, 5, 1, 2, (11, 12, 13)),       # K = 1000: ragged last k-tile, N not a multiple of 8 on the dx side
    (False, 3, 64, 128, 3, 1, 1, (40, 36)),        # X-ray stem layer (diagnostic_losses.py:87)
    (False, 2, 128, 256, 3, 1, 1, (21, 19)),
    (True, 1, 16, 8, 1, 1, 0, (9, 10, 11)),        # 1x1x1
    (True, 1, 16, 24, 3, 1, 1, (12, 11, 10)),      # 24 output channels: ragged narrow tiles (256 x 32 forward / dx, 32 x 128 dW)
    (True, 1, 40, 56, 3, 1, 1, (9, 12, 13)),       # 256 x 64 and 64 x 128 tiles with ragged channel counts
    (True, 1, 8, 16, 5, 3, 2, (11, 10, 12)),       # strided input gradient as parity classes: k5 s3 (27 classes of 1..2 taps per axis)
    (True, 1, 16, 8, 1, 2, 0, (6, 7, 8)),          # k1 s2: odd positions are reached by no tap (dx = 0 there)
    (True, 1, 8, 8, 4, 2, 1, (9, 8, 7)),           # even kernel, odd extents
    (False, 2, 8, 16, 3, 2, 1, (14, 9)),           # 2-D strided layer (cascade X-ray encoder, model_progressive.py:46-49)
])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_implicit_gemm_vs_fp64_and_im2col(cfg, dtype):
    """Implicit-GEMM convolution (hvc_conv_gemm: forward, stride-1 input gradient over dy with mirrored taps, split-K weight
    gradient) against F.conv in fp64 and against the im2col path: the forward is bit-identical to im2col + GEMM (same
    products in the same order), the gradients agree to the operand precision."""
    import torch.nn.functional as F
    from hvc import functional as HF
    from hvc import ops
    is3d, B, Cin, Cout, k, stride, pad, sp = cfg
    g = torch.Generator().manual_seed(Cin * 131 + Cout + k)
    x = torch.randn(B, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, *([k] * len(sp)), generator=g) / (Cin * k ** len(sp)) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.to(dtype).double().requires_grad_(True) for t in (x, w, b))
    y_ref = (F.conv3d if is3d else F.conv2d)(xr, wr, br, stride=stride, padding=pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.to(dtype).double())
    sp3 = tuple(sp) if is3d else (1, *sp)
    geom = ops.ConvGeometry(B, Cin, sp3, (k,) * 3 if is3d else (1, k, k), stride, (pad,) * 3 if is3d else (0, pad, pad))
    to_cl = lambda t: (t if is3d else t.unsqueeze(2)).permute(0, 2, 3, 4, 1).contiguous()
    from_cl = lambda t: (lambda u: u if is3d else u.squeeze(2))(t.permute(0, 4, 1, 2, 3))
    results = {}
    old, old_direct = HF.CONV_IMPLICIT, HF.CONV_DIRECT
    HF.CONV_DIRECT = False           # this test is about the GEMM paths (the streaming / halo-tile kernels have their own below)
    try:
        for path in ("implicit", "im2col"):
            HF.CONV_IMPLICIT = path == "implicit"
            xd = to_cl(x.to(dev()).to(dtype)).requires_grad_(True)
            wd = w.to(dev()).to(dtype).float().requires_grad_(True)
            bd = b.to(dev()).to(dtype).float().requires_grad_(True)
            y = HF.ConvFn.apply(xd, wd, bd, None, geom, dtype, dtype)
            (y.float() * to_cl(dy.to(dev()).to(dtype)).float()).sum().backward()
            results[path] = (from_cl(y.detach()).float().cpu(), from_cl(xd.grad).float().cpu(), wd.grad.cpu(), bd.grad.cpu())
    finally:
        HF.CONV_IMPLICIT, HF.CONV_DIRECT = old, old_direct
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    refs = (y_ref.detach(), xr.grad, wr.grad, br.grad)
    for path, got in results.items():
        for name, a, r in zip(("y", "dx", "dw", "db"), got, refs):
            err = ((a.double() - r).abs().max() / (r.abs().max() + 1e-12)).item()
            assert err < tol, (path, name, err)
    assert torch.equal(results["implicit"][0], results["im2col"][0])


@pytest.mark.parametrize("cfg", [
    # B, Cin, Cout, k, stride, spatial
    (2, 1, 32, 3, 1, (9, 10, 37)),       # cascade glue Conv3d(1, 32, 3, padding=1) (model_progressive.py:171,240): ragged blocks on every axis
    (1, 1, 64, 3, 1, (5, 17, 33)),       # detail_enhancer[0] (model_progressive.py:260)
    (1, 1, 32, 3, 1, (4, 8, 32)),        # exactly one output block
    (2, 1, 64, 3, 2, (13, 9, 70)),       # first voxel-embed layer of the direct model (hybrid_vit_backbone.py:199): stride 2
    (1, 1, 32, 3, 2, (16, 16, 64)),
    (2, 32, 1, 1, 1, (7, 9, 11)),        # detail_enhancer[-1] Conv3d(32, 1, 1) (model_progressive.py:266): ragged row count
    (1, 64, 1, 1, 1, (16, 16, 17)),
    (1, 8, 1, 1, 1, (5, 5, 5)),
    (1, 128, 1, 1, 1, (6, 10, 33)),
    (2, 64, 32, 3, 1, (5, 9, 37)),       # detail_enhancer[3] Conv3d(64, 32, 3, padding=1) (model_progressive.py:263) through the LDS halo tile
    (1, 32, 64, 3, 1, (4, 6, 33)),       # (hvc_conv3_halo), forward and - mirrored taps, swapped channels - input gradient; ragged blocks
    (1, 32, 32, 3, 1, (2, 8, 32)),       # exactly one block
    (1, 64, 64, 3, 1, (3, 5, 70)),
    (1, 1, 32, 3, 1, (24, 20, 72)),      # several blocks along every axis (interior blocks with full halos)
    (1, 64, 32, 3, 1, (10, 20, 72)),
])
def test_conv_single_channel_streaming_kernels_vs_fp64_and_gemm_path(cfg):
    """The single-channel layers of the cascade glue as streaming kernels (hvc_conv_c1_fwd / _dw, hvc_conv_o1_fwd / _bwd) against F.conv3d
    in fp64 on the bf16-rounded operands, and against the im2col / implicit GEMM path they replace: same products, fp32 accumulation in
    a different order, so outputs agree to a bf16 ulp and the fp32 weight / bias gradients to 1e-3."""
    import torch.nn.functional as F
    from hvc import functional as HF
    from hvc import ops
    B, Cin, Cout, k, stride, sp = cfg
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(Cin * 131 + Cout + sp[2])
    x = torch.randn(B, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = k // 2
    xr, wr, br = (t.to(dtype).double().requires_grad_(True) for t in (x, w, b))
    y_ref = F.conv3d(xr, wr, br, stride=stride, padding=pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.to(dtype).double())
    geom = ops.ConvGeometry(B, Cin, sp, (k,) * 3, stride, (pad,) * 3)
    results = {}
    old = HF.CONV_DIRECT
    try:
        for path in ("direct", "gemm"):
            HF.CONV_DIRECT = path == "direct"
            xd = x.to(dev()).to(dtype).permute(0, 2, 3, 4, 1).contiguous().requires_grad_(True)
            wd = w.to(dev()).to(dtype).float().requires_grad_(True)
            bd = b.to(dev()).to(dtype).float().requires_grad_(True)
            ops.PROFILE = prof = []
            try:
                y = HF.ConvFn.apply(xd, wd, bd, None, geom, dtype, dtype)
            finally:
                ops.PROFILE = None
            (y.float() * dy.to(dev()).to(dtype).permute(0, 2, 3, 4, 1).float()).sum().backward()
            results[path] = (y.detach().permute(0, 4, 1, 2, 3).float().cpu(), xd.grad.permute(0, 4, 1, 2, 3).float().cpu(), wd.grad.cpu(), bd.grad.cpu())
            ran = {n for n, *_ in prof}
            if Cin == 1:
                assert ("conv_c1_fwd_kernel" in ran) == (path == "direct"), "the streaming kernel must be the path that ran"
            elif k == 3:
                assert ("conv3_halo_kernel" in ran) == (path == "direct"), "the halo-tile kernel must be the path that ran"
    finally:
        HF.CONV_DIRECT = old
    refs = (y_ref.detach(), xr.grad, wr.grad, br.grad)
    for path, got in results.items():
        wtol = 1e-3 if path == "direct" and 1 in (Cin, Cout) else 2e-2
        for name, a, r, tol in zip(("y", "dx", "dw", "db"), got, refs, (8e-3, 8e-3, wtol, wtol)):
            assert a.shape == r.shape, (path, name, a.shape, r.shape)
            err = ((a.double() - r).abs().max() / (r.abs().max() + 1e-12)).item()
            assert err < tol, (path, name, err)
    for name, a, c in zip(("y", "dx"), results["direct"][:2], results["gemm"][:2]):
        assert ((a - c).abs().max() / (c.abs().max() + 1e-12)).item() < 8e-3, name


def test_conv_gemm_rejects_what_it_cannot_gather():
    """hvc_conv_gemm error behaviour: channel counts that are not a multiple of 8 (the gather moves 16-byte channel vectors; such
    layers stay on im2col), a wrong weight shape, a wrong activation shape and depth-slab geometries raise instead of computing."""
    from hvc import ops
    x = torch.randn(1, 4, 5, 6, 12, device=dev(), dtype=torch.bfloat16)
    w = torch.randn(8, 27 * 12, device=dev(), dtype=torch.bfloat16)
    geom = ops.ConvGeometry(1, 12, (4, 5, 6), (3, 3, 3), 1, (1, 1, 1))
    with pytest.raises(ValueError, match="C % 8"):
        ops.conv_gemm(x, w, geom)
    x16 = torch.randn(1, 4, 5, 6, 16, device=dev(), dtype=torch.bfloat16)
    g16 = ops.ConvGeometry(1, 16, (4, 5, 6), (3, 3, 3), 1, (1, 1, 1))
    with pytest.raises(ValueError, match="weights"):
        ops.conv_gemm(x16, w, g16)
    with pytest.raises(ValueError, match="channels-last"):
        ops.conv_gemm(x16[:, :3], torch.randn(8, 27 * 16, device=dev(), dtype=torch.bfloat16), g16)
    slab = ops.ConvGeometry(1, 16, (4, 5, 6), (3, 3, 3), 1, (1, 1, 1), out_depth=2)
    with pytest.raises(ValueError, match="slab"):
        ops.conv_gemm(x16, torch.randn(8, 27 * 16, device=dev(), dtype=torch.bfloat16), slab)
    from hvc import _lib
    lib = _lib.load()                                     # and the C ABI itself, behind the Python checks
    y = torch.empty(120, 8, device=dev(), dtype=torch.bfloat16)
    rc = lib.hvc_conv_gemm(0, x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 12, 4, 5, 6, 3, 3, 3, 1, 1, 1, 1, 0, 8, 27 * 12, 8,
                           None, None, 0, 0, None, 0, 1, 1, None)
    assert rc != 0 and b"C % 8" in lib.hvc_last_error()
    rc = lib.hvc_conv_gemm(7, x16.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 16, 4, 5, 6, 3, 3, 3, 1, 1, 1, 1, 0, 8, 27 * 16, 8,
                           None, None, 0, 0, None, 0, 1, 1, None)
    assert rc != 0 and b"mode" in lib.hvc_last_error()


@pytest.mark.parametrize("seed", range(8))
def test_ssim_l1_loss_random_sizes_and_weights(seed):
    """DirectRegressionLoss (model_direct.py:69-133) at random non-cubic volume sizes from the 11-voxel window upwards, random
    loss weights and value ranges: the three loss terms and d(total)/d(pred) against the oracle's autograd; an extent below
    the window is the RuntimeError F.avg_pool3d raises in the reference."""
    from direct_regression.model_direct import DirectRegressionLoss
    from oracle import hvc_oracle as O
    gen = torch.Generator().manual_seed(13000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen).item())
    B = ri(1, 3)
    size = (ri(11, 29), ri(11, 29), ri(11, 40)) if seed else (11, 11, 11)
    wl1, wss = (1.0, 0.5) if seed % 2 else (float(torch.rand(1, generator=gen)) + 0.1, float(torch.rand(1, generator=gen)) + 0.1)
    amp = (1.0, 0.05, 3.0, 1.0)[seed % 4]
    t = (torch.rand(B, 1, *size, generator=gen) * 2 - 1) * amp
    p = (t * float(torch.rand(1, generator=gen)) + torch.randn(B, 1, *size, generator=gen) * 0.3 * amp)
    pr = p.clone().requires_grad_(True)
    ref = O.direct_regression_loss(pr, t, l1_weight=wl1, ssim_weight=wss)
    ref["total_loss"].backward()
    pg = p.to(dev()).requires_grad_(True)
    got = DirectRegressionLoss(wl1, wss)(pg, t.to(dev()))
    got["total_loss"].backward()
    for k in ("total_loss", "l1_loss", "ssim_loss"):
        assert abs(got[k].item() - ref[k].item()) < 1e-5 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())
    err = (pg.grad.cpu() - pr.grad).norm() / pr.grad.norm()
    assert err < 1e-3, err
    if seed == 0:
        small = torch.zeros(1, 1, 9, 12, 12, device=dev())
        with pytest.raises(RuntimeError, match="smaller than kernel size"):
            DirectRegressionLoss()(small, small)


@pytest.mark.parametrize("shape", [(1, 11, 11, 11), (2, 40, 33, 50), (1, 70, 16, 130), (2, 128, 128, 128), (1, 96, 200, 72)])
def test_ssim_fused_pass_matches_three_pass_pipeline(shape, hvc_option):
    """The one-pass SSIM + L1 kernels (16 x 16 columns marching along D, eleven (W, H)-summed planes per position in registers;
    model_direct.py:88-131) against the three-axis-pass pipeline they replace (HVC_LOSS_FUSED=0, the form the oracle tests have
    covered since round 1): every window sum is the same ascending zero-padded sum axis by axis; what differs is where hipcc
    contracts a multiply-add into an fma inside the two kernels' point functions (one-ulp differences that the 1 / (D1 D2) factor
    of the derivative maps passes on), so maps and dpred are held to 1e-5 of their largest entry and the scalar losses - block sums
    over a different block partition - to 2e-6.  Shapes: the smallest legal volume, extents that are not multiples of the 16 x 16
    tile, D cut into chunks with halo planes, the 128^3 benchmark size."""
    from hvc import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    p = (torch.rand(B, D, H, W, generator=g) * 2 - 1).to(dev())
    t = (p + 0.3 * torch.randn(B, D, H, W, generator=g).to(dev())).clamp(-1, 1)
    gs = torch.tensor([0.7, 0.2, -0.4], device=dev())
    res = {}
    for fused in (0, 1):
        hvc_option("HVC_LOSS_FUSED", fused)
        out, gmaps = ops.ssim_l1_fwd(p, t, 11, 1.0, 0.5)
        dp = ops.ssim_l1_bwd(p, t, gmaps, gs, 11, 1.0, 0.5)
        res[fused] = (out.clone(), gmaps.clone(), dp.clone())
    for name, i in (("derivative maps", 1), ("dpred", 2)):
        e = ((res[0][i] - res[1][i]).abs().max() / res[0][i].abs().max()).item()
        _note(f"ssim_fused/{name}/{'x'.join(map(str, shape))}", e, 1e-5)
        assert e < 1e-5, (name, e)
    assert torch.allclose(res[0][0], res[1][0], rtol=2e-6, atol=1e-7), (res[0][0], res[1][0])


@pytest.mark.parametrize("seed", range(8))
def test_attention_forward_64_row_kernel_matches_oracle_and_32_row_kernel(seed, hvc_option):
    """attn_fwd2_kernel (two query blocks per wavefront, 32-key tiles; picked automatically from 512 workgroups up) pinned
    through HVC_ATTN_FWD_ROWS on small ragged shapes: output and log-sum-exp against the oracle, the same dropout mask as the
    32-row kernel (zero pattern of O for one-hot V), a late spike that forces the deferred rescale, packed strides, and
    the backward kernels consuming its output."""
    from hvc import ops
    from oracle import hvc_oracle as O
    rng = torch.Generator().manual_seed(14000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    B, H, D = ri(1, 2), ri(1, 3), (64, 32)[seed % 2]
    N, M = (ri(1, 700), ri(1, 300)) if seed else (256, 64)
    qkv = torch.randn(B, max(N, M), 3, H, D, generator=rng)
    if seed % 3 == 1:
        qkv[0, M - 1, 1, 0] = qkv[0, min(5, N - 1), 0, 0] * 7.0        # spike in the last key tile
    qkv = qkv.to(dev(), torch.bfloat16)
    q, k, v = qkv[:, :N, 0], qkv[:, :M, 1], qkv[:, :M, 2]
    ref = O.attention_core(*(t.float().cpu().permute(0, 2, 1, 3) for t in (q, k, v)), D ** -0.5).permute(0, 2, 1, 3)
    rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).abs().max() / max(b.float().abs().max().item(), 1.0)).item()
    out = {}
    for rows in ("32", "64"):
        hvc_option("HVC_ATTN_FWD_ROWS", rows)
        o, lse = ops.attention_fwd(q, k, v, D ** -0.5)
        assert rel(o, ref) < 3e-2, (rows, B, H, N, M, D)
        od, lsed = ops.attention_fwd(q, k, v, D ** -0.5, 0.25, 99 + seed)
        out[rows] = (o, lse, od, lsed)
    assert rel(out["64"][0], out["32"][0]) < 2e-2
    assert (out["64"][1] - out["32"][1]).abs().max().item() < 2e-3            # lse (natural log), fp32
    assert (out["64"][3] - out["32"][3]).abs().max().item() < 2e-3
    assert rel(out["64"][2], out["32"][2]) < 2e-2                              # same keep mask => same dropped output
    # the mask itself: V = one-hot over keys in feature 0 .. D-1 (M <= D keys covered), O(q, d) != 0 <=> key d kept
    if M <= D:
        eye = torch.zeros(B, M, H, D, device=dev(), dtype=torch.bfloat16)
        eye[:, torch.arange(M), :, torch.arange(M)] = 1
        masks = []
        for rows in ("32", "64"):
            hvc_option("HVC_ATTN_FWD_ROWS", rows)
            masks.append(ops.attention_fwd(q, k, eye, D ** -0.5, 0.25, 99 + seed)[0] != 0)
        assert torch.equal(masks[0], masks[1])
    hvc_option("HVC_ATTN_FWD_ROWS", "64")
    o, lse = out["64"][2], out["64"][3]
    do = torch.randn(B, N, H, D, generator=rng).to(dev(), torch.bfloat16)
    _, _, dv = ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, 0.25, 99 + seed)
    lhs, rhs = (do.double() * o.double()).sum().item(), (dv.double() * v.double()).sum().item()
    assert abs(lhs - rhs) < 5e-2 * (abs(lhs) + abs(rhs) + 1.0)


def test_total_variation_loss_vs_golden_and_oracle(golden):
    """TotalVariationLoss (loss_multiscale.py:140-188) through the fused HIP pass: the reference's own numbers (both modes,
    gradients, the sqrt(eps) plateau of a constant volume), then random multi-channel / non-cubic shapes against the oracle,
    extents of 2 and a bf16 input (cast to fp32 as the reference does)."""
    from direct_regression.progressive_cascade.loss_multiscale import TotalVariationLoss
    from oracle import hvc_oracle as O
    g = golden("tv")
    tv = TotalVariationLoss()
    for target, tag in ((None, "tv_pred"), (g.t("target").to(dev()), "tv_match")):
        p = g.t("pred").to(dev()).requires_grad_(True)
        loss = tv(p, target)
        g.check("", tag, loss, 1e-5)
        loss.backward()
        g.check("", tag + "_grad", p.grad, 1e-4)
    g.check("", "tv_flat", tv(torch.full((1, 1, 4, 5, 6), 0.25, device=dev())), 1e-5)
    gen = torch.Generator().manual_seed(77)
    for shape, dtype in (((1, 1, 2, 2, 2), torch.float32), ((2, 3, 5, 17, 9), torch.float32), ((1, 1, 33, 20, 64), torch.float32),
                         ((2, 1, 16, 16, 16), torch.bfloat16)):
        p = (torch.rand(*shape, generator=gen) * 2 - 1).to(dtype)
        t = (torch.rand(*shape, generator=gen)).to(dtype)
        pr = p.clone().requires_grad_(True)
        ref = O.total_variation_loss(pr, t)
        ref.backward()
        pg = p.to(dev()).requires_grad_(True)
        got = tv(pg, t.to(dev()))
        got.backward()
        assert abs(got.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item())), shape
        err = (pg.grad.float().cpu() - pr.grad.float()).abs().max() / pr.grad.float().abs().max()
        assert err < (1e-3 if dtype == torch.float32 else 2e-2), (shape, err.item())
    with pytest.raises(RuntimeError, match="HIP path only"):
        tv(torch.zeros(1, 1, 4, 4, 4))


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment starts its own two ranks (torch.distributed.run child) and
    rank 0 prints one JSON line whose rccl_ranks says two ranks took part.  Rehearsal on the one-GPU test box: both
    ranks on cuda:0 over gloo (HVC_TEST_SINGLE_DEVICE / HVC_DIST_BACKEND); the production launch is identical with
    one rank per GPU over nccl = RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HVC_TEST_SINGLE_DEVICE="1", HVC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "direct64", "--no-profile"], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 8
    assert out["value"] > 0 and out["scaling"] == "weak"


def _run_bench(args, timeout=850):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1100)
def test_bench_runs_the_ddp_path_over_rccl_on_one_gpu():
    """`bench.py --gpus 1 --ddp`: the multi-GPU code path on the one GPU of the test box - init_process_group("nccl") (= RCCL),
    the bucketed gradient all-reduce (hvc.reducer) hooked into the HIP autograd Functions' backward, barrier + max-over-ranks timing - so that the first
    RCCL initialisation does not happen on the driver's 8-GPU run (reference: direct_regression/train_direct_4gpu.py:25-37, :146).
    A world-size-1 all-reduce moves no data over xGMI: same-box runs differ by 0.5 - 1.5 % (profiles/r04_bench_ddp_rccl_world1.json), one box in
    five showed 5 % (RCCL's world-size-1 copy kernels next to the attention kernels); the bound is 8 %, the measured value goes to the margins log."""
    common = ["--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-extra"]      # profiling on, as the driver runs it
    # alternating runs, best of two each: run-to-run spread of a 20-step measurement on one box is ~2 % (clock / placement)
    runs = [_run_bench(common + (["--ddp"] if i % 2 == 0 else [])) for i in range(4)]
    ddp = min(runs[0::2], key=lambda d: d["ms_per_step"])
    plain = min(runs[1::2], key=lambda d: d["ms_per_step"])
    assert ddp["dist_backend"] == "nccl" and ddp["rccl_ranks"] == 1 and ddp["nccl_version"], ddp
    assert ddp["ddp"]["gradient_as_bucket_view"] and ddp["n_gpus"] == 1
    assert "BucketedGradReducer" in ddp["ddp"]["gradient_exchange"]["reducer"] and ddp["ddp"]["gradient_exchange"]["buckets"] >= 2
    assert ddp["host_enqueue_ms_per_step"] < 0.9 * ddp["ms_per_step"], "the launch thread must stay ahead of the GPU"
    assert "dist_backend" not in plain
    assert "after the timed region" in ddp["roofline"]["measured_over"] and "timed steps" in plain["roofline"]["measured_over"]
    rel = abs(ddp["ms_per_step"] - plain["ms_per_step"]) / plain["ms_per_step"]
    _note("ddp_vs_plain_ms_per_step", rel, 0.08)
    assert rel < 0.08, (plain["ms_per_step"], ddp["ms_per_step"])
    assert math.isfinite(ddp["config"]["loss"])


@pytest.mark.timeout(900)
def test_bench_captures_the_ddp_step_with_its_rccl_all_reduce_in_a_hipgraph():
    """The launch-bound 64^3 step under DDP as ONE hipGraph (static_graph DDP, >= 11 eager warm-up iterations, RCCL collectives
    recorded into the capture): replays must train (finite loss) at about the speed of the single-GPU graph."""
    common = ["--gpus", "1", "--workload", "direct64", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-extra", "--no-profile"]
    runs = [_run_bench(common + (["--ddp"] if i % 2 == 0 else [])) for i in range(4)]      # alternating, best of two each
    ddp = min(runs[0::2], key=lambda d: d["ms_per_step"])
    plain = min(runs[1::2], key=lambda d: d["ms_per_step"])
    assert plain["config"]["launch"].startswith("hipGraph") and ddp["config"]["launch"].startswith("hipGraph")
    assert ddp["dist_backend"] == "nccl" and ddp["ddp"]["captured_in_hipgraph"] and ddp["ddp"]["static_graph"]
    assert math.isfinite(ddp["config"]["loss"])
    rel = ddp["ms_per_step"] / plain["ms_per_step"] - 1
    _note("ddp_graph_vs_plain_graph_ms_per_step", rel, 0.10)
    assert rel < 0.10, (plain["ms_per_step"], ddp["ms_per_step"])


# ----------------------------------------------------------------------------------------------
# round-2 parity additions ("close the parity holes")
# ----------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_direct_model_full_size_64_forward_and_backward_vs_oracle(monkeypatch):
    """BASELINE config #1/#2 geometry (64^3, 2-view 512^2, 4096 + 4096 tokens): forward AND backward of the whole model
    + DirectRegressionLoss against the CPU oracle on identical weights / inputs, fp32 mode, 1e-3 max-rel per tensor.
    Eval mode (running BN statistics; dropout is off in eval).  The stem weights sit behind ReLU + max-pool routing: the
    oracle is evaluated with the routing the HIP stem used (see test_xray_conditioning_vs_golden)."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import synthetic
    from oracle import hvc_oracle as O
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(64, 64, 64)).eval()
    gen = torch.Generator().manual_seed(31)
    with torch.no_grad():
        for blk in m.vit_backbone.blocks:
            blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)
            blk.adaln.linear.bias.copy_(torch.randn(blk.adaln.linear.bias.shape, generator=gen) * 0.02)
    xr, ct = synthetic.sample(3, (64, 64, 64), 512)
    xr, ct = xr[None], ct[None]
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k) for k, v in m.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    records = _capture_stem_routing(monkeypatch)
    m.to(dev())
    pred = m(xr.to(dev()))
    loss = DirectRegressionLoss(1.0, 0.5)(pred, ct.to(dev()))
    loss["total_loss"].backward()
    # the oracle, evaluated with the routing the HIP stem used (identical function wherever the two routings agree - and they
    # must agree on all but < 1e-3 of the live windows / gates): every tensor, routed or not, then holds the flat 1e-3
    hip_route = _hip_routing(records)
    route = {"use": hip_route}
    ref = O.direct_ct_regression(xr, P, route=route)
    ref_loss = O.direct_regression_loss(ref, ct)
    ref_loss["total_loss"].backward()
    _routing_agreement(route, hip_route)
    assert _maxrel(pred, ref) < F32_TOL
    for k in ("total_loss", "l1_loss", "ssim_loss"):
        assert abs(loss[k].item() - ref_loss[k].item()) < F32_TOL * abs(ref_loss[k].item()) + 1e-6, k
    worst = {}
    for k, p in m.named_parameters():
        r = P[k].grad
        if r.abs().max() < 1e-9:
            continue
        err = _maxrel(p.grad, r)
        _note(f"pgrad/{k}", err, F32_TOL)
        assert err < F32_TOL, (k, err)
        worst[k] = err
    print("64^3 fwd+bwd vs oracle: worst", max(worst.items(), key=lambda kv: kv[1]))


@pytest.mark.timeout(900)
def test_direct_model_128_forward_and_backward_vs_oracle_probes(golden, monkeypatch):
    """The headline workload's model (DirectCTRegression at 128^3 = 32768 tokens, BASELINE config #3 geometry, B = 1) in fp32 mode:
    loss, output volume and EVERY parameter gradient against tests/golden/direct128_probes.npz at 1e-3.  The fixture is
    ORACLE-generated (tests/golden/make_direct128_probes.py: the reference raises at 128^3, hybrid_vit_backbone.py:178-188, so
    only the oracle - pinned to the reference by the other fixtures at N <= 4096 - can say what this model computes): 512 probes +
    checksums per large tensor.  X-ray stem gradients (behind its ReLU / max-pool routing) by the routing-aware method: the
    upstream gradients d loss / d features and d loss / d cond are compared with the fixture, then pushed through the oracle stem
    evaluated with the routing the HIP stem used."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import synthetic
    from oracle import hvc_oracle as O
    g = golden("direct128_probes")
    vol = (128, 128, 128)
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=vol).eval()
    gen = torch.Generator().manual_seed(31)
    with torch.no_grad():
        for blk in m.vit_backbone.blocks:
            blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)
            blk.adaln.linear.bias.copy_(torch.randn(blk.adaln.linear.bias.shape, generator=gen) * 0.02)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    chk = np.array([sum(float(v.double().sum()) for v in state.values() if v.dtype.is_floating_point),
                    sum(float(v.double().abs().sum()) for v in state.values() if v.dtype.is_floating_point)])
    assert np.allclose(chk, g.z["weights_checksum"], rtol=1e-9), "the seeded weights differ from the ones the fixture was made with"
    xr, ct = synthetic.sample(3, vol, 512)
    xr, ct = xr[None], ct[None]
    records = _capture_stem_routing(monkeypatch)
    m.to(dev())
    # DirectCTRegression.forward (model_direct.py:59-85) step by step, to keep handles on the stem's outputs
    xd = xr.to(dev())
    t = torch.zeros(1, 256, device=dev())
    _, cond, feats = m.xray_encoder(xd, t)
    cond.retain_grad()
    feats.retain_grad()
    pred = m.vit_backbone(m.initial_volume.expand(1, -1, -1, -1, -1), feats.flatten(2).transpose(1, 2), cond)
    with torch.no_grad():
        assert torch.equal(pred, m(xd))                                   # the same function as the module's own forward
    loss = DirectRegressionLoss(1.0, 0.5)(pred, ct.to(dev()))
    loss["total_loss"].backward()
    for k in ("total_loss", "l1_loss", "ssim_loss"):
        assert abs(loss[k].item() - float(g.z["loss/" + k])) < F32_TOL * abs(float(g.z["loss/" + k])), k
    g.check("out", "pred", pred, F32_TOL)
    # gradients arriving at the stem's two outputs: through the cross-attention context (the HIP module's `feats` output feeds
    # nothing else; its pooled sibling feeds cond) and through the conditioning vector
    g.check("xgrad", "ctx", feats.grad.flatten(2).transpose(1, 2), F32_TOL)
    g.check("xgrad", "cond", cond.grad, F32_TOL)
    routed = lambda k: k.startswith("xray_encoder.encoder.")
    for k, p in m.named_parameters():
        if not routed(k):
            g.check("pgrad", k, p.grad, F32_TOL)
    # stem: oracle with the HIP routing, fed with the HIP path's own upstream gradients (just checked against the fixture)
    hip_route = _hip_routing(records)
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k) for k, v in state.items() if k.startswith("xray_encoder.")}
    route = {"use": hip_route}
    _, cond_c, feats_c = O.xray_conditioning(xr, torch.zeros(1, 256), P, "xray_encoder.", False, None, route)
    _routing_agreement(route, hip_route)
    assert _maxrel(feats, feats_c) < F32_TOL and _maxrel(cond, cond_c) < F32_TOL
    torch.autograd.backward([feats_c, cond_c], [feats.grad.cpu(), cond.grad.cpu()])      # cond_c's own dependence on the features: autograd
    for k, p in m.named_parameters():
        if routed(k):
            _note(f"pgrad(routed)/{k}", _maxrel(p.grad, P[k].grad), F32_TOL)
            assert _maxrel(p.grad, P[k].grad) < F32_TOL, (k, _maxrel(p.grad, P[k].grad))


def test_attention_full_size_128_dk_dv_rows_vs_fp64():
    """N = 32768, h x d = 4 x 64 (config #3 attention): dK and dV of probed KEY rows against a direct fp64 evaluation.
    A key row's gradient needs every query: dV[j] = sum_i P[i,j] dO[i], dK[j] = scale * sum_i dS[i,j] Q[i] with
    dS = P * (dP - delta); the 8 x 32768 column slab of P is formed from the kernel-independent fp64 row statistics."""
    from hvc import ops
    B, H, N, D = 1, 4, 32768, 64
    g = torch.Generator().manual_seed(129)
    q, k, v = (torch.randn(B, N, H, D, generator=g).to(dev()) for _ in range(3))
    do = torch.randn(B, N, H, D, generator=g).to(dev())
    scale = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, scale, 0.0, 0)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, scale, 0.0, 0)
    b, hh = 0, 1
    keys = torch.tensor([0, 1, 63, 64, 127, 128, 16383, 32767])
    qd, kd, vd, dod = (t[b, :, hh].double() for t in (q, k, v, do))
    # fp64 row statistics in chunks (N x N never materialised): lse_i and delta_i = sum_j P_ij dP_ij
    lse_ref = torch.empty(N, dtype=torch.float64, device=dev())
    delta_ref = torch.empty(N, dtype=torch.float64, device=dev())
    for s in range(0, N, 2048):
        sc = (qd[s:s + 2048] @ kd.t()) * scale
        l_ = torch.logsumexp(sc, dim=-1)
        p_ = torch.exp(sc - l_[:, None])
        lse_ref[s:s + 2048] = l_
        delta_ref[s:s + 2048] = (p_ * (dod[s:s + 2048] @ vd.t())).sum(-1)
    pcol = torch.exp((qd @ kd[keys].t()) * scale - lse_ref[:, None])          # (N, 8)
    dv_ref = pcol.t() @ dod
    dscol = pcol * (dod @ vd[keys].t() - delta_ref[:, None])
    dk_ref = (dscol.t() @ qd) * scale
    e_dv = ((dv[b, keys, hh].double() - dv_ref).abs().max() / dv_ref.abs().max()).item()
    e_dk = ((dk[b, keys, hh].double() - dk_ref).abs().max() / dk_ref.abs().max()).item()
    assert e_dv < F32_TOL and e_dk < F32_TOL, (e_dv, e_dk)


@pytest.mark.timeout(900)
def test_direct_model_full_size_128_forward_backward_batch2():
    """The benchmark's own step (config #3 per-GPU slice: 128^3, batch 2, 32768 tokens) as a checked test: fwd + loss + bwd
    is finite, reproducible bit for bit from one run to the next (no atomics anywhere on the path), and the bf16 (autocast)
    gradients agree with the fp32-mode gradients in norm."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import synthetic
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(128, 128, 128)).eval()     # eval: running BN statistics, dropout off -> deterministic
    gen = torch.Generator().manual_seed(33)
    with torch.no_grad():
        for blk in m.vit_backbone.blocks:
            blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)
    m.to(dev())
    xr, ct = synthetic.batch(5, 2, (128, 128, 128), 512)
    xr, ct = xr.to(dev()), ct.to(dev())
    crit = DirectRegressionLoss(1.0, 0.5)

    def run(bf16):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            pred = m(xr)
            loss = crit(pred.float(), ct)["total_loss"]
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}
    l1, g1 = run(False)
    l2, g2 = run(False)
    l16, g16 = run(True)
    assert math.isfinite(l1) and l1 == l2 and abs(l16 - l1) < 2e-2 * abs(l1)
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g1[k], g2[k]), f"{k}: gradient differs between two identical runs"
    num = sum(((g16[k] - g1[k]).double() ** 2).sum().item() for k in g1)
    den = sum((g1[k].double() ** 2).sum().item() for k in g1)
    assert (num / den) ** 0.5 < 6e-2, (num / den) ** 0.5                          # whole-gradient relative error, bf16 vs fp32 mode
    for k in ("vit_backbone.blocks.0.self_attn.qkv.weight", "vit_backbone.blocks.3.mlp.0.weight", "vit_backbone.blocks.1.cross_attn.kv.weight"):
        e = ((g16[k] - g1[k]).norm() / g1[k].norm()).item()
        assert e < 8e-2, (k, e)


@pytest.mark.timeout(1100)
@pytest.mark.parametrize("fp8,policy", [(False, "on"), (True, "on"), (False, "auto")],
                         ids=["bf16_attention", "fp8_attention", "bf16_attention_auto_policy"])
def test_progressive_trainer_stage3_256_full_losses_with_checkpointing(tmp_path, fp8, policy):
    """BASELINE config #5 as a test (per-GPU slice, batch 1; with the "mi355x": {"fp8_attention": true} switch the attention
    forward products run as e4m3 MFMAs, as the config names): one optimisation step of cascade stage 3 at 256^3 through
    train_progressive_4gpu.build_stage / train_step with gradient checkpointing (reference model_progressive.py:286-291,
    train_progressive_4gpu.py:214-219) and the full Stage3Loss (L1 + SSIM + TV + frequency + 0.3 DRR reprojection,
    loss_multiscale.py:384-432; the VGG term needs downloaded weights and is skipped with a notice).  Stages 1 and 2 are
    frozen: no gradients, weights untouched; stage 3 and its X-ray head move; every loss term is finite.
    policy "on" = the reference's behaviour (recompute the ViT in the backward pass); "auto" = the MI355X default, which keeps the
    ~4 GB of activations instead while HBM has room (hvc.functional.use_checkpoint) - same step, no recomputation."""
    import copy, json, os, sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "hybrid-vit-cascade_amd", "direct_regression"))
    from progressive_cascade import train_progressive_4gpu as T
    from hvc import synthetic
    cfg = json.load(open(os.path.join(here, "hybrid-vit-cascade_amd", "direct_regression", "progressive_cascade", "config_progressive.json")))
    torch.manual_seed(0)
    m1, _, _, _ = T.build_stage(cfg, 1, tmp_path, dev())
    torch.save({"model_state_dict": m1.state_dict()}, tmp_path / "stage1_best.pth")
    torch.save({"model_state_dict": m1.state_dict()}, tmp_path / "stage2_best.pth")
    del m1
    from hvc import functional as HF
    cfg.setdefault("mi355x", {})["fp8_attention"] = fp8
    HF.set_fp8_attention(bool(cfg["mi355x"]["fp8_attention"]))            # what train_stage does with the config section
    HF.set_checkpoint_policy(policy)
    try:
        _stage3_256_step(T, cfg, tmp_path, synthetic, fp8)
    finally:
        HF.set_fp8_attention(False)
        HF.set_checkpoint_policy("auto")


def _stage3_256_step(T, cfg, tmp_path, synthetic, fp8):
    import copy
    model, crit, opt, _ = T.build_stage(cfg, 3, tmp_path, dev())
    model.train()
    assert model.stage3.use_gradient_checkpointing
    xr, ct = synthetic.batch(7, 1, (256, 256, 256), 512)
    xr, ct = xr.to(dev()), ct.to(dev())
    frozen = copy.deepcopy({k: v for k, v in model.state_dict().items() if k.startswith(("stage1.", "stage2."))})
    w0 = model.stage3.vit_refiner.blocks[0].mlp[0].weight.detach().clone()
    torch.cuda.reset_peak_memory_stats()
    losses = T.train_step(model, crit, opt, None, xr, ct, 3, cfg["training"]["gradient_clip"])
    assert {"total_loss", "l1_loss", "ssim_loss", "tv_loss", "freq_loss", "drr_loss"} <= set(losses), sorted(losses)
    for k, v in losses.items():
        assert torch.isfinite(torch.as_tensor(v)).all(), k
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith(("stage1.", "stage2.")))
    for k, v in model.state_dict().items():
        if k in frozen and "running_" not in k and "num_batches" not in k:
            assert torch.equal(v, frozen[k]), k
    assert not torch.equal(model.stage3.vit_refiner.blocks[0].mlp[0].weight.detach(), w0)
    print(f"stage-3 256^3 step ({'fp8' if fp8 else 'bf16'} attention): peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, "
          f"loss {float(torch.as_tensor(losses['total_loss']).detach()):.4f}")
    # the same seeds, weights and batch: the e4m3 forward moves the step's loss by a few per cent at most
    total = float(torch.as_tensor(losses["total_loss"]).detach())
    assert abs(total - _STAGE3_LOSS.setdefault("bf16", total)) <= 0.05 * abs(_STAGE3_LOSS["bf16"])


_STAGE3_LOSS = {}


def test_device_prefetcher_yields_the_loader_batches_on_the_device():
    """utils.prefetch.DevicePrefetcher (the trainers' batch loop): same batches, same order, tensors on the GPU, extra keys passed
    through, and a train_epoch through it matches the plain loop's losses bit for bit."""
    from utils.prefetch import DevicePrefetcher
    from torch.utils.data import DataLoader
    from hvc.synthetic import SyntheticPatientDataset
    ds = SyntheticPatientDataset(n=5, volume_size=(16, 16, 16), xray_size=64)
    loader = DataLoader(ds, batch_size=2, shuffle=False, pin_memory=True)
    plain = list(loader)
    got = list(DevicePrefetcher(loader, dev()))
    assert len(got) == len(plain) == 3 and len(DevicePrefetcher(loader, dev())) == 3
    for a, b in zip(got, plain):
        assert set(a) == set(b)
        for k in ("drr_stacked", "ct_volume"):
            assert a[k].is_cuda and torch.equal(a[k].cpu(), b[k])
        for k in set(b) - {"drr_stacked", "ct_volume"}:
            assert (torch.equal(a[k], b[k]) if torch.is_tensor(b[k]) else a[k] == b[k])
    assert list(DevicePrefetcher([], dev())) == []
    with pytest.raises(RuntimeError, match="HIP device only"):
        DevicePrefetcher(loader, "cpu")


def test_direct_trainer_checkpoint_resume_on_the_hip_path(tmp_path):
    """save_checkpoint / load_checkpoint of the direct trainer around real HIP steps (reference train_direct_4gpu.py:177-189,
    :273-298): a model + AdamW restored from the checkpoint continue BIT FOR BIT like the run that wrote it - dropout on, so
    the step after the resume also has to find the cached bf16 / conv weight copies refreshed from the loaded parameters."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from direct_regression import train_direct_4gpu as T
    from hvc import synthetic

    def make():
        torch.manual_seed(5)
        m = DirectCTRegression(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=64, vit_depth=2, num_heads=2, xray_feature_dim=64).to(dev()).train()
        o = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
        return m, o, torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=10, eta_min=1e-6)

    crit = DirectRegressionLoss(1.0, 0.5)
    xr, ct = synthetic.batch(11, 2, (16, 16, 16), 64)
    xr, ct = xr.to(dev()), ct.to(dev())

    def step(m, o, s, seed):
        torch.manual_seed(seed)                      # the dropout seeds of a step come from the torch generator
        out = T.train_step(m, crit, o, None, xr, ct, 1.0)
        s.step()
        return float(out["total_loss"].detach())

    m, o, s = make()
    for i in range(2):
        step(m, o, s, 100 + i)
    path = tmp_path / "checkpoint_epoch_2.pt"
    T.save_checkpoint(path, 2, m, o, s, 20.0, 21.0, {"training": {"num_epochs": 10}})
    m2, o2, s2 = make()
    step(m2, o2, s2, 999)                            # populate m2's weight-copy caches with weights the checkpoint will replace
    start, best = T.load_checkpoint(str(path), m2, o2, s2, dev())
    assert (start, best) == (3, 21.0)
    for i in range(2):
        la, lb = step(m, o, s, 200 + i), step(m2, o2, s2, 200 + i)
        assert la == lb, (i, la, lb)
    for (n, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_train_step_vs_reference_fixture(golden, mode):
    """Two consecutive optimisation steps of the HIP trainer's train_step (train_direct_4gpu.py mirror) against the reference's
    own step (direct_regression/train_direct_4gpu.py:59-75, captured by tests/golden/make_golden.py:train_step_fixture):
    loss, pre-clip gradient norm, clipped gradients, and the post-AdamW weights compared by how far they MOVED (units of lr)."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from direct_regression import train_direct_4gpu as T
    g = golden("train_step")
    cfg = [int(v) for v in g.z["cfg"]]
    m = DirectCTRegression(volume_size=tuple(cfg[:3]), xray_img_size=cfg[3], voxel_dim=cfg[4], vit_depth=cfg[5],
                           num_heads=cfg[6], xray_feature_dim=cfg[7])
    _load(m, g.group("params"))
    _zero_dropout(m)
    m.train()
    crit = DirectRegressionLoss(1.0, 0.5)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, fused=True)
    xr, target = g.t("xrays").to(dev()), g.t("target").to(dev())
    tol = _tol(mode)
    lr = 1e-4
    routed = ("xray_encoder.encoder.",)                       # behind ReLU / max-pool routing: norm comparison (see the stem test)
    resolvable = {}          # bf16 mode: per tensor, the elements whose reference gradient stood above the run's gradient error so far
    for step in (1, 2):
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        norms = []
        real_clip = torch.nn.utils.clip_grad_norm_
        try:
            torch.nn.utils.clip_grad_norm_ = lambda params, c: norms.append(real_clip(params, c)) or norms[-1]
            losses = T.train_step(m, crit, opt, None, xr, target, 1.0, autocast_dtype=None if mode == "f32" else torch.bfloat16)
        finally:
            torch.nn.utils.clip_grad_norm_ = real_clip
        assert abs(losses["total_loss"].item() - float(g.z[f"step{step}_loss"])) < tol * float(g.z[f"step{step}_loss"])
        ref_norm = float(g.z[f"step{step}_gradnorm"])
        assert abs(norms[0].item() - ref_norm) < tol * ref_norm, (norms[0].item(), ref_norm)
        moved_badly = {}
        for k, p in m.named_parameters():
            if k.endswith(("encoder.0.bias", "encoder.4.bias", "encoder.8.bias")):
                continue                                     # zero-gradient biases ahead of train-mode BN: both sides hold noise
            if mode == "f32":
                # flat 1e-3; routed tensors (behind the stem's ReLU / max-pool choices) in the Frobenius norm: this is a REFERENCE
                # fixture, the reference's own routing cannot be replaced by the HIP one, and a single window that resolves a
                # near-tie the other way moves one element by O(1) while leaving the norm at the 1e-4 level
                g.check(f"step{step}_clipped", k, p.grad, tol, metric="l2" if k.startswith(routed) else "max")
            else:
                g.check(f"step{step}_clipped", k, p.grad, _ROUTED_BF16_TOL if k.startswith(routed) else _gtol(mode), metric="l2")
            # AdamW moves every weight by <= ~lr per step whatever the gradient's size (step 1 is sign-like: -lr g / (|g| + eps)
            # - lr wd w), so weights are compared by how far they MOVED from the initial fixture weights, in units of lr: even the
            # bf16 run must land on the reference's weights except where a tiny gradient's sign is decided by rounding
            # (such a weight ends 2 lr away per step, never more).
            ref_after, pick = g.ref_values(f"step{step}_after", k)
            ref_move = ref_after - pick(g.z[f"params/{k}"].astype(np.float64))
            got_move = pick((p.detach().double().cpu() - torch.from_numpy(g.z[f"params/{k}"]).double()).numpy())
            diff = np.abs(got_move - ref_move)
            assert diff.max() <= 2.1 * lr * step, (k, diff.max() / lr)
            moved_badly[k] = float((diff > 0.3 * lr).mean())
            # fp32 mode is the parity statement: <= 2 % of the weights of EVERY tensor.
            # bf16 mode: AdamW's first steps are sign-like, -lr g / (|g| + 1e-8): a weight can only be expected to move as in the
            # reference where its reference gradient is RESOLVED by this run, i.e. stands above the run's own gradient error (the
            # norm of that error is bounded above: 0.15, routed 0.4).  So: per tensor, sigma = rms(g_hip - g_ref); among the
            # elements with |g_ref| >= 5 sigma in every step so far, <= 2 % may move differently (and these elements must exist,
            # unless the whole gradient is rounding noise); elements below the noise floor keep the 2.1 lr-per-step bound above.
            if mode == "f32":
                _note(f"step{step}_moved/{k}", moved_badly[k], 0.02, "moved>0.3lr")
                assert moved_badly[k] <= 0.02, (k, moved_badly[k])
            else:
                gref, gpick = g.ref_values(f"step{step}_clipped", k)
                ggot = gpick(p.grad.detach().double().cpu().numpy())
                sigma = float(np.sqrt(np.mean((ggot - gref) ** 2)))
                ok = np.abs(gref) >= 5 * sigma
                resolvable[k] = ok if k not in resolvable else (resolvable[k] & ok)
                nres = int(resolvable[k].sum())
                bad = float((diff[resolvable[k]] > 0.3 * lr).mean()) if nres else 0.0
                _note(f"step{step}_moved_resolved/{k}", bad, 0.02, f"moved>0.3lr among {nres}/{ok.size} resolved")
                assert bad <= 0.02, (k, bad, nres)
                assert nres >= 0.25 * ok.size or k.startswith(routed) or np.abs(gref).max() < 1e-7, (k, nres, ok.size)
        print(f"train_step [{mode}] step {step}: worst fraction of weights that moved differently (> 0.3 lr): {max(moved_badly.values()):.3%}")
    for k, v in m.state_dict().items():
        if "running_" in k:
            g.check("step2_after", k, v, tol if mode == "f32" else 10 * tol)      # bf16 conv outputs feed the batch statistics


def test_sinusoidal_time_embedding_vs_golden(golden):
    """SinusoidalTimeEmbedding (reference models/vit_components.py:152-174) on the GPU against the reference's values."""
    from models.vit_components import SinusoidalTimeEmbedding
    g = golden("time_embedding")
    t = g.t("t").to(dev())
    for dim in (32, 256):
        got = SinusoidalTimeEmbedding(dim)(t)
        assert got.shape == (t.shape[0], dim)
        # arguments reach 1000 rad: one fp32 ulp of the argument is 6e-5, so the bound is absolute
        assert (got.cpu() - g.t(f"emb{dim}")).abs().max().item() < 3e-4


def test_frequency_loss_vs_golden(golden):
    """FrequencyLoss (reference loss_multiscale.py:191-236): rocFFT transform + the fused HIP magnitude / mask / L1 pass,
    loss and gradient against the reference's values (non-cubic volume, default and custom high-frequency weight)."""
    from direct_regression.progressive_cascade.loss_multiscale import FrequencyLoss
    g = golden("frequency")
    for tag, w in (("w2", 2.0), ("w05", 0.5)):
        p = g.t("pred").to(dev()).requires_grad_(True)
        loss = FrequencyLoss(high_freq_weight=w)(p, g.t("target").to(dev()))
        g.check("", f"loss_{tag}", loss, F32_TOL)
        loss.backward()
        g.check("", f"grad_{tag}", p.grad, F32_TOL)


def test_weight_cast_cache_sees_data_assignment_and_device_moves():
    """The bf16 weight copies cached on the parameter object are keyed by (version, device, storage): assigning `.data`,
    or moving the module off the device and back, yields fresh results; an in-place `.data` write needs
    invalidate_param_casts() (documented in hvc/functional.py)."""
    from hvc import functional as HF
    torch.manual_seed(5)
    lin = torch.nn.Linear(64, 32, bias=False).to(dev())
    x = torch.randn(16, 64, device=dev())

    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return HF.linear(x, lin.weight).float()
    y0 = run()
    assert torch.equal(run(), y0)
    lin.weight.data = lin.weight.data * 2                      # new storage, same Parameter object and version
    assert torch.allclose(run(), 2 * y0, rtol=2e-2, atol=1e-3)
    lin.cpu()
    lin.to(dev())                                              # module.to(): same Parameter, storage moved
    assert torch.allclose(run(), 2 * y0, rtol=2e-2, atol=1e-3)
    lin.weight.data.mul_(0.5)                                  # in place through .data: invisible to the key ...
    HF.invalidate_param_casts()                                # ... so the documented call is required
    assert torch.allclose(run(), y0, rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("cfg", [(2, 13, 9, 32, 24, False, 0), (1, 64, 64, 128, 128, False, 0), (3, 40, 56, 16, 20, False, 0),
                                 (2, 13, 9, 32, 24, True, 1), (2, 24, 24, 24, 24, True, 1), (1, 50, 7, 20, 31, True, 1), (2, 8, 8, 8, 8, False, 0)])
def test_resize_loss_kernels_vs_aten(cfg):
    """Fused bilinear resize + L1 / MSE (tails of DRRReprojectionLoss and ProjectionLoss) against F.interpolate + loss:
    value and gradient, both corner conventions, up- and down-sampling, non-square rasters, a strided target view."""
    import torch.nn.functional as F
    from hvc import functional as HF
    B, h, w, S1, S2, ac, mode = cfg
    g = torch.Generator().manual_seed(B * 1000 + h * 31 + w)
    proj = torch.randn(B, h, w, generator=g).to(dev()).requires_grad_(True)
    packed = torch.randn(B, 2, 1, S1, S2, generator=g).to(dev())
    target = packed[:, 1, 0]                                         # strided view, as the X-ray views are
    got = HF.ResizeLossFn.apply(proj, target, ac, mode)
    got.backward()
    p2 = proj.detach().clone().requires_grad_(True)
    r = F.interpolate(p2[:, None], size=(S1, S2), mode="bilinear", align_corners=ac)[:, 0]
    ref = F.mse_loss(r, target) if mode else F.l1_loss(r, target)
    ref.backward()
    assert abs(got.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7
    assert (proj.grad - p2.grad).abs().max().item() < 1e-5 * p2.grad.abs().max().item() + 1e-9


@pytest.mark.parametrize("cfg", [(2, 2, 64, 32, torch.float32), (1, 2, 4096, 512, torch.bfloat16), (3, 1, 50, 64, torch.float32),
                                 (2, 3, 17, 256, torch.bfloat16)])
def test_view_mean_gap_kernels_vs_aten(cfg):
    """Fused mean over views + global average pool of the channels-last X-ray feature maps, forward and backward."""
    from hvc import functional as HF
    B, V, P, E, dt = cfg
    g = torch.Generator().manual_seed(P + E)
    f = torch.randn(B * V, P, E, generator=g).to(dev(), dt).requires_grad_(True)
    wm = torch.randn(B, P, E, generator=g).to(dev())
    wp = torch.randn(B, E, generator=g).to(dev())
    mean, pooled = HF.ViewMeanGapFn.apply(f, V)
    ((mean * wm).sum() + (pooled * wp).sum()).backward()
    f2 = f.detach().float().requires_grad_(True)
    m2 = f2.view(B, V, P, E).mean(1)
    p2 = m2.mean(1)
    ((m2 * wm).sum() + (p2 * wp).sum()).backward()
    assert (mean - m2).abs().max().item() < 1e-5 * m2.abs().max().item()
    assert (pooled - p2).abs().max().item() < 2e-5 * p2.abs().max().item() + 1e-7
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert (f.grad.float() - f2.grad).abs().max().item() < tol * f2.grad.abs().max().item()


def test_graphed_train_step_matches_eager_and_draws_fresh_dropout_masks():
    """hvc.graph.GraphedStep: the whole training step captured as one hipGraph.
      * dropout off: three replays produce bit for bit the losses and the final weights of three eager steps;
      * dropout on: successive replays draw different masks (the device step counter advances inside the graph) and two
        identically seeded captures agree bit for bit."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import synthetic
    from hvc.graph import GraphedStep
    cfg = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=64, vit_depth=1, num_heads=2, xray_feature_dim=32)
    xr, ct = synthetic.batch(3, 2, cfg["volume_size"], cfg["xray_img_size"], device="cuda:0")
    crit = DirectRegressionLoss(1.0, 0.5)

    def make(p_drop):
        torch.manual_seed(0)
        m = DirectCTRegression(**cfg).to(dev()).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = p_drop
        gen = torch.Generator().manual_seed(9)
        with torch.no_grad():
            for blk in m.vit_backbone.blocks:
                blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01, fused=True, capturable=True)
        params = list(m.parameters())

        def step(a, b):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = crit(m(a).float(), b)["total_loss"]
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            return loss
        return m, step

    def run(p_drop, graphed, n=3):
        """n + 1 optimisation steps: eagerly, or one eager warm-up step (it creates the optimizer state, which must exist before
        the capture) followed by n replays of the captured step."""
        m, step = make(p_drop)
        torch.manual_seed(77)
        losses = []
        if graphed:
            g = GraphedStep(step, [xr, ct], warmup=1)
            try:
                for _ in range(n):
                    losses.append(g(xr, ct).item())
            finally:
                g.close()
        else:
            for _ in range(n + 1):
                losses.append(step(xr, ct).item())
            losses = losses[1:]
        return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}

    # dropout off: graph replays == eager steps, bit for bit
    le, we = run(0.0, False)
    lg, wg = run(0.0, True)
    assert le == lg, (le, lg)
    for k in we:
        assert torch.equal(we[k], wg[k]), k
    # dropout on: masks change from replay to replay, and the whole sequence is reproducible
    l1, w1 = run(0.1, True, n=4)
    l2, w2 = run(0.1, True, n=4)
    assert l1 == l2 and len(set(l1)) == len(l1), (l1, l2)
    for k in w1:
        assert torch.equal(w1[k], w2[k]), k


def test_graphed_step_after_load_state_dict_and_counter_ownership():
    """(a) a replay right after load_state_dict (no eager forward in between) must run on the LOADED weights - the
    graph holds the addresses of the cached bf16 / conv-layout copies, whose keys went stale with the load; (b) closing an OLDER
    GraphedStep must not detach the seed counter a newer one installed."""
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    from hvc import _lib, synthetic
    from hvc.graph import GraphedStep
    import ctypes as C
    cfg = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=64, vit_depth=1, num_heads=2, xray_feature_dim=32)
    xr, ct = synthetic.batch(3, 2, cfg["volume_size"], cfg["xray_img_size"], device="cuda:0")
    crit = DirectRegressionLoss(1.0, 0.5)
    torch.manual_seed(0)
    m = DirectCTRegression(**cfg).to(dev()).eval()                  # eval: no dropout, running BN statistics -> deterministic forward
    torch.manual_seed(1)
    other = DirectCTRegression(**cfg).to(dev()).eval()
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for mod in (m, other):
            for blk in mod.vit_backbone.blocks:
                blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gen) * 0.02)

    def fwd(a, b):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return crit(m(a).float(), b)["total_loss"]

    g = GraphedStep(fwd, [xr, ct], warmup=1)
    try:
        before = g(xr, ct).item()
        m.load_state_dict(other.state_dict())                         # bumps every parameter's version; no eager forward follows
        got = g(xr, ct).item()
        want = fwd(xr, ct).item()                                     # eager: re-casts lazily, certainly on the new weights
        assert got == want and got != before, (before, got, want)
        # (b) an older object's finaliser vs a newer object's counter
        lib = _lib.load()
        g2 = GraphedStep(fwd, [xr, ct], warmup=1)
        try:
            g.close()                                                 # older one goes away AFTER the newer one took over
            probe = torch.zeros(1, dtype=torch.int32, device=dev())
            lib.hvc_clear_seed_counter_if(probe.data_ptr())          # a stranger's pointer: no effect either
            # the newer graph still sees its own counter: its captured advance node and kernels agree -> replays stay valid
            assert g2(xr, ct).item() == want
            assert g2.counter.item() >= 2
        finally:
            g2.close()
    finally:
        g.close()


@pytest.mark.parametrize("shape", [(2, 4, 512, 512, 64, 0.0), (1, 8, 1100, 130, 32, 0.0), (1, 2, 65, 257, 64, 0.0), (2, 8, 2048, 1024, 32, 0.1),
                                   (1, 4, 4096, 4096, 64, 0.1)])
@pytest.mark.parametrize("variant", ["x16", "x64"])
def test_attention_fp8_forward_vs_oracle_and_bf16_kernel(shape, variant, hvc_option):
    """fp8 (e4m3) MFMA attention forward (BASELINE configs[4]; hvc_attention_fwd_fp8) against the fp64 softmax and the bf16
    kernel.  Stated tolerance: relative Frobenius error <= 6e-2 on O for WHITE-NOISE operands - e4m3 keeps 3 mantissa bits
    (rms rounding error 3.7 %), P and V are both rounded, and with zero-mean random V the output is itself a noise average, so
    the two errors add in quadrature to ~5.2 % with no averaging gain (measured 5.26-5.29 %); structured activations average
    the rounding away (block-level golden test below: 5e-2).  LSE (fp8 Q, K only) within 8e-2, the same dropout mask as the
    bf16 kernel, and a backward pass (bf16 kernels on the fp8 forward's o / lse) within 8e-2 of the all-bf16 gradients.
    variant x64 = the round-3 experiment kernel (HVC_FP8_MX=1: 32x32x64 products, reference moved by a vote on the row sums)."""
    from hvc import ops
    hvc_option("HVC_FP8_MX", 1 if variant == "x64" else 0)
    B, H, Nq, Nk, D, p = shape
    g = torch.Generator().manual_seed(Nq * 7 + Nk + D)
    q, k, v, do = (torch.randn(B, n, H, D, generator=g).to(dev(), torch.bfloat16) for n in (Nq, Nk, Nk, Nq))
    scale = D ** -0.5
    o8, lse8 = ops.attention_fwd(q, k, v, scale, p, 11, fp8=True)
    o16, lse16 = ops.attention_fwd(q, k, v, scale, p, 11)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    assert torch.isfinite(o8.float()).all()
    if p == 0.0:
        qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
        s_ = (qd @ kd.transpose(-1, -2)) * scale
        ref = (torch.softmax(s_, -1) @ vd).permute(0, 2, 1, 3)
        assert rel(o8, ref) < 6e-2, rel(o8, ref)
        assert (lse8.double() - torch.logsumexp(s_, -1)).abs().max().item() < 8e-2
    # same function of (seed, b, h, q, k) for the dropout lots: were the masks different, the two outputs would differ by ~sqrt(2 p)
    assert rel(o8, o16) < 7e-2, rel(o8, o16)
    assert (lse8 - lse16).abs().max().item() < 8e-2
    g8 = ops.attention_bwd(q, k, v, o8, do, lse8, scale, p, 11)
    g16 = ops.attention_bwd(q, k, v, o16, do, lse16, scale, p, 11)
    for a_, b_ in zip(g8, g16):
        assert rel(a_, b_) < 8e-2, rel(a_, b_)


@pytest.mark.parametrize("variant", ["x16", "x64"])
@pytest.mark.parametrize("case", ["flat", "ramp_up", "ramp_down", "huge"])
def test_attention_fp8_reference_tracking_cases(case, variant, hvc_option):
    """Score patterns that stress how the fp8 kernels keep their scaled probabilities inside e4m3: flat rows (every probability
    equals the reference: the x64 kernel's row-sum vote must stay silent and the result is the mean of V), scores that climb
    tile after tile (the reference has to move again and again), scores that fall (small probabilities against an early
    maximum) and scores large enough to overflow exp2 against a stale reference.  fp64 softmax, fp8 tolerance (6e-2)."""
    from hvc import ops
    hvc_option("HVC_FP8_MX", 1 if variant == "x64" else 0)
    B, H, Nq, Nk, D = 1, 2, 256, 1024 + 37, 32
    g = torch.Generator().manual_seed(17)
    q = torch.randn(B, Nq, H, D, generator=g)
    k = torch.randn(B, Nk, H, D, generator=g)
    v = torch.randn(B, Nk, H, D, generator=g) + 0.5
    # The pattern rides on channel 0 with values e4m3 holds exactly (fp8 rounds q * scale * log2 e and k to 4 significant bits:
    # a large score carried by inexact operands would be off by whole nats, which is the format, not the reference tracking):
    # scale = 1/4 / log2 e makes q0 = 1 -> 0.25, and k0 = 16 n for n <= 16.
    scale = 0.25 / math.log2(math.e)
    tile = (torch.arange(Nk) // 64).float().view(1, Nk, 1)            # 0 .. 16
    if case == "flat":
        q = torch.zeros_like(q)
    elif case in ("ramp_up", "ramp_down"):                            # 4 log2 units per 64-key tile, 64 over the row
        q[..., 0] = 1.0
        k[..., 0] = 16.0 * (tile if case == "ramp_up" else 16.0 - tile)
    else:                                                             # a jump of 224 log2 units in the middle of the row: exp2 overflows against the old reference
        q[..., 0] = 2.0
        k[..., 0] = 0.0
        k[:, 700:, :, 0] = 448.0
    q, k, v = (t.to(dev(), torch.bfloat16) for t in (q, k, v))
    o8, lse8 = ops.attention_fwd(q, k, v, scale, 0.0, 11, fp8=True)
    qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    s_ = (qd @ kd.transpose(-1, -2)) * scale
    ref = (torch.softmax(s_, -1) @ vd).permute(0, 2, 1, 3)
    assert torch.isfinite(o8.float()).all() and torch.isfinite(lse8).all()
    err = ((o8.double() - ref).norm() / ref.norm()).item()
    _note(f"o/{case}/{variant}", err, 6e-2, "fro")
    assert err < 6e-2, err
    lse_ref = torch.logsumexp(s_, -1)
    assert ((lse8.double() - lse_ref).abs() / lse_ref.abs().clamp_min(1.0)).max().item() < 8e-2


def test_block_with_fp8_attention_vs_golden(golden):
    """HybridViTBlock3D with the fp8 attention switch on (hvc.functional.set_fp8_attention), bf16 autocast, against the reference's
    golden block outputs and gradients at the fp8 tolerance."""
    from models.hybrid_vit_backbone import HybridViTBlock3D
    from hvc import functional as HF
    g = golden("block")
    B, N, M, Cn, Cc, cond_dim, heads = (int(v) for v in g.z["meta"])
    blk = HybridViTBlock3D(Cn, num_heads=heads, context_dim=Cc, cond_dim=cond_dim)
    _load(blk, g.group("params")).eval()
    x, ctx, cond = (g.t(k).to(dev()).requires_grad_(True) for k in ("x", "ctx", "cond"))
    HF.set_fp8_attention(True)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x, ctx, cond)
        (y.float() * g.t("w").to(dev())).sum().backward()
    finally:
        HF.set_fp8_attention(False)
    g.check("", "out", y, 5e-2, metric="l2")
    g.check("igrad", "x", x.grad, 1e-1, metric="l2")
    for k in ("self_attn.qkv.weight", "cross_attn.kv.weight", "mlp.0.weight"):
        g.check("pgrad", k, dict(blk.named_parameters())[k].grad, 1e-1, metric="l2")
