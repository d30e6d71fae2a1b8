"""Pins oracle/hvc_oracle.py to the reference: every function is compared with golden vectors that
tests/golden/make_golden.py captured from the imported reference (outputs AND gradients)."""
import numpy as np
import pytest
import torch

from oracle import hvc_oracle as O

RTOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order only


def _pre_bn_bias(key):
    """Conv biases directly ahead of a train-mode BatchNorm have an exactly-zero gradient (BN removes
    the per-channel mean); reference and oracle both hold only rounding noise there."""
    return key.endswith(("encoder.0.bias", "encoder.4.bias", "encoder.8.bias"))


def _req(d):
    return {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k) for k, v in d.items()}


def test_attention_modules(golden):
    g = golden("attention")
    B, N, M, Cn, Cc, heads = (int(v) for v in g.z["meta"])
    x = g.t("x").requires_grad_(True)
    ctx = g.t("ctx").requires_grad_(True)
    P = _req(g.group("sa_params"))
    y = O.self_attention(x, {"sa." + k: v for k, v in P.items()}, "sa.", heads)
    g.check("", "sa_out", y, RTOL)
    (y * g.t("w_sa")).sum().backward()
    g.check("sa_igrad", "x", x.grad, RTOL)
    for k, v in P.items():
        g.check("sa_pgrad", k, v.grad, RTOL)
    x.grad = None
    P = _req(g.group("ca_params"))
    y = O.cross_attention(x, ctx, {"ca." + k: v for k, v in P.items()}, "ca.", heads)
    g.check("", "ca_out", y, RTOL)
    (y * g.t("w_ca")).sum().backward()
    g.check("ca_igrad", "x", x.grad, RTOL)
    g.check("ca_igrad", "ctx", ctx.grad, RTOL)
    for k, v in P.items():
        g.check("ca_pgrad", k, v.grad, RTOL)
    # chunked evaluation is the same arithmetic per row
    with torch.no_grad():
        y2 = O.cross_attention(x, ctx, {"ca." + k: v for k, v in P.items()}, "ca.", heads, q_chunk=7)
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-6)


def test_block(golden):
    g = golden("block")
    B, N, M, Cn, Cc, cond_dim, heads = (int(v) for v in g.z["meta"])
    x, ctx, cond = (g.t(k).requires_grad_(True) for k in ("x", "ctx", "cond"))
    P = _req(g.group("params"))
    y = O.vit_block(x, ctx, cond, {"b." + k: v for k, v in P.items()}, "b.", heads)
    g.check("", "out", y, RTOL)
    (y * g.t("w")).sum().backward()
    for k, t in (("x", x), ("ctx", ctx), ("cond", cond)):
        g.check("igrad", k, t.grad, RTOL)
    for k in g.keys("pgrad"):
        g.check("pgrad", k, P[k].grad, RTOL)


@pytest.mark.parametrize("tag", ["8", "32", "64x32x32"])
def test_hybrid_vit3d(golden, tag):
    g = golden("vit3d_" + tag)
    meta = [int(v) for v in g.z["meta"]]
    B, M, Cn, Cc, cond_dim, heads, depth, in_ch = meta[:8]
    vs, ds = tuple(meta[8:11]), tuple(meta[11:14])
    layers, ref_ds, grid = O.voxel_embed_plan(vs, in_ch, Cn)
    assert ref_ds == ds == grid          # reference geometry is self-consistent at these sizes
    x, ctx, cond = (g.t(k).requires_grad_(True) for k in ("x", "ctx", "cond"))
    P = _req(g.group("params"))
    y = O.hybrid_vit3d(x, ctx, cond, {"m." + k: v for k, v in P.items()}, "m.", vs, in_ch, Cn, depth, heads)
    g.check("", "out", y, RTOL)
    (y * g.t("w")).sum().backward()
    for k, t in (("x", x), ("ctx", ctx), ("cond", cond)):
        g.check("igrad", k, t.grad, RTOL, 5)
    for k in g.keys("pgrad"):
        g.check("pgrad", k, P[k].grad, RTOL, 5)


def test_voxel_embed_plan_matches_reference_geometry():
    # reference formula (models/hybrid_vit_backbone.py:178-188) and the 128^3 inconsistency
    assert O.voxel_embed_plan((64, 64, 64), 1, 256)[1:] == ((16, 16, 16), (16, 16, 16))
    assert O.voxel_embed_plan((256, 256, 256), 32, 256)[1:] == ((32, 32, 32), (32, 32, 32))
    layers, ref_ds, grid = O.voxel_embed_plan((128, 128, 128), 1, 256)
    assert ref_ds == (25, 25, 25) and grid == (32, 32, 32)
    assert O.voxel_embed_plan((128, 128, 128), 1, 256, token_grid=16)[2] == (16, 16, 16)
    convs = [l for l in O.voxel_embed_plan((64, 64, 64), 1, 256)[0] if l[0] == "conv"]
    assert convs == [("conv", 1, 64, 2), ("conv", 64, 128, 2), ("conv", 128, 256, 1)]
    convs = [l for l in O.voxel_embed_plan((256, 256, 256), 32, 256)[0] if l[0] == "conv"]
    assert convs == [("conv", 32, 64, 2), ("conv", 64, 128, 2), ("conv", 128, 256, 2)]


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_xray_conditioning(golden, mode):
    g = golden("xray_cond")
    xr, t = g.t("xrays").requires_grad_(True), g.t("t").requires_grad_(True)
    P = _req(g.group(f"{mode}_params"))
    new_stats = {}
    ctx, cond, feats = O.xray_conditioning(xr, t, {"e." + k: v for k, v in P.items()}, "e.", mode == "train", new_stats)
    g.check("", f"{mode}_ctx", ctx, RTOL)
    g.check("", f"{mode}_cond", cond, RTOL)
    g.check("", f"{mode}_feats", feats, RTOL)
    ((ctx * g.t("w_ctx")).sum() + (cond * g.t("w_cond")).sum() + (feats * g.t("w_f")).sum()).backward()
    g.check("", f"{mode}_dxr", xr.grad, RTOL, 5)
    g.check("", f"{mode}_dt", t.grad, RTOL)
    for k in g.keys(f"{mode}_pgrad"):
        if mode == "train" and _pre_bn_bias(k):
            assert P[k].grad.abs().max() < 1e-4
            continue
        g.check(f"{mode}_pgrad", k, P[k].grad, RTOL, 10)
    if mode == "train":
        for k, v in new_stats.items():
            g.check("train_stats_after", k[2:], v, RTOL)


def test_drr_known_answers(golden):
    g = golden("drr")
    vol = g.t("vol")
    ap, lat = O.drr_render(vol.squeeze(1), 0), O.drr_render(vol.squeeze(1), 90)
    g.check("", "ap", ap, 1e-6)
    g.check("", "lat", lat, 1e-6)
    # SURVEY.md §9 values measured on the reference
    assert ap.shape == (2, 6, 4) and lat.shape == (2, 6, 8)
    assert abs(ap.sum().item() - 285.885590) < 1e-3 and abs(ap[0, 0, 0].item() - 6.184685) < 1e-5
    assert abs(lat[0, 0, 0].item() - 3.426665) < 1e-5
    pl = O.projection_loss(vol, g.t("xr"), 0)
    assert abs(pl.item() - 35.666653) < 1e-3
    g.check("", "proj_loss_0", pl, 1e-6)
    g.check("", "proj_loss_90", O.projection_loss(vol, g.t("xr"), 90), 1e-6)
    v = vol.clone().requires_grad_(True)
    l = O.drr_reprojection_loss(v, g.t("xr2"), 16)
    assert abs(l.item() - 0.516497) < 1e-5
    l.backward()
    g.check("", "reproj_dvol", v.grad, 1e-5)
    g.check("", "reproj_ap", O.mean_projection(vol, 0, 16), 1e-6)
    g.check("", "reproj_lat", O.mean_projection(vol, 90, 16), 1e-6)
    v = vol.clone().requires_grad_(True)
    (O.projection_loss(v, g.t("xr"), 0) + O.projection_loss(v, g.t("xr"), 90)).backward()
    g.check("", "proj_dvol", v.grad, 1e-5)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_direct_regression_small(golden, mode):
    g = golden("direct_small")
    cfg = [int(v) for v in g.z["cfg"]]
    P = _req(g.group("params"))
    xr = g.t("xrays").requires_grad_(True)
    new_stats = {}
    pred = O.direct_ct_regression(xr, P, tuple(cfg[:3]), cfg[4], cfg[5], cfg[6], training=(mode == "train"), new_stats=new_stats)
    g.check("", f"{mode}_pred", pred, RTOL, 5)
    losses = O.direct_regression_loss(pred, g.t("target"))
    ref = g.z[f"{mode}_loss"]
    got = [losses[k].item() for k in ("total_loss", "l1_loss", "ssim_loss")]
    assert np.allclose(got, ref, rtol=1e-5)
    assert abs(O.psnr(pred.detach(), g.t("target")) - float(g.z[f"{mode}_psnr"])) < 1e-3
    losses["total_loss"].backward()
    g.check("", f"{mode}_dxr", xr.grad, RTOL, 20)
    for k in g.keys(f"{mode}_pgrad"):
        if mode == "train" and _pre_bn_bias(k):
            assert P[k].grad.abs().max() < 1e-6
            continue
        g.check(f"{mode}_pgrad", k, P[k].grad, RTOL, 20)
    if mode == "train":
        for k, v in new_stats.items():
            g.check("train_stats_after", k, v, RTOL)


def test_cascade_refiners_small(golden):
    g = golden("cascade_small")
    P = {}
    for pre in ("enc", "s2", "s3"):
        P.update({f"{pre}.{k}": v for k, v in _req(g.group(f"{pre}_params")).items()})
    xr = g.t("xrays")
    v16 = g.t("v16").requires_grad_(True)
    f1, _, _ = O.multiscale_xray_encoder(xr, P, "enc.", 1)
    f2, cond2, _ = O.multiscale_xray_encoder(xr, P, "enc.", 2)
    v32 = O.stage2_refiner(v16, f2, cond2, P, "s2.", (32, 32, 32), 32, 1, 1)
    f3, cond3, _ = O.multiscale_xray_encoder(xr, P, "enc.", 3)
    v64 = O.stage3_refiner(v32, f3, cond3, P, "s3.", (64, 64, 64), 32, 1, 1)
    g.check("", "feats1", f1, RTOL, 5)
    g.check("", "feats2", f2, RTOL, 5)
    g.check("", "v32", v32, RTOL, 5)
    g.check("", "v64", v64, RTOL, 5)
    ((v32 * g.t("w2")).sum() + (v64 * g.t("w3")).sum() + f1.sum() * 0.1).backward()
    g.check("", "dv16", v16.grad, RTOL, 20)
    for pre in ("enc", "s2", "s3"):
        for k in g.keys(f"{pre}_pgrad"):
            g.check(f"{pre}_pgrad", k, P[f"{pre}.{k}"].grad, RTOL, 50)


def test_total_variation_loss(golden):
    g = golden("tv")
    for target, tag in ((None, "tv_pred"), (g.t("target"), "tv_match")):
        p = g.t("pred").requires_grad_(True)
        loss = O.total_variation_loss(p, target)
        g.check("", tag, loss, RTOL)
        loss.backward()
        g.check("", tag + "_grad", p.grad, RTOL)
    g.check("", "tv_flat", O.total_variation_loss(torch.full((1, 1, 4, 5, 6), 0.25)), RTOL)


def frequency_loss_oracle(pred, target, high_freq_weight=2.0):
    """FrequencyLoss.forward restated (direct_regression/progressive_cascade/loss_multiscale.py:203-236) in closed form:
    (1/N) sum_f w_f | |P_f| - |T_f| |, w_f = 1 inside radius min(D,H,W)//4 of index (D//2,H//2,W//2), else the weight."""
    P = torch.fft.fftn(pred.double(), dim=(-3, -2, -1)).abs()
    T = torch.fft.fftn(target.double(), dim=(-3, -2, -1)).abs()
    D, H, W = pred.shape[-3:]
    dd, hh, ww = torch.meshgrid(torch.arange(D) - D // 2, torch.arange(H) - H // 2, torch.arange(W) - W // 2, indexing="ij")
    high = (torch.sqrt((dd ** 2 + hh ** 2 + ww ** 2).double()) > min(D, H, W) // 4)
    w = torch.where(high, torch.tensor(float(high_freq_weight), dtype=torch.float64), torch.tensor(1.0, dtype=torch.float64))
    return ((P - T).abs() * w).mean()


def test_frequency_loss_closed_form(golden):
    g = golden("frequency")
    for tag, w in (("w2", 2.0), ("w05", 0.5)):
        p = g.t("pred").requires_grad_(True)
        loss = frequency_loss_oracle(p, g.t("target"), w)
        g.check("", f"loss_{tag}", loss, 1e-5)
        loss.backward()
        g.check("", f"grad_{tag}", p.grad, 1e-4)


def test_sinusoidal_time_embedding(golden):
    g = golden("time_embedding")
    t = g.t("t")
    # sin/cos of arguments up to ~1000 rad: fp32 argument rounding alone is 6e-5 absolute
    assert torch.allclose(O.sinusoidal_time_embedding(t, 32), g.t("emb32"), rtol=0, atol=2e-4)
    assert torch.allclose(O.sinusoidal_time_embedding(t, 256), g.t("emb256"), rtol=0, atol=2e-4)


def test_train_step_two_steps(golden):
    """The oracle driven through the reference's step (direct_regression/train_direct_4gpu.py:59-75) reproduces the
    reference's loss, pre-clip gradient norm, clipped gradients and post-AdamW weights for two consecutive steps."""
    g = golden("train_step")
    cfg = [int(v) for v in g.z["cfg"]]
    P = _req(g.group("params"))
    leaves = {k: v for k, v in P.items() if v.requires_grad}
    opt = torch.optim.AdamW(list(leaves.values()), lr=1e-4, weight_decay=0.01)
    xr, target = g.t("xrays"), g.t("target")
    for step in (1, 2):
        opt.zero_grad()
        stats = {}
        before = {k: v.detach().clone() for k, v in leaves.items()}
        pred = O.direct_ct_regression(xr, P, tuple(cfg[:3]), cfg[4], cfg[5], cfg[6], training=True, new_stats=stats)
        loss = O.direct_regression_loss(pred, target)["total_loss"]
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(list(leaves.values()), 1.0)
        assert abs(loss.item() - float(g.z[f"step{step}_loss"])) < 1e-5
        assert abs(norm.item() - float(g.z[f"step{step}_gradnorm"])) < 1e-4 * float(g.z[f"step{step}_gradnorm"])
        for k, v in leaves.items():
            if not _pre_bn_bias(k):
                g.check(f"step{step}_clipped", k, v.grad, 1e-4)
        opt.step()
        with torch.no_grad():
            for k, v in stats.items():
                P[k].copy_(v)
        for k, v in leaves.items():
            # AdamW moves every weight by ~lr: compare the MOVE (sign-like on step 1), skipping the zero-gradient biases
            if _pre_bn_bias(k):
                continue
            g.check_step_move("params" if step == 1 else "step1_after", f"step{step}_after", k, before[k], v, 1e-4, 0.0, 0.01) \
                if step == 1 else None
            g.check(f"step{step}_after", k, v, 1e-3)          # weights themselves, relative to max|w|
        for k in stats:
            g.check(f"step{step}_after", k, P[k], 1e-4)


def test_xray_encoder_explicit_routing_is_the_same_function(golden):
    """oracle._relu_pool: replaying an evaluation's own ReLU / max-pool routing reproduces its outputs and gradients,
    so the routed form used by the GPU gradient test is the reference's function wherever the routing agrees."""
    g = golden("xray_cond")
    outs = []
    route_use = None
    for _ in range(2):
        P = _req(g.group("train_params"))
        xr = g.t("xrays").requires_grad_(True)
        route = {} if route_use is None else {"use": route_use}
        ctx, cond, feats = O.xray_conditioning(xr, g.t("t"), P, "", True, {}, route)
        ((ctx * g.t("w_ctx")).sum() + (cond * g.t("w_cond")).sum() + (feats * g.t("w_f")).sum()).backward()
        outs.append((feats.detach(), xr.grad.clone(), P["encoder.0.weight"].grad.clone()))
        route_use = route["own"]
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    g.check("", "train_dxr", outs[1][1], RTOL * 5)


def test_direct128_probe_fixture_matches_the_seeded_model(golden):
    """tests/golden/direct128_probes.npz (oracle-generated: the reference raises at 128^3) belongs to the model the GPU test
    rebuilds: same seeded weights (checksum), one probe / full entry per parameter, finite values."""
    import numpy as np
    from tests.golden.make_direct128_probes import build_inputs, weight_checksum
    g = golden("direct128_probes")
    m, xr, ct = build_inputs()
    assert np.allclose(weight_checksum(m.state_dict()), g.z["weights_checksum"], rtol=1e-9)
    assert xr.shape == (1, 2, 1, 512, 512) and ct.shape == (1, 1, 128, 128, 128)
    names = {k for k, p in m.named_parameters()}
    assert names == set(g.keys("pgrad")), names ^ set(g.keys("pgrad"))
    assert m.vit_backbone.pos_embed.shape[1] == 32 ** 3                       # the A2-fix geometry: 32^3 = 32768 tokens
    for k in g.z.files:
        assert np.isfinite(g.z[k]).all(), k
    assert abs(float(g.z["loss/total_loss"]) - (float(g.z["loss/l1_loss"]) + 0.5 * float(g.z["loss/ssim_loss"]))) < 1e-6


def test_oracle_dropout_with_a_given_mask_is_torch_dropout_with_that_mask():
    """oracle.dropout(x, p, keep) is what the dropout-on GPU parity tests feed the kernels' recovered masks through: with the
    mask torch itself drew (read off the zeros of F.dropout's output) it must reproduce F.dropout bit for bit - the formula of
    nn.Dropout in train mode (reference: models/vit_components.py:48, :55; models/hybrid_vit_backbone.py:77, :79)."""
    import torch.nn.functional as F
    from oracle import hvc_oracle as O
    torch.manual_seed(3)
    x = torch.randn(7, 33, 5) + 3.0                       # no exact zeros in x
    for p in (0.1, 0.25, 0.5):
        y = F.dropout(x, p, True)
        keep = y != 0
        assert torch.equal(O.dropout(x, p, keep), y)
        assert torch.allclose(O.dropout(x.double(), p, keep), y.double(), rtol=1e-6)
    # attention_core with torch's own mask recovered the same way == the unmasked-path call under the same generator state
    q, k, v = (torch.randn(1, 2, 9, 8) for _ in range(3))
    torch.manual_seed(11)
    ref = O.attention_core(q, k, v, 8 ** -0.5, p_drop=0.25)
    torch.manual_seed(11)
    probs = ((q @ k.transpose(-2, -1)) * 8 ** -0.5).softmax(dim=-1)
    keep = F.dropout(probs, 0.25, True) != 0
    assert torch.allclose(O.attention_core(q, k, v, 8 ** -0.5, p_drop=0.25, keep=keep), ref, atol=1e-6)
