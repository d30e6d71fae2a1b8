"""Dropout-ON parity (the configuration bench.py times: the reference trains with attn_drop / proj_drop / mlp dropout p = 0.1
hard-coded - models/vit_components.py:48, :55, :110, :117; models/hybrid_vit_backbone.py:38, :77, :79).

The HIP kernels draw counter-based masks (not torch's Philox stream), so the oracle cannot draw the same mask by itself.  These
tests RECOVER the keep mask each kernel really applied from kernel outputs - attention: V = one-hot over a chunk of keys makes
O(q, d) != 0 <=> keep(q, key d); GEMM epilogues: an all-ones product is zero exactly where the epilogue dropped it - and feed it
to the fp64 oracle (oracle.dropout: x * keep / (1 - p), torch's formula with that mask).  Then every output of the forward and of
the backward is held to the oracle at the usual bounds: fp32 mode max|d| <= 1e-3 max|ref|, bf16 relative Frobenius <= 4e-2.
Each case pins one shipped DROP instantiation (hvc_set_option): 32-row forward, 64-row forward with 4 and 8 wavefronts, dQ and
dK/dV with 4 and 8 wavefronts, head dims 32 and 64, ragged Nq / Nk, unaligned (non-vector) operands, the query-split dK/dV slabs
of cross-attention, the fp8 forward.  The last tests do the same for a whole HybridViTBlock3D in train mode.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3
BF16_TOL = 4e-2


def dev():
    return torch.device("cuda:0")


def recover_attention_keep(B, H, Nq, Nk, D, p, seed, dtype, fp8=False):
    """(B, H, Nq, Nk) bool keep mask of the attention-probability dropout, read off the forward kernel that is currently pinned:
    zero q / k give uniform probabilities, V = one-hot over D keys at a time, O(b, q, h, d) != 0 <=> key D c + d kept."""
    from hvc import ops
    q = torch.zeros(B, Nq, H, D, device=dev(), dtype=dtype)
    k = torch.zeros(B, Nk, H, D, device=dev(), dtype=dtype)
    keep = torch.zeros(B, H, Nq, Nk, dtype=torch.bool, device=dev())
    for c in range((Nk + D - 1) // D):
        n = min(D, Nk - D * c)
        v = torch.zeros(B, Nk, H, D, device=dev(), dtype=dtype)
        idx = torch.arange(n, device=dev())
        v[:, D * c + idx, :, idx] = 1.0
        o, _ = ops.attention_fwd(q, k, v, D ** -0.5, p, seed, fp8=fp8)
        keep[:, :, :, D * c:D * c + n] = (o[..., :n] != 0).permute(0, 2, 1, 3)
    return keep


def recover_gemm_keep(M, N, p, seed):
    """(M, N) bool keep mask of a GEMM-epilogue dropout: ones x ones^T = 8 everywhere, zero exactly where the epilogue dropped."""
    from hvc import ops
    a = torch.ones(M, 8, device=dev())
    b = torch.ones(N, 8, device=dev())
    return ops.gemm(a, b, p_drop=p, seed=seed, out_dtype=torch.float32) != 0


def _err(got, ref, bf16):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    if bf16:
        return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _note(key, err, tol, metric):
    from conftest import _report
    _report(key, metric, err, tol)


def _unaligned(t):
    """Same values in a buffer whose rows are not 16-byte addressable (head-dim stride 1, row pitch D + 1 elements, offset 1)."""
    B, N, H, D = t.shape
    buf = torch.zeros(B, N, H, D + 1, device=t.device, dtype=t.dtype)
    buf[..., 1:] = t
    return buf[..., 1:]


# (B, H, Nq, Nk, D, dtype, fwd rows, fwd waves, bwd waves, p, unaligned); fwd rows "pipe" = the software-pipelined forward
# (attn_fwdp_kernel, HVC_ATTN_PIPE=2: taken on any grid), Nk chosen to hit every remainder of its four-tile unrolled sweep
CASES = {
    "pipe_bf16_d64_1tile_ragged":  (1, 2, 300, 50, 64, torch.bfloat16, "pipe", 0, 4, 0.1, False),
    "pipe_bf16_d64_2tiles":        (2, 2, 260, 128, 64, torch.bfloat16, "pipe", 0, 4, 0.1, False),
    "pipe_bf16_d64_3tiles_ragged": (1, 3, 515, 190, 64, torch.bfloat16, "pipe", 0, 4, 0.1, False),
    "pipe_bf16_d64_4tiles":        (1, 2, 256, 256, 64, torch.bfloat16, "pipe", 0, 8, 0.1, False),
    "pipe_bf16_d64_5tiles_ragged": (2, 2, 700, 300, 64, torch.bfloat16, "pipe", 0, 8, 0.1, False),
    "pipe_bf16_d64_11tiles":       (1, 4, 1000, 700, 64, torch.bfloat16, "pipe", 0, 8, 0.25, False),
    "pipe_bf16_d32_7tiles_ragged": (1, 8, 530, 401, 32, torch.bfloat16, "pipe", 0, 8, 0.1, False),
    "pipe_bf16_d32_16tiles":       (2, 2, 300, 1024, 32, torch.bfloat16, "pipe", 0, 4, 0.1, False),
    "fwd32_f32_d32_ragged":        (2, 2, 193, 131, 32, torch.float32, 32, 0, 4, 0.1, False),
    "fwd32_f32_d64_ragged":        (1, 3, 130, 257, 64, torch.float32, 32, 0, 4, 0.1, False),
    "fwd32_f32_d64_scalar_loads":  (1, 2, 97, 70, 64, torch.float32, 32, 0, 4, 0.25, True),
    "fwd32_bf16_d64_ragged":       (2, 2, 300, 190, 64, torch.bfloat16, 32, 0, 4, 0.1, False),
    "fwd32_bf16_d32_scalar_loads": (1, 2, 140, 75, 32, torch.bfloat16, 32, 0, 4, 0.1, True),
    "fwd64w4_bf16_d64_bwd4":       (2, 4, 700, 515, 64, torch.bfloat16, 64, 4, 4, 0.1, False),
    "fwd64w8_bf16_d64_bwd8":       (2, 2, 900, 515, 64, torch.bfloat16, 64, 8, 8, 0.1, False),
    "fwd64w8_bf16_d64_even":       (1, 2, 1024, 512, 64, torch.bfloat16, 64, 8, 8, 0.1, False),
    "fwd64w4_bf16_d32_bwd8":       (1, 8, 1030, 1100, 32, torch.bfloat16, 64, 4, 8, 0.1, False),
    "fwd64w4_bf16_d32_bwd4":       (2, 3, 520, 330, 32, torch.bfloat16, 64, 4, 4, 0.25, False),
    "cross_qsplit_bf16_d32_bwd4":  (1, 2, 2048, 64, 32, torch.bfloat16, 64, 4, 4, 0.1, False),
    "cross_qsplit_bf16_d64_bwd8":  (1, 2, 2100, 250, 64, torch.bfloat16, 64, 8, 8, 0.1, False),
    "cross_qsplit_f32_d64":        (1, 1, 1100, 130, 64, torch.float32, 32, 0, 4, 0.1, False),
}


@pytest.mark.parametrize("case", list(CASES))
def test_attention_dropout_on_vs_masked_fp64_oracle(case, hvc_option):
    """O, LSE, dQ, dK, dV of one pinned set of DROP instantiations against the fp64 oracle evaluated WITH the kernels' own keep
    mask (reference: softmax -> nn.Dropout -> @ v, models/vit_components.py:46-51 / :103-113)."""
    from hvc import ops
    from oracle import hvc_oracle as O
    B, H, Nq, Nk, D, dtype, rows, fwaves, bwaves, p, unaligned = CASES[case]
    bf16 = dtype == torch.bfloat16
    if rows == "pipe":
        hvc_option("HVC_ATTN_FWD_ROWS", 0)
        hvc_option("HVC_ATTN_PIPE", 2)
    else:
        hvc_option("HVC_ATTN_FWD_ROWS", rows)
        hvc_option("HVC_ATTN_PIPE", 0)
    hvc_option("HVC_ATTN_FWD_WAVES", fwaves)
    hvc_option("HVC_ATTN_BWD_WAVES", bwaves)
    if case.startswith("cross_qsplit"):      # few key blocks: dK / dV come from query-range slices summed in a second pass
        from hvc import _lib
        assert _lib.load().hvc_attention_bwd_workspace(B, H, Nq, Nk, D) > B * H * Nq, "shape no longer takes the query-split path"
    seed = 1000 + sum(CASES[case][:5])
    g = torch.Generator().manual_seed(seed)
    q, k, v, do = (torch.randn(B, n, H, D, generator=g).to(dev(), dtype) for n in (Nq, Nk, Nk, Nq))
    keep = recover_attention_keep(B, H, Nq, Nk, D, p, seed, dtype)
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 5 * (p * (1 - p) / keep.numel()) ** 0.5 + 2e-3, rate
    # fp64 oracle with that mask, on the operands as the kernels see them (bf16-rounded in bf16 mode)
    qr, kr, vr = (t.double().cpu().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))
    o_ref = O.attention_core(qr, kr, vr, D ** -0.5, p_drop=p, keep=keep.cpu())
    o_ref.backward(do.double().cpu().permute(0, 2, 1, 3))
    lse_ref = torch.logsumexp((qr.detach() @ kr.detach().transpose(-2, -1)) * D ** -0.5, dim=-1)      # (B, H, Nq)
    if unaligned:
        q, k, v = _unaligned(q), _unaligned(k), _unaligned(v)
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, seed)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, seed)
    tol, metric = (BF16_TOL, "l2") if bf16 else (F32_TOL, "max")
    for name, got, ref in (("o", o, o_ref.permute(0, 2, 1, 3)), ("dq", dq, qr.grad.permute(0, 2, 1, 3)),
                           ("dk", dk, kr.grad.permute(0, 2, 1, 3)), ("dv", dv, vr.grad.permute(0, 2, 1, 3))):
        e = _err(got, ref, bf16)
        _note(f"{case}/{name}", e, tol, metric)
        assert e <= tol, (case, name, e)
    e = ((lse.double().cpu().view(B, H, Nq) - lse_ref).abs().max() / lse_ref.abs().max().clamp_min(1.0)).item()
    _note(f"{case}/lse", e, 2e-2 if bf16 else F32_TOL, "max")
    assert e <= (2e-2 if bf16 else F32_TOL), (case, "lse", e)


@pytest.mark.parametrize("variant", ["x16", "x64"])
@pytest.mark.parametrize("shape", [(2, 2, 520, 300, 64), (1, 4, 300, 1030, 32)])
def test_attention_fp8_forward_dropout_mask_and_output_vs_masked_oracle(shape, variant, hvc_option):
    """The fp8 (e4m3) forward draws the same lots as the bf16 kernels (its backward IS the bf16 kernels): its recovered mask must
    equal the bf16 forward's, and its output must match the masked fp64 oracle at the fp8 bound of DESIGN 5.1.2 (6e-2)."""
    from hvc import ops
    from oracle import hvc_oracle as O
    B, H, Nq, Nk, D = shape
    p, seed = 0.1, 4242 + Nq
    hvc_option("HVC_FP8_MX", 1 if variant == "x64" else 0)
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(B, n, H, D, generator=g).to(dev(), torch.bfloat16) for n in (Nq, Nk, Nk))
    keep8 = recover_attention_keep(B, H, Nq, Nk, D, p, seed, torch.bfloat16, fp8=True)
    keep16 = recover_attention_keep(B, H, Nq, Nk, D, p, seed, torch.bfloat16)
    assert torch.equal(keep8, keep16)
    ref = O.attention_core(*(t.double().cpu().permute(0, 2, 1, 3) for t in (q, k, v)), D ** -0.5, p_drop=p, keep=keep8.cpu())
    o, _ = ops.attention_fwd(q, k, v, D ** -0.5, p, seed, fp8=True)
    e = _err(o, ref.permute(0, 2, 1, 3), True)
    _note(f"fp8_{variant}_{D}/o", e, 6e-2, "l2")
    assert e <= 6e-2, e


def _block_seeds(torch_seed):
    """The six seeds HybridViTBlock3D.forward draws in train mode, in its order (self-attention: probabilities, proj output;
    cross-attention: probabilities, proj output; MLP: fc1 output, fc2 output)."""
    from hvc import functional as HF
    torch.manual_seed(torch_seed)
    return [HF.new_seed() for _ in range(6)]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", ["golden_d32", "d64_pinned_large_forms", "d64_phase_separated_64row"])
def test_block_train_mode_dropout_on_vs_masked_oracle(golden, mode, cfg, hvc_option):
    """HybridViTBlock3D in TRAIN mode, p = 0.1 on all six dropouts (models/hybrid_vit_backbone.py:38, :77, :79, :117-139): the six
    masks the HIP chain drew are recovered (attention masks by one-hot V under the block's own seeds, GEMM-epilogue masks from
    the zeros of an all-ones product under the same seeds) and the whole block - output, every input gradient, every parameter
    gradient - is held to the oracle evaluated with those masks."""
    from models.hybrid_vit_backbone import HybridViTBlock3D
    from oracle import hvc_oracle as O
    p = 0.1
    if cfg == "golden_d32":
        gd = golden("block")
        B, N, M, Cn, Cc, cond_dim, heads = (int(v) for v in gd.z["meta"])
        blk = HybridViTBlock3D(Cn, num_heads=heads, context_dim=Cc, cond_dim=cond_dim)
        blk.load_state_dict(gd.group("params"), strict=True)
        x, ctx, cond, w = (gd.t(k) for k in ("x", "ctx", "cond", "w"))
    else:
        B, N, M, Cn, Cc, cond_dim, heads = 2, 600, 130, 128, 96, 48, 2
        torch.manual_seed(5)
        blk = HybridViTBlock3D(Cn, num_heads=heads, context_dim=Cc, cond_dim=cond_dim)
        gg = torch.Generator().manual_seed(6)
        with torch.no_grad():      # AdaLN is zero-initialised in the reference: make the gated branches visible
            blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=gg) * 0.05)
            blk.adaln.linear.bias.copy_(torch.randn(blk.adaln.linear.bias.shape, generator=gg) * 0.5)
        x, ctx, cond, w = (torch.randn(*s, generator=gg) for s in ((B, N, Cn), (B, M, Cc), (B, cond_dim), (B, N, Cn)))
        if cfg == "d64_phase_separated_64row":
            hvc_option("HVC_ATTN_PIPE", 0)
            hvc_option("HVC_ATTN_FWD_ROWS", 64)
            hvc_option("HVC_ATTN_FWD_WAVES", 8)
        else:
            hvc_option("HVC_ATTN_PIPE", 2)        # the forms the 128^3 benchmark runs: pipelined forward, 8-wavefront backward workgroups
        hvc_option("HVC_ATTN_BWD_WAVES", 8)
    D = Cn // heads
    bf16 = mode == "bf16"
    cdt = torch.bfloat16 if bf16 else torch.float32
    blk = blk.to(dev()).train()
    assert blk.self_attn.attn_drop.p == p and blk.mlp[2].p == p
    xd, cd, dd = (t.to(dev()).requires_grad_(True) for t in (x, ctx, cond))
    TS = 777
    torch.manual_seed(TS)
    if bf16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xd, cd, dd)
    else:
        y = blk(xd, cd, dd)
    (y * w.to(dev())).sum().backward()
    s = _block_seeds(TS)
    keeps = {
        "sa_attn": recover_attention_keep(B, heads, N, N, D, p, s[0], cdt).cpu(),
        "sa_proj": recover_gemm_keep(B * N, Cn, p, s[1]).view(B, N, Cn).cpu(),
        "ca_attn": recover_attention_keep(B, heads, N, M, D, p, s[2], cdt).cpu(),
        "ca_proj": recover_gemm_keep(B * N, Cn, p, s[3]).view(B, N, Cn).cpu(),
        "fc1": recover_gemm_keep(B * N, 4 * Cn, p, s[4]).view(B, N, 4 * Cn).cpu(),
        "fc2": recover_gemm_keep(B * N, Cn, p, s[5]).view(B, N, Cn).cpu(),
    }
    for name, kp in keeps.items():
        assert abs(kp.float().mean().item() - (1 - p)) < 0.02, name
    P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in blk.state_dict().items()}
    xr, cr, dr = (t.double().requires_grad_(True) for t in (x, ctx, cond))
    y_ref = O.vit_block(xr, cr, dr, P, "", heads, p_drop=p, keeps=keeps)
    (y_ref * w.double()).sum().backward()
    tol, metric = (BF16_TOL, "l2") if bf16 else (F32_TOL, "max")
    checks = [("out", y, y_ref), ("igrad/x", xd.grad, xr.grad), ("igrad/ctx", cd.grad, cr.grad), ("igrad/cond", dd.grad, dr.grad)]
    checks += [("pgrad/" + k, prm.grad, P[k].grad) for k, prm in blk.named_parameters()]
    for name, got, ref in checks:
        e = _err(got, ref, bf16)
        _note(f"{cfg}/{name}", e, tol, metric)
        assert e <= tol, (cfg, mode, name, e)


@pytest.mark.parametrize("factor", [3.0, 7.0, 30.0])
@pytest.mark.parametrize("D", [64, 32])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_pipelined_forward_late_score_spikes(factor, D, p, hvc_option):
    """The software-pipelined forward fixes a row's reference exponent at tile 0 and never moves it: a later score far above it
    makes probabilities of up to 2^60 times the tile-0 scale (factor 3: ~2^30, the relative precision of P, of the row sum and of O
    must not suffer), and past the fp32 range the wavefront must notice (sticky maximum of the tile sums) and redo its rows with
    the classic online softmax (factors 7 and 30: ~2^77 and overflow to inf).  O and LSE against the fp64 oracle (with the
    kernel's own keep mask when dropout is on); the spiked rows and their plain neighbours are both checked."""
    from hvc import ops
    from oracle import hvc_oracle as O
    B, H, Nq, Nk = 1, 2, 300, 450
    hvc_option("HVC_ATTN_PIPE", 2)
    g = torch.Generator().manual_seed(int(factor) * 100 + D)
    q, k, v = (torch.randn(B, n, H, D, generator=g) for n in (Nq, Nk, Nk))
    for row, key in ((17, 200), (140, 449), (299, 70)):             # tiles 3, 7 (ragged last), 1; three different wavefronts
        k[0, key, 0] = q[0, row, 0] * factor
    k[0, 5, 1] = q[0, 33, 1] * factor                                # and one in tile 0 itself (adopted as the reference: no redo)
    q, k, v = (t.to(dev(), torch.bfloat16) for t in (q, k, v))
    seed = 99
    keep = recover_attention_keep(B, H, Nq, Nk, D, p, seed, torch.bfloat16).cpu() if p > 0 else None
    qr, kr, vr = (t.double().cpu().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = O.attention_core(qr, kr, vr, D ** -0.5, p_drop=p, keep=keep).permute(0, 2, 1, 3)
    lse_ref = torch.logsumexp((qr @ kr.transpose(-2, -1)) * D ** -0.5, dim=-1)
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, seed)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    e = _err(o, ref, True)
    _note(f"spike{factor}/d{D}/p{p}/o", e, BF16_TOL, "l2")
    assert e <= BF16_TOL, e
    rows = _err(o[0, [17, 140, 299]], ref[0, [17, 140, 299]], True)
    assert rows <= BF16_TOL, rows
    e = ((lse.double().cpu().view(B, H, Nq) - lse_ref).abs() / lse_ref.abs().clamp_min(1.0)).max().item()
    _note(f"spike{factor}/d{D}/p{p}/lse", e, 2e-2, "max")
    assert e <= 2e-2, e
    # the phase-separated kernels (reference moved whenever a score outgrows it by 2^6) must agree with it
    hvc_option("HVC_ATTN_PIPE", 0)
    o2, lse2 = ops.attention_fwd(q, k, v, D ** -0.5, p, seed)
    assert _err(o, o2, True) <= 1e-2 and (lse - lse2).abs().max().item() <= 2e-2 * max(1.0, lse2.abs().max().item())
