"""Developer check: every C-ABI kernel against plain torch math ON THE GPU (fast iteration on a
gpurun box).  The formal parity tests (tests/, -m gpu) compare against oracle/ on the CPU."""
import math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
fails = []

def rel(a, b):
    a = a.float(); b = b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()

def report(name, err, tol):
    ok = err <= tol and err == err
    print(f"{'OK  ' if ok else 'FAIL'} {name:58s} err={err:.3e} tol={tol:.1e}", flush=True)
    if not ok:
        fails.append(name)

def ref_attn(q, k, v, scale):
    # (B,N,H,D) fp32
    qh, kh, vh = (t.permute(0, 2, 1, 3).double() for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * scale
    p = s.softmax(-1)
    o = (p @ vh).permute(0, 2, 1, 3)
    return o, torch.logsumexp(s, -1)

def attn_case(B, H, Nq, Nk, D, dtype, packed):
    scale = D ** -0.5
    if packed and Nq == Nk:
        qkv = torch.randn(B, Nq, 3, H, D, device=dev)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = torch.randn(B, Nq, H, D, device=dev); k = torch.randn(B, Nk, H, D, device=dev); v = torch.randn(B, Nk, H, D, device=dev)
    if packed and Nq == Nk:
        qkv_t = qkv.to(dtype); qt, kt, vt = qkv_t[:, :, 0], qkv_t[:, :, 1], qkv_t[:, :, 2]
    else:
        qt, kt, vt = q.to(dtype), k.to(dtype), v.to(dtype)
    qr, kr, vr = (t.float().detach().clone().requires_grad_(True) for t in (qt, kt, vt))
    o_ref, lse_ref = ref_attn(qr, kr, vr, scale)
    do = torch.randn(o_ref.shape, device=dev)
    o_ref.backward(do.double())
    o, lse = ops.attention_fwd(qt, kt, vt, scale)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
    tag = f"attn B{B} H{H} Nq{Nq} Nk{Nk} D{D} {str(dtype)[6:]}{' packed' if packed else ''}"
    report(tag + " o", rel(o, o_ref), tol)
    report(tag + " lse", rel(lse, lse_ref), 1e-4 if dtype == torch.float32 else 2e-3)
    dot = do.to(dtype)
    if packed and Nq == Nk:
        dqkv = torch.empty_like(qkv_t)
        dq, dk, dv = ops.attention_bwd(qt, kt, vt, o, dot, lse, scale, dq=dqkv[:, :, 0], dk=dqkv[:, :, 1], dv=dqkv[:, :, 2])
    else:
        dq, dk, dv = ops.attention_bwd(qt, kt, vt, o, dot, lse, scale)
    report(tag + " dq", rel(dq, qr.grad), tol)
    report(tag + " dk", rel(dk, kr.grad), tol)
    report(tag + " dv", rel(dv, vr.grad), tol)

for dtype in (torch.float32, torch.bfloat16):
    attn_case(1, 1, 32, 64, 64, dtype, False)
    attn_case(2, 2, 24, 10, 32, dtype, False)
    attn_case(2, 4, 200, 333, 64, dtype, False)
    attn_case(1, 2, 256, 256, 32, dtype, True)
    attn_case(1, 8, 130, 130, 64, dtype, True)

def gemm_case(M, N, K, akm, bkm, dtype, out_dtype, epi):
    A = torch.randn(M, K, device=dev); Bm = torch.randn(N, K, device=dev)
    At, Bt = A.to(dtype), Bm.to(dtype)
    a_in = At.t().contiguous() if akm else At
    b_in = Bt.t().contiguous() if bkm else Bt
    ref = At.double() @ Bt.double().t()
    kw = {}
    if epi:
        nb = 2 if M % 2 == 0 else 1
        bias = torch.randn(N, device=dev); gate = torch.randn(nb, N, device=dev); res = torch.randn(M, N, device=dev)
        aux = torch.empty(M, N, dtype=out_dtype, device=dev)
        pre = ref * 0.5 + bias.double()
        ref = res.double() + gate.double().repeat_interleave(M // nb, 0) * torch.nn.functional.gelu(pre)
        kw = dict(alpha=0.5, bias=bias, act=ops.ACT_GELU, aux=aux, gate=gate, residual=res, rows_per_batch=M // nb)
    C = ops.gemm(a_in, b_in, a_kmajor=akm, b_kmajor=bkm, out_dtype=out_dtype, **kw)
    tol = (2e-2 if out_dtype == torch.bfloat16 else 3e-3) if dtype == torch.bfloat16 else 1e-4
    tag = f"gemm M{M} N{N} K{K} akm{int(akm)} bkm{int(bkm)} {str(dtype)[6:]}->{str(out_dtype)[6:]}{' epi' if epi else ''}"
    report(tag, rel(C, ref), tol)
    if epi:
        report(tag + " aux", rel(aux, pre), tol)

for dtype, odt in ((torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)):
    for akm in (False, True):
        for bkm in (False, True):
            gemm_case(256, 128, 64, akm, bkm, dtype, odt, False)
            gemm_case(200, 72, 104, akm, bkm, dtype, odt, False)
    gemm_case(384, 256, 1024, False, False, dtype, odt, True)
    gemm_case(48, 24, 32, False, False, dtype, odt, True)

# gelu grad epilogue
M, N, K = 128, 96, 64
A = torch.randn(M, K, device=dev); Bm = torch.randn(N, K, device=dev); pre = torch.randn(M, N, device=dev)
prer = pre.double().requires_grad_(True)
torch.nn.functional.gelu(prer).backward((A.double() @ Bm.double().t()))
C = ops.gemm(A, Bm, act=ops.ACT_GELU_GRAD, aux=pre)
report("gemm gelu-grad epilogue f32", rel(C, prer.grad), 1e-4)

def ln_case(B, N, Cn, mod, odt):
    x = torch.randn(B * N, Cn, device=dev) * 2 + 0.5
    g = torch.randn(Cn, device=dev); b = torch.randn(Cn, device=dev)
    sc = torch.randn(B, Cn, device=dev) * 0.3 if mod else None
    sh = torch.randn(B, Cn, device=dev) if mod else None
    xr = x.double().requires_grad_(True); gr = g.double().requires_grad_(True); br = b.double().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xr, (Cn,), gr, br, 1e-5)
    if mod:
        scr = sc.double().requires_grad_(True); shr = sh.double().requires_grad_(True)
        y_ref = y_ref.view(B, N, Cn) * (1 + scr[:, None]) + shr[:, None]
        y_ref = y_ref.reshape(B * N, Cn)
    dy = torch.randn(B * N, Cn, device=dev)
    dres = torch.randn(B * N, Cn, device=dev)
    y_ref.backward(dy.to(odt).double())
    y, mean, rstd = ops.layernorm_fwd(x, g, b, sc, sh, rows_per_batch=N, out_dtype=odt)
    tol = 1e-2 if odt == torch.bfloat16 else 1e-5
    tag = f"ln B{B} N{N} C{Cn} mod{int(mod)} {str(odt)[6:]}"
    report(tag + " y", rel(y, y_ref), tol)
    dx, dg, db, dsc, dsh = ops.layernorm_bwd(dy.to(odt), x, g, b, sc, mean, rstd, dres=dres, rows_per_batch=N)
    report(tag + " dx", rel(dx, xr.grad + dres.double()), 1e-4)
    report(tag + " dgamma", rel(dg, gr.grad), 1e-4)
    report(tag + " dbeta", rel(db, br.grad), 1e-4)
    if mod:
        report(tag + " dscale", rel(dsc, scr.grad), 1e-4)
        report(tag + " dshift", rel(dsh, shr.grad), 1e-4)

for odt in (torch.float32, torch.bfloat16):
    ln_case(2, 100, 256, True, odt)
    ln_case(3, 37, 30, True, odt)
    ln_case(2, 512, 384, False, odt)
    ln_case(1, 5, 1024, False, odt)

# branch bwd / colsum / cast
for odt in (torch.float32, torch.bfloat16):
    B, N, Cn = 3, 70, 96
    dy = torch.randn(B * N, Cn, device=dev); z = torch.randn(B * N, Cn, device=dev).to(odt); gate = torch.randn(B, Cn, device=dev)
    dz, dgate, dbias = ops.branch_bwd(dy, z, gate, rows_per_batch=N, out_dtype=odt)
    dz_ref = dy.double() * gate.double().repeat_interleave(N, 0)
    report(f"branch_bwd dz {odt}", rel(dz, dz_ref), 1e-2 if odt == torch.bfloat16 else 1e-6)
    report(f"branch_bwd dgate {odt}", rel(dgate, (dy.double() * z.double()).view(B, N, Cn).sum(1)), 1e-5)
    report(f"branch_bwd dbias {odt}", rel(dbias, dz_ref.sum(0)), 1e-5)
    report(f"colsum {odt}", rel(ops.colsum(z), z.double().sum(0)), 1e-5)
x = torch.randn(1000, device=dev)
report("cast f32->bf16", rel(ops.cast(x, torch.bfloat16), x.to(torch.bfloat16)), 0.0)

def drr_case(B, D, H, W, dtype):
    vol = (torch.rand(B, D, H, W, device=dev) * 2 - 1).to(dtype)
    for axis, exp_mode, tr in ((0, True, False), (2, True, True), (0, False, False), (2, False, False)):
        vr = vol.double().requires_grad_(True)
        f = torch.exp(-0.3 * (vr + 1)) if exp_mode else vr
        n = D if axis == 0 else W
        scale = 1.0 if exp_mode else 1.0 / n
        ref = f.sum(1 if axis == 0 else 3) * scale
        if tr: ref = ref.transpose(1, 2)
        cm = 1e-6 if exp_mode else -math.inf
        if exp_mode: ref = ref.clamp(min=1e-6)
        out = ops.drr_fwd(vol, axis, exp_mode=exp_mode, out_scale=scale, clamp_min=cm, transpose_out=tr)
        tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
        report(f"drr fwd {B}x{D}x{H}x{W} ax{axis} exp{int(exp_mode)} {str(dtype)[6:]}", rel(out, ref), tol)
        do = torch.randn_like(ref).to(dtype)
        ref.backward(do.double())
        dv = ops.drr_bwd(vol, out, do, axis, exp_mode=exp_mode, out_scale=scale, clamp_min=cm, transpose_out=tr)
        report(f"drr bwd {B}x{D}x{H}x{W} ax{axis} exp{int(exp_mode)} {str(dtype)[6:]}", rel(dv, vr.grad), tol)

drr_case(2, 8, 6, 4, torch.float32)
drr_case(2, 64, 64, 64, torch.float32)
drr_case(1, 9, 7, 5, torch.float32)
drr_case(2, 32, 32, 32, torch.bfloat16)

# ---- conv stems / norms / resize / loss vs torch fp64 on the CPU ----
import torch.nn.functional as F
from hvc import functional as HF

def conv_case(dims, B, Cin, Cout, k, stride, pad, S, dtype):
    g = torch.Generator().manual_seed(Cin * 100 + Cout + S)
    sp = (S,) * dims
    x = torch.randn(B, Cin, *sp, generator=g)
    w = torch.randn(Cout, Cin, *((k,) * dims), generator=g) / (Cin * k ** dims) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.to(dtype).double().requires_grad_(True) for t in (x, w, b))
    conv = F.conv3d if dims == 3 else F.conv2d
    y_ref = conv(xr, wr, br, stride=stride, padding=pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.to(dtype).double())
    xd = x.to(dev).to(dtype).requires_grad_(True)
    wd = w.to(dev).to(dtype).float().requires_grad_(True)
    bd = b.to(dev).to(dtype).float().requires_grad_(True)
    perm = (0, *range(2, 2 + dims), 1)
    x_cl = xd.permute(*perm).contiguous()
    if dims == 2:
        x_cl = x_cl.reshape(B, 1, S, S, Cin)
        geom = ops.ConvGeometry(B, Cin, (1, S, S), (1, k, k), stride, (0, pad, pad))
    else:
        geom = ops.ConvGeometry(B, Cin, sp, (k,) * 3, stride, (pad,) * 3)
    y = HF.ConvFn.apply(x_cl, wd, bd, None, geom, dtype, dtype)          # (B, OD, OH, OW, Cout)
    y_nc = y.permute(0, 4, 1, 2, 3) if dims == 3 else y[:, 0].permute(0, 3, 1, 2)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    tag = f"conv{dims}d B{B} {Cin}->{Cout} k{k} s{stride} S{S} {str(dtype)[6:]}"
    report(tag + " y", rel(y_nc.cpu(), y_ref), tol)
    dyd = dy.to(dev).to(dtype)
    dy_cl = dyd.permute(0, 2, 3, 4, 1) if dims == 3 else dyd.permute(0, 2, 3, 1).unsqueeze(1)
    (y.float() * dy_cl.float()).sum().backward()
    report(tag + " dx", rel(xd.grad.cpu(), xr.grad), tol)
    report(tag + " dw", rel(wd.grad.cpu(), wr.grad), tol)
    report(tag + " db", rel(bd.grad.cpu(), br.grad), tol)

for dtype in (torch.float32, torch.bfloat16):
    conv_case(3, 2, 1, 16, 3, 2, 1, 16, dtype)
    conv_case(3, 1, 16, 32, 3, 2, 1, 12, dtype)
    conv_case(3, 2, 32, 64, 3, 1, 1, 8, dtype)
    conv_case(2, 2, 1, 64, 7, 2, 3, 64, dtype)
    conv_case(2, 2, 64, 128, 3, 1, 1, 16, dtype)

def gn_case(B, P, Cn, G, dtype):
    g = torch.Generator().manual_seed(Cn + P)
    x = torch.randn(B, P, Cn, generator=g) * 2 + 0.3
    gam, bet = torch.randn(Cn, generator=g), torch.randn(Cn, generator=g)
    xr = x.to(dtype).double().requires_grad_(True); gr = gam.double().requires_grad_(True); br = bet.double().requires_grad_(True)
    y_ref = F.silu(F.group_norm(xr.transpose(1, 2), G, gr, br, 1e-5)).transpose(1, 2)
    dy = torch.randn(B, P, Cn, generator=g)
    y_ref.backward(dy.to(dtype).double())
    xd = x.to(dev).to(dtype).requires_grad_(True); gd = gam.to(dev).requires_grad_(True); bd = bet.to(dev).requires_grad_(True)
    y = HF.GroupNormSiluFn.apply(xd, gd, bd, G, 1e-5)
    y.backward(dy.to(dev).to(dtype))
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    tag = f"gn+silu B{B} P{P} C{Cn} G{G} {str(dtype)[6:]}"
    report(tag + " y", rel(y.cpu(), y_ref), tol)
    report(tag + " dx", rel(xd.grad.cpu(), xr.grad), tol)
    report(tag + " dgamma", rel(gd.grad.cpu(), gr.grad), tol)
    report(tag + " dbeta", rel(bd.grad.cpu(), br.grad), tol)

for dtype in (torch.float32, torch.bfloat16):
    gn_case(2, 4096, 64, 8, dtype)
    gn_case(3, 100, 16, 8, dtype)
    gn_case(1, 700, 128, 8, dtype)

def bn_case(N, H, Cn, pool, training, dtype):
    g = torch.Generator().manual_seed(Cn + H + int(training))
    x = torch.randn(N, Cn, H, H, generator=g) * 1.5 + 0.2
    gam, bet = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.3
    rm, rv = torch.randn(Cn, generator=g) * 0.1, torch.rand(Cn, generator=g) + 0.5
    xr = x.to(dtype).double().requires_grad_(True); gr = gam.double().requires_grad_(True); br = bet.double().requires_grad_(True)
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    y_ref = F.relu(F.batch_norm(xr, rm_r, rv_r, gr, br, training, 0.1, 1e-5))
    if pool: y_ref = F.max_pool2d(y_ref, *pool)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.to(dtype).double())
    xd = x.to(dev).to(dtype).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    gd = gam.to(dev).requires_grad_(True); bd = bet.to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev).clone(), rv.to(dev).clone()
    y = HF.BnReluPoolFn.apply(xd, gd, bd, rmd, rvd, pool, training, 1e-5, 0.1)
    y.backward(dy.to(dev).to(dtype).permute(0, 2, 3, 1).contiguous())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    tag = f"bn+relu+pool N{N} H{H} C{Cn} pool{pool} train{int(training)} {str(dtype)[6:]}"
    report(tag + " y", rel(y.permute(0, 3, 1, 2).cpu(), y_ref), tol)
    report(tag + " dx", rel(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad), 3 * tol)
    report(tag + " dgamma", rel(gd.grad.cpu(), gr.grad), 3 * tol)
    report(tag + " dbeta", rel(bd.grad.cpu(), br.grad), 3 * tol)
    if training:
        report(tag + " running_mean", rel(rmd.cpu(), rm_r), 1e-3 if dtype == torch.bfloat16 else 1e-5)
        report(tag + " running_var", rel(rvd.cpu(), rv_r), 1e-3 if dtype == torch.bfloat16 else 1e-5)

for dtype in (torch.float32, torch.bfloat16):
    for training in (True, False):
        bn_case(4, 32, 64, (3, 2, 1), training, dtype)
        bn_case(2, 16, 128, (2, 2, 0), training, dtype)
        bn_case(2, 8, 32, None, training, dtype)

for (B, ins, outs) in ((2, (4, 4, 4), (16, 16, 16)), (1, (16, 8, 8), (64, 32, 32)), (1, (5, 3, 2), (7, 9, 4)), (1, (8, 8, 8), (8, 8, 8))):
    g = torch.Generator().manual_seed(sum(ins))
    x = torch.randn(B, 1, *ins, generator=g)
    xr = x.double().requires_grad_(True)
    y_ref = F.interpolate(xr, size=outs, mode="trilinear", align_corners=True)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    xd = x.to(dev).requires_grad_(True)
    y = HF.TrilinearFn.apply(xd, outs)
    y.backward(dy.to(dev))
    report(f"trilinear {ins}->{outs} y", rel(y.cpu(), y_ref), 1e-5)
    report(f"trilinear {ins}->{outs} dx", rel(xd.grad.cpu(), xr.grad), 1e-5)

sys.path.insert(0, ROOT)
from oracle import hvc_oracle as O
for shape in ((2, 1, 16, 16, 16), (1, 1, 24, 20, 28), (1, 1, 11, 12, 13)):
    g = torch.Generator().manual_seed(sum(shape))
    p = torch.rand(*shape, generator=g) * 2 - 1; t = torch.rand(*shape, generator=g) * 2 - 1
    pr = p.double().requires_grad_(True)
    ref = O.direct_regression_loss(pr, t.double())
    (ref["total_loss"] + 0.3 * ref["l1_loss"] - 0.2 * ref["ssim_loss"]).backward()
    pd = p.to(dev).requires_grad_(True)
    out = HF.SsimL1LossFn.apply(pd, t.to(dev), 1.0, 0.5, 11)
    (out[0] + 0.3 * out[1] - 0.2 * out[2]).backward()
    report(f"ssim+l1 {shape} total", abs(out[0].item() - ref["total_loss"].item()) / abs(ref["total_loss"].item()), 1e-5)
    report(f"ssim+l1 {shape} ssim", abs(out[2].item() - ref["ssim_loss"].item()) / abs(ref["ssim_loss"].item()), 1e-5)
    report(f"ssim+l1 {shape} dpred", rel(pd.grad.cpu(), pr.grad), 1e-4)

# ---- timing ----
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

if os.environ.get("HVC_TIME", "1") == "1":
    for (B, H, N, M, D) in ((4, 4, 4096, 4096, 64), (2, 4, 32768, 32768, 64), (2, 8, 32768, 1024, 32), (2, 8, 32768, 32768, 32)):
        q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
        o, lse = ops.attention_fwd(q, k, v, D ** -0.5)
        do = torch.randn_like(o)
        tf = timeit(lambda: ops.attention_fwd(q, k, v, D ** -0.5), 5)
        tb = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5), 3)
        fl = 4.0 * B * H * N * M * D
        print(f"attn bf16 B{B} H{H} N{N} M{M} D{D}: fwd {tf*1e3:.3f} ms {fl/tf/1e12:.1f} TF/s | bwd {tb*1e3:.3f} ms {2.5*fl/tb/1e12:.1f} TF/s (algorithmic)", flush=True)
    for (M, N, K) in ((16384, 768, 256), (16384, 256, 1024), (16384, 1024, 256), (65536, 1024, 256)):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, b), 10)
        print(f"gemm bf16 M{M} N{N} K{K}: {t*1e6:.1f} us {2.0*M*N*K/t/1e12:.1f} TF/s", flush=True)
        dyb = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(dyb, a, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32), 10)
        print(f"gemm bf16 dW (TN) M{M} N{N} K{K}: {t*1e6:.1f} us {2.0*M*N*K/t/1e12:.1f} TF/s", flush=True)
    x = torch.randn(131072, 256, device=dev); g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, out_dtype=torch.bfloat16), 10)
    print(f"layernorm fwd 131072x256: {t*1e6:.1f} us {(x.numel()*6)/t/1e9:.0f} GB/s", flush=True)
    vol = torch.rand(2, 256, 256, 256, device=dev)
    for ax in (0, 2):
        t = timeit(lambda: ops.drr_fwd(vol, ax, exp_mode=True, clamp_min=1e-6), 10)
        print(f"drr fwd axis{ax} 2x256^3 f32: {t*1e6:.1f} us {vol.numel()*4/t/1e9:.0f} GB/s", flush=True)

if os.environ.get("HVC_TIME", "1") == "1":
    p = torch.rand(2, 1, 128, 128, 128, device=dev); t = torch.rand_like(p)
    tt = timeit(lambda: HF.SsimL1LossFn.apply(p, t, 1.0, 0.5, 11), 5)
    print(f"ssim+l1 fwd 2x128^3: {tt*1e3:.3f} ms", flush=True)
    for (B, Cin, Cout, S, st) in ((2, 64, 128, 64, 2), (2, 128, 256, 32, 1)):
        x = torch.randn(B, S, S, S, Cin, device=dev, dtype=torch.bfloat16)
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev)
        geom = ops.ConvGeometry(B, Cin, (S, S, S), (3, 3, 3), st, (1, 1, 1))
        tt = timeit(lambda: HF.ConvFn.apply(x, w, None, None, geom, torch.bfloat16, torch.bfloat16), 5)
        fl = 2.0 * geom.M * Cout * Cin * 27
        print(f"conv3d fwd bf16 B{B} {Cin}->{Cout} S{S} s{st}: {tt*1e3:.3f} ms {fl/tt/1e12:.1f} TF/s", flush=True)

print("FAILED:" if fails else "ALL OK", fails)
sys.exit(1 if fails else 0)
