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

print("FAILED:" if fails else "ALL OK", fails)
sys.exit(1 if fails else 0)
