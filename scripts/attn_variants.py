"""Dev aid: the two dK/dV kernels (HVC_ATTN_DKV=1: 32 keys / wave, two waves per SIMD; =2: 64 keys / wave, one wave per SIMD)
checked against each other and timed in alternating launches in ONE process (guide rule 24).  usage: attn_variants.py [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def run(variant, q, k, v, o, do, lse, scale, p, phases_only_dkv=True):
    os.environ["HVC_ATTN_DKV"] = str(variant)
    return ops.attention_bwd(q, k, v, o, do, lse, scale, p, 7)


def check(B, H, Nq, Nk, D, p):
    g = torch.Generator(device="cpu").manual_seed(Nq * 31 + Nk)
    q = torch.randn(B, Nq, H, D, generator=g).to(dev, torch.bfloat16)
    k = torch.randn(B, Nk, H, D, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(B, Nk, H, D, generator=g).to(dev, torch.bfloat16)
    do = torch.randn(B, Nq, H, D, generator=g).to(dev, torch.bfloat16)
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7)
    r1 = run(1, q, k, v, o, do, lse, D ** -0.5, p)
    r2 = run(2, q, k, v, o, do, lse, D ** -0.5, p)
    errs = [((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item() for a, b in zip(r2, r1)]
    ok = all(e < 1.5e-2 for e in errs) and all(torch.isfinite(t.float()).all().item() for t in r2)
    print(f"check B{B} H{H} Nq{Nq} Nk{Nk} D{D} p{p}: rel L2 (dq, dk, dv) v2 vs v1 = {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e} {'OK' if ok else 'MISMATCH'}", flush=True)
    return ok


def bench(B, H, Nq, Nk, D, p, label):
    q = torch.randn(B, Nq, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, Nk, H, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7)
    do = torch.randn_like(o)
    times = {1: [], 2: []}
    for variant in (1, 2):
        run(variant, q, k, v, o, do, lse, D ** -0.5, p)
    for _ in range(reps):
        for variant in (1, 2):
            os.environ["HVC_ATTN_DKV"] = str(variant)
            ops.PROFILE, ops.PROFILE_ONLY = [], None
            ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7)
            torch.cuda.synchronize()
            t = {n: s.elapsed_time(e) for n, w, s, e in ops.PROFILE}
            ops.PROFILE = None
            times[variant].append(t["attn_bwd_dkv_kernel"])
    flops = 8.0 * B * H * Nq * Nk * D
    out = []
    for variant in (1, 2):
        tt = sorted(times[variant])
        out.append(f"v{variant} min {tt[0]:.3f} med {tt[len(tt) // 2]:.3f} ms = {flops / tt[len(tt) // 2] / 1e9:.0f} TF/s")
    print(f"{label} B{B} H{H} Nq{Nq} Nk{Nk} D{D} p{p}: " + " | ".join(out), flush=True)


ok = True
for shp in [(2, 4, 32768, 32768, 64, 0.1), (1, 2, 1100, 130, 64, 0.1), (2, 3, 65, 257, 64, 0.0), (1, 8, 4096, 1024, 32, 0.1), (1, 2, 129, 300, 32, 0.0),
            (1, 4, 32768, 4096, 64, 0.1)]:
    ok = check(*shp) and ok
if not ok:
    print("MISMATCH: timings skipped")
    sys.exit(1)
for p in (0.1, 0.0):
    bench(2, 4, 32768, 32768, 64, p, "self ")
    bench(2, 4, 32768, 4096, 64, p, "cross")
bench(1, 8, 32768, 32768, 32, 0.1, "self d32")
bench(4, 4, 4096, 4096, 64, 0.1, "self 64^3")
