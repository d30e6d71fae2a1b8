"""The HBM-bound kernels north_star names (DRR ray sums: diagnostic_losses.py:31-65, loss_multiscale.py:250-273; LayerNorm:
hybrid_vit_backbone.py:84-86; trilinear resize :272; SSIM + L1: model_direct.py:88-131) on working sets far above the 256 MiB
Infinity Cache, every launch on buffers of its own (a ring of distinct tensors, so no launch re-reads what the previous one left in
cache).  Run it under rocprofv3 (--kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes) and
summarise with scripts/hbm_summary.py; run alone it prints HIP-event timings.

Prints one line per case: name, kernel-name needle, launches, algorithmic bytes per launch (the bytes the operation must move:
every input read once, every output written once)."""
import json
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops  # noqa: E402

dev = torch.device("cuda:0")
REPS = 6
RING = 3             # distinct operand sets per case (each >= 268 MB): the ring alone exceeds the 256 MiB Infinity Cache 3x over


def ring(make):
    return [make() for _ in range(RING)]


cases = []


_sentinel_src = torch.zeros(4096, device=dev)


def sentinel():
    """One hvc cast launch = a boundary mark in the kernel trace (the only cast launches of this script): scripts/hbm_summary.py
    cuts the trace at these marks - three per case: [warm-up] [timed launches] [whatever follows]."""
    ops.cast(_sentinel_src, torch.bfloat16)


def case(name, needle, bytes_per_launch, fn, sets):
    sentinel()
    for i in range(2):
        fn(sets[i % RING])
    torch.cuda.synchronize()
    sentinel()
    evs = []
    for i in range(REPS):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(sets[i % RING])
        b.record()
        evs.append((a, b))
    sentinel()               # end of the timed window: whatever prepares the next case stays outside
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    cases.append(dict(name=name, kernel=needle, launches=REPS, bytes=bytes_per_launch, event_ms=ms,
                      event_GBps=bytes_per_launch / ms / 1e6))
    print(f"{name:44s} {bytes_per_launch / 1e6:9.1f} MB  {ms * 1e3:9.1f} us  {bytes_per_launch / ms / 1e6:8.0f} GB/s (HIP events, incl. launch gap)", flush=True)


# ---- DRR ray sums: 4 x 256^3 fp32 = 268 MB per volume batch
B, D = 4, 256
vols = ring(lambda: torch.rand(B, D, D, D, device=dev) * 2 - 1)
vbytes = vols[0].numel() * 4
pbytes = B * D * D * 4
case("drr_fwd axis D (exp, ProjectionLoss AP)", "drr_fwd_d", vbytes + pbytes, lambda v: ops.drr_fwd(v, 0, exp_mode=True, clamp_min=1e-6), vols)
case("drr_fwd axis W (exp, lateral)", "drr_fwd_w", vbytes + pbytes, lambda v: ops.drr_fwd(v, 2, exp_mode=True, clamp_min=1e-6, transpose_out=True), vols)
case("drr_fwd axis D (mean, DRRReprojectionLoss)", "drr_fwd_d", vbytes + pbytes, lambda v: ops.drr_fwd(v, 0, exp_mode=False, out_scale=1.0 / D), vols)
outs = [ops.drr_fwd(v, 0, exp_mode=True, clamp_min=1e-6) for v in vols]
douts = [torch.rand_like(o) for o in outs]
sets = list(zip(vols, outs, douts))
case("drr_bwd axis D (exp)", "drr_bwd", 2 * vbytes + 2 * pbytes, lambda s: ops.drr_bwd(s[0], s[1], s[2], 0, exp_mode=True, clamp_min=1e-6), sets)
del outs, douts, sets

# ---- SSIM + L1 on the same volumes (pred, target -> 3 gradient maps; backward: pred, target, 3 maps -> dpred)
tg = ring(lambda: torch.rand(B, D, D, D, device=dev) * 2 - 1)
sets = list(zip(vols, tg))
case("ssim_l1_fwd 4 x 256^3 (all passes)", "ssim", 2 * vbytes + 3 * vbytes, lambda s: ops.ssim_l1_fwd(s[0], s[1]), sets)
gm = [ops.ssim_l1_fwd(v, t)[1] for v, t in sets]
gs = torch.ones(1, device=dev)
sets2 = [(v, t, g) for (v, t), g in zip(sets, gm)]
case("ssim_l1_bwd 4 x 256^3 (all passes)", "ssim", 5 * vbytes + vbytes, lambda s: ops.ssim_l1_bwd(s[0], s[1], s[2], gs), sets2)
del gm, sets2, tg, sets

# ---- trilinear resize 32^3 -> 256^3 (head upsample, write-bound) and its adjoint (read-bound)
small = ring(lambda: torch.rand(B * 4, 32, 32, 32, device=dev))
big_bytes = B * 4 * D ** 3 * 4
case("trilinear_fwd 16 x 32^3 -> 256^3", "trilinear_fwd", big_bytes + small[0].numel() * 4, lambda x: ops.trilinear_fwd(x, (D, D, D)), small)
bigs = ring(lambda: torch.rand(B * 4, D, D, D, device=dev))
case("trilinear_bwd 16 x 256^3 -> 32^3", "trilinear", big_bytes + small[0].numel() * 4, lambda x: ops.trilinear_bwd(x, (32, 32, 32)), bigs)
del bigs, small, vols

# ---- LayerNorm over the residual stream: 1 M rows x 256 fp32 = 1.07 GB in, bf16 out
rows, C = 1 << 20, 256
xs = ring(lambda: torch.randn(rows, C, device=dev))
gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
sc, sh = torch.randn(rows // 32768, C, device=dev), torch.randn(rows // 32768, C, device=dev)
case("ln_fwd 1M x 256 fp32 -> bf16 (+AdaLN)", "ln_fwd", rows * C * (4 + 2) + rows * 8,
     lambda x: ops.layernorm_fwd(x, gamma, beta, sc, sh, rows_per_batch=32768, out_dtype=torch.bfloat16), xs)
st = [ops.layernorm_fwd(x, gamma, beta, sc, sh, rows_per_batch=32768, out_dtype=torch.bfloat16) for x in xs]
dys = ring(lambda: torch.randn(rows, C, device=dev).to(torch.bfloat16))
dres = ring(lambda: torch.randn(rows, C, device=dev))
sets = [(dys[i], xs[i], st[i][1], st[i][2], dres[i]) for i in range(RING)]
case("ln_bwd 1M x 256 (dy bf16, x, dres -> dx)", "ln_bwd", rows * C * (2 + 4 + 4 + 4) + rows * 8,
     lambda s: ops.layernorm_bwd(s[0], s[1], gamma, beta, sc, s[2], s[3], dres=s[4], rows_per_batch=32768), sets)

out = os.environ.get("HVC_HBM_CASES")
if out:
    json.dump(cases, open(out, "w"), indent=1)
