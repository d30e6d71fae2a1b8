"""Per-layer HIP-event timings of the stage-3 cascade glue at 256^3 (model_progressive.py:169-174, 259-271): every Conv3d / GroupNorm+GELU /
trilinear layer of `upsample_from_128` and `detail_enhancer`, forward and backward apart.  usage: glue_layers.py [size=256] [reps=3] [up|detail|all]"""
import os, sys, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import stem as HS

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = sys.argv[3] if len(sys.argv) > 3 else "all"
torch.manual_seed(0)
cdt = torch.bfloat16


def timed(fn):
    best = 1e9
    out = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, out


def layer_times(name, seq, x):
    h = x
    total_f = total_b = 0.0
    layers = list(seq)
    i = 0
    while i < len(layers):
        n = 2 if isinstance(layers[i], nn.GroupNorm) else 1
        sub = nn.Sequential(*layers[i:i + n])
        hin = h.detach().requires_grad_(True)
        tf, y = timed(lambda: HS.glue_sequential(sub, hin, cdt))
        dy = torch.randn_like(y)
        tb, _ = timed(lambda: torch.autograd.grad(HS.glue_sequential(sub, hin, cdt), [hin] + list(sub.parameters()), dy, allow_unused=True))
        tb -= tf
        tb_w = None
        if list(sub.parameters()):
            hin2 = h.detach()
            tb_w, _ = timed(lambda: torch.autograd.grad(HS.glue_sequential(sub, hin2, cdt), list(sub.parameters()), dy))
            tb_w -= tf
        print(f"{name:10s} {type(layers[i]).__name__:10s} in {tuple(h.shape)} -> {tuple(y.shape)}  fwd {tf:8.3f} ms   bwd (dx+dw) {tb:8.3f} ms" +
              (f"   bwd (dw only) {tb_w:8.3f} ms" if tb_w is not None else ""), flush=True)
        total_f += tf
        total_b += tb
        h = y.detach()
        del y, dy
        i += n
    print(f"{name:10s} total fwd {total_f:.3f} ms  bwd {total_b:.3f} ms", flush=True)


up = nn.Sequential(nn.Upsample(scale_factor=2, mode="trilinear", align_corners=False), nn.Conv3d(1, 32, 3, padding=1), nn.GroupNorm(8, 32), nn.GELU()).to(dev)
det = nn.Sequential(nn.Conv3d(1, 64, 3, padding=1), nn.GroupNorm(16, 64), nn.GELU(), nn.Conv3d(64, 32, 3, padding=1), nn.GroupNorm(8, 32), nn.GELU(),
                    nn.Conv3d(32, 1, 1)).to(dev)
if which in ("all", "up"):
    layer_times("upsample", up, torch.randn(1, S // 2, S // 2, S // 2, 1, device=dev))
if which in ("all", "detail"):
    layer_times("detail", det, torch.randn(1, S, S, S, 1, device=dev))
