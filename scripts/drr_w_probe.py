"""Dev aid: drr_fwd along W with and without the transposed output (is the scattered 4-byte store the bound?)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
vols = [torch.rand(4, 256, 256, 256, device=dev) * 2 - 1 for _ in range(3)]
for name, kw in (("W exp transposed", dict(exp_mode=True, clamp_min=1e-6, transpose_out=True)), ("W exp plain", dict(exp_mode=True, clamp_min=1e-6)),
                 ("W mean plain", dict(exp_mode=False, out_scale=1 / 256)), ("D exp", None)):
    ts = []
    for i in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if kw is None: ops.drr_fwd(vols[i % 3], 0, exp_mode=True, clamp_min=1e-6)
        else: ops.drr_fwd(vols[i % 3], 2, **kw)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name:20s} {ms * 1e3:8.1f} us  {vols[0].numel() * 4 / ms / 1e6:7.0f} GB/s")
