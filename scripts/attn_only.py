"""Runs only the attention kernels at the 128^3 / 64^3 benchmark shapes (profiling target)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "128"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
B, H, N, M, D = (2, 4, 32768, 32768, 64) if which == "128" else (4, 4, 4096, 4096, 64)
q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
for _ in range(3):
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 1)
do = torch.randn_like(o)
for _ in range(3):
    ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 1)
torch.cuda.synchronize()
print("done")
