"""Dev aid: a few launches of two token-matrix GEMM shapes through hvc_gemm and torch.mm, for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
M = 65536
for (N, K) in ((768, 256), (256, 1024)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(x, w)
    for _ in range(3): torch.mm(x, w.t())
    torch.cuda.synchronize()
