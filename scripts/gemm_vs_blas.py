"""Dev aid: plain GEMM shapes of the block, hvc_gemm vs torch.mm (hipBLASLt / rocBLAS)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for (N, K) in ((256, 256), (768, 256), (1024, 256), (256, 1024)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: ops.gemm(x, w)); t2 = timeit(lambda: torch.mm(x, w.t()))
    print(f"fwd  y = x W^T   M{M} N{N} K{K}: hvc {t1:6.1f} us  torch {t2:6.1f} us")
    t1 = timeit(lambda: ops.gemm(dy, w, b_kmajor=True)); t2 = timeit(lambda: torch.mm(dy, w))
    print(f"dx = dy W        M{M} N{K} K{N}: hvc {t1:6.1f} us  torch {t2:6.1f} us")
    t1 = timeit(lambda: ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)); t2 = timeit(lambda: torch.mm(dy.t(), x))
    print(f"dW = dy^T x      M{N} N{K} K{M}: hvc {t1:6.1f} us (fp32 out, deterministic split-K)  torch {t2:6.1f} us (bf16 out)", flush=True)
