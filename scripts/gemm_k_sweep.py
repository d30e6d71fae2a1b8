"""Dev aid: GEMM time vs K at M=65536 (separates per-tile fixed cost from the k-loop)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
M = 65536
for N in (256, 1024):
    for K in (64, 128, 256, 512, 1024, 2048):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        bk = b.t().contiguous()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_nn = timeit(lambda: ops.gemm(a, b, out=out))
        t_nt = timeit(lambda: ops.gemm(a, bk, b_kmajor=True, out=out))
        by = (M * K + N * K + M * N) * 2
        print(f"M{M} N{N} K{K}: NN {t_nn:7.1f} us  NT {t_nt:7.1f} us | {2.0*M*N*K/t_nn/1e6:6.0f} TF/s | floor {max(by/5e12, 2.0*M*N*K/2.5e15)*1e6:6.1f} us", flush=True)
