"""Times the attention kernels at the benchmark shapes, with and without dropout (per-kernel split of the backward)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def split(fn, n):
    ops.PROFILE = []
    for _ in range(n): fn()
    torch.cuda.synchronize()
    acc = {}
    for name, work, s, e in ops.PROFILE:
        acc.setdefault(name, []).append(s.elapsed_time(e))
    ops.PROFILE = None
    return {k: sum(v) / len(v) for k, v in acc.items()}
shapes = ((2, 4, 32768, 32768, 64), (2, 4, 32768, 4096, 64), (4, 4, 4096, 4096, 64), (2, 8, 32768, 32768, 32))
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:1]
if len(sys.argv) > 1 and sys.argv[1] == "cross":
    shapes = shapes[1:2]
for (B, H, N, M, D) in shapes:
    q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    for p in (0.0, 0.1):
        o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7)
        do = torch.randn_like(o)
        tf = timeit(lambda: ops.attention_fwd(q, k, v, D ** -0.5, p, 7), 5)
        tb = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7), 3)
        sp = split(lambda: ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7), 3)
        fl = 4.0 * B * H * N * M * D
        print(f"attn bf16 B{B} H{H} N{N} M{M} D{D} p={p}: fwd {tf*1e3:.3f} ms {fl/tf/1e12:.0f} TF/s | bwd {tb*1e3:.3f} ms {2.5*fl/tb/1e12:.0f} TF/s (algorithmic)"
              f" | dkv {sp.get('attn_bwd_dkv_kernel', 0):.3f} ms {2*fl/sp.get('attn_bwd_dkv_kernel', 1)/1e9:.0f} TF/s, dq {sp.get('attn_bwd_dq_kernel', 0):.3f} ms", flush=True)
