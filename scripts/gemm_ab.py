"""Dev aid: same-box A/B of two library builds on the block's GEMM shapes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "hybrid-vit-cascade_amd"))
from hvc import _lib
_lib.LIB_PATH = sys.argv[1]
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
M = 65536
res = []
for (N, K) in ((256, 256), (768, 256), (1024, 256), (256, 1024)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    res.append(f"{N}x{K}: fwd {timeit(lambda: ops.gemm(x, w)):.1f} dx {timeit(lambda: ops.gemm(dy, w, b_kmajor=True)):.1f} "
               f"dW {timeit(lambda: ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)):.1f}")
print(" | ".join(res))
''' % ROOT
libs = sys.argv[1:3]
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print(os.path.basename(lib), "|", out.stdout.strip(), flush=True)
        if out.returncode: print(out.stderr[-2000:])
