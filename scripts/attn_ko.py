"""Dev aid: timing-only knock-out builds of the dK/dV kernel (build_ab/lib_KO_*.so, wrong results by design) against the
base build on one box: what each ingredient costs at the margin.  usage: attn_ko.py LIB [LIB ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "hybrid-vit-cascade_amd"))
from hvc import _lib
_lib.LIB_PATH = sys.argv[1]
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
out = []
for (B, H, N, M, D) in ((2, 4, 32768, 32768, 64), (1, 8, 32768, 32768, 32)):
    q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    for p in (0.0, 0.1):
        o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7); do = torch.randn_like(o)
        for _ in range(2): ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7)
        ops.PROFILE = []
        for _ in range(5): ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7)
        torch.cuda.synchronize()
        t = min(s.elapsed_time(e) for n, w, s, e in ops.PROFILE if n == "attn_bwd_dkv_kernel")
        ops.PROFILE = None
        out.append(f"D{D} p={p}: {t:.3f}")
print("  ".join(out))
''' % ROOT
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
    print(f"{os.path.basename(lib):22s} dK/dV ms  {r.stdout.strip()}", flush=True)
    if r.returncode: print(r.stderr[-1500:]); sys.exit(1)
