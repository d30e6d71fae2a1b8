"""Dev aid: same-box A/B of two builds of libhvc_hip.so on the self-attention benchmark shape (alternating runs)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.join(%r, "hybrid-vit-cascade_amd"))
from hvc import _lib
_lib.LIB_PATH = sys.argv[1]
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, H, N, M, D = 2, 4, 32768, 32768, 64
q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
for p in (0.0, 0.1):
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7); do = torch.randn_like(o)
    fwd_only = os.environ.get("ATTN_AB_FWD_ONLY") == "1"
    for _ in range(2): fwd_only or ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7)
    ops.PROFILE = []
    for _ in range(6):
        with ops._Timed("attn_fwd_kernel", 0): ops.attention_fwd(q, k, v, D ** -0.5, p, 7)
        fwd_only or ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7)
    torch.cuda.synchronize()
    acc = {}
    for name, w, s, e in ops.PROFILE: acc.setdefault(name, []).append(s.elapsed_time(e))
    ops.PROFILE = None
    print(f"p={p}: " + "  ".join(f"{n.replace('attn_', '').replace('_kernel', '')} {min(x):.3f}" for n, x in sorted(acc.items()) if "delta" not in n), flush=True)
''' % ROOT
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2
libs = [a for a in sys.argv[1:] if not a.isdigit()]
for rnd in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print(os.path.basename(lib), "|", " | ".join(l for l in out.stdout.strip().splitlines()), flush=True)
        if out.returncode: print(out.stderr[-2000:])
