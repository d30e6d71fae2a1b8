"""Dev aid: HIP-event timings of the attention kernels of THIS tree's library on the shapes of the BASELINE configs
(direct 128^3: 4 x 64 heads; cascade stage 2 / 3: 8 x 32 heads, self- and cross-attention), dropout off / on.
usage: python scripts/attn_shapes.py [ROOT]   (ROOT = repo root or a worktree of it; default: this file's repo)"""
import os, sys, torch
ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
has_bits = hasattr(ops, "attention_dropmask")
shapes = [("direct128 self", 2, 4, 32768, 32768, 64), ("direct128 cross", 2, 4, 32768, 4096, 64),
          ("stage3 self", 1, 8, 32768, 32768, 32), ("stage3 cross", 1, 8, 32768, 4096, 32), ("stage2 cross", 2, 8, 32768, 1024, 32)]
for name, B, H, N, M, D in shapes:
    q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    for p in (0.0, 0.1):
        kw = {}
        if has_bits and p > 0:
            kw["bits"] = ops.attention_dropmask(B, H, N, M, p, 7, dev)
        o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7, **kw); do = torch.randn_like(o)
        for _ in range(2): ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7, **kw)
        ops.PROFILE = []
        for _ in range(5):
            if has_bits and p > 0: ops.attention_dropmask(B, H, N, M, p, 7, dev)
            ops.attention_fwd(q, k, v, D ** -0.5, p, 7, **kw)
            ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7, **kw)
        torch.cuda.synchronize()
        acc = {}
        for nm, w, s, e in ops.PROFILE: acc.setdefault(nm, []).append(s.elapsed_time(e))
        ops.PROFILE = None
        tot = sum(min(x) for n_, x in acc.items())
        print(f"{name:16s} p={p}: " + "  ".join(f"{n_.replace('attn_', '').replace('_kernel', '')} {min(x):.3f}" for n_, x in sorted(acc.items()) if "delta" not in n_) + f"  | sum {tot:.3f} ms", flush=True)
