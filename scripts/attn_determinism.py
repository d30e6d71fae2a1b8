"""Dev aid: run the attention forward / backward repeatedly on fixed inputs and report outputs that are not bit-identical."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (B, H, N, M, D) in ((2, 2, 4096, 4096, 32), (2, 2, 4096, 64, 32), (2, 4, 4096, 4096, 64), (2, 4, 4096, 4096 - 37, 64), (1, 8, 32768, 1024, 32)):
    for dt in (torch.bfloat16, torch.float32):
        q = torch.randn(B, N, H, D, device=dev, dtype=dt); k = torch.randn(B, M, H, D, device=dev, dtype=dt); v = torch.randn_like(k)
        for p in (0.0, 0.1):
            ref = None
            bad = set()
            for it in range(6):
                o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7)
                do = torch.ones_like(o) * 0.5
                dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7)
                cur = dict(o=o, lse=lse, dq=dq, dk=dk, dv=dv)
                if ref is None:
                    ref = {n: t.clone() for n, t in cur.items()}
                else:
                    for n, t in cur.items():
                        if not torch.equal(t, ref[n]):
                            bad.add(f"{n}({(t.float() - ref[n].float()).abs().max().item():.2e}, {(t != ref[n]).sum().item()} elts)")
            print(f"B{B} H{H} N{N} M{M} D{D} {str(dt)[6:]} p={p}: {'deterministic' if not bad else 'DIFFERS: ' + ' '.join(sorted(bad))}", flush=True)
