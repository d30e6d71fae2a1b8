"""Dev aid: fp8 (e4m3) attention forward vs the bf16 forward, alternating launches in one process, at the cascade stage-3 shape
(B1 H8 N32768 d32, self- and cross-attention) and at the direct 128^3 shape (B2 H4 N32768 d64)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, Nq, Nk, D, label) in ((1, 8, 32768, 32768, 32, "stage-3 self "), (1, 8, 32768, 4096, 32, "stage-3 cross"), (2, 4, 32768, 32768, 64, "direct128 self")):
    q = torch.randn(B, Nq, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, Nk, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    for p in (0.1, 0.0):
        times = {False: [], True: []}
        for fp8 in (False, True):
            ops.attention_fwd(q, k, v, D ** -0.5, p, 3, fp8=fp8)
        for _ in range(6):
            for fp8 in (False, True):
                ops.PROFILE, ops.PROFILE_ONLY = [], None
                ops.attention_fwd(q, k, v, D ** -0.5, p, 3, fp8=fp8)
                torch.cuda.synchronize()
                times[fp8].append(sum(s.elapsed_time(e) for _, _, s, e in ops.PROFILE))
                ops.PROFILE = None
        fl = 4.0 * B * H * Nq * Nk * D
        b16, f8 = sorted(times[False])[3], sorted(times[True])[3]
        print(f"{label} B{B} H{H} Nq{Nq} Nk{Nk} d{D} p={p}: bf16 {b16:.3f} ms ({fl / b16 / 1e9:.0f} TF/s) | fp8 incl. quantisation pre-pass {f8:.3f} ms ({fl / f8 / 1e9:.0f} TF/s)", flush=True)
