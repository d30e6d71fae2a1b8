"""Dev aid (round 4): same-box alternating A/B of the software-pipelined attention kernels (HVC_ATTN_PIPE=1) against the
phase-separated ones (=0) on the shapes of the BASELINE configs, dropout on / off.  HIP-event minimum and median per form.
usage: python scripts/attn_pipe_ab.py [fwd|bwd|all]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dev = torch.device("cuda:0"); torch.manual_seed(0)
shapes = [("direct128 self", 2, 4, 32768, 32768, 64), ("direct128 cross", 2, 4, 32768, 4096, 64),
          ("stage3 self", 1, 8, 32768, 32768, 32), ("stage2 cross", 2, 8, 32768, 1024, 32)]
def timed(fn, n=3):
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return ts
for name, B, H, N, M, D in shapes:
    q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    for p in (0.1, 0.0):
        res = {0: {"fwd": [], "bwd": []}, 1: {"fwd": [], "bwd": []}, 2: {"fwd": [], "bwd": []}}
        outs = {}
        for rnd in range(4):
            for form in (0, 1, 2):                        # 2 = pipelined, 128-row workgroups of four wavefronts (two per CU)
                ops.set_option("HVC_ATTN_PIPE", min(form, 1))
                ops.set_option("HVC_ATTN_FWD_WAVES", 4 if form == 2 else 0)
                if what in ("fwd", "all"):
                    res[form]["fwd"] += timed(lambda: ops.attention_fwd(q, k, v, D ** -0.5, p, 7))
                if what in ("bwd", "all"):
                    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7); do = torch.ones_like(o)
                    res[form]["bwd"] += timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, D ** -0.5, p, 7))
                if rnd == 0:
                    outs[form] = ops.attention_fwd(q, k, v, D ** -0.5, p, 7)
        ops.set_option("HVC_ATTN_PIPE", 1)
        ops.set_option("HVC_ATTN_FWD_WAVES", 0)
        if res[2]["fwd"]:
            print(f"{name:16s} p={p}: fwd pipelined, 4-wavefront workgroups: min {min(res[2]['fwd']):.3f} med {statistics.median(res[2]['fwd']):.3f} ms", flush=True)
        d = (outs[0][0].float() - outs[1][0].float()).norm() / outs[0][0].float().norm()
        flops = 4.0 * B * H * N * M * D
        line = f"{name:16s} p={p}: "
        for key in ("fwd", "bwd"):
            if res[0][key]:
                a, b = res[0][key], res[1][key]
                line += f"{key} phase-separated min {min(a):.3f} med {statistics.median(a):.3f} | pipelined min {min(b):.3f} med {statistics.median(b):.3f} ms ({100 * (min(b) / min(a) - 1):+.1f} %)"
                if key == "fwd":
                    line += f" = {flops / min(b) / 1e9:.0f} TF/s ({flops / min(b) / 1e9 / 2500:.3f} of peak)  "
        print(line + f" | rel diff of O {d.item():.2e}", flush=True)
