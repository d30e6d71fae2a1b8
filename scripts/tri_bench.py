"""Dev aid: trilinear resize kernel timings at the model shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for (B, i, o, ac) in ((2, 32, 128, True), (4, 16, 64, True), (1, 128, 256, False), (1, 32, 256, True)):
    x = torch.randn(B, i, i, i, device=dev); dy = torch.randn(B, o, o, o, device=dev)
    tf = timeit(lambda: ops.trilinear_fwd(x, (o, o, o), ac))
    ts = timeit(lambda: ops.trilinear_bwd(dy, (i, i, i), ac, separable=True))
    tg = timeit(lambda: ops.trilinear_bwd(dy, (i, i, i), ac, separable=False))
    print(f"trilinear B{B} {i}^3 -> {o}^3 ac={ac}: fwd {tf:.1f} us | bwd separable {ts:.1f} us | bwd gather {tg:.1f} us", flush=True)
