import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hybrid-vit-cascade_amd"))
from hvc import ops
dev = torch.device("cuda:0")
def timed(fn, n=5):
    best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record(); fn(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e))
    return best
x = torch.randn(16, 32, 32, 32, device=dev)
dy = torch.randn(16, 256, 256, 256, device=dev)
for ac in (True, False):
    tb = timed(lambda: ops.trilinear_bwd(dy, (32, 32, 32), ac))
    mb = (x.numel() + dy.numel()) * 4 / 1e6
    print(f"align_corners={ac}: bwd {tb*1e3:.1f} us = {mb/tb/1e3:.2f} TB/s ({mb/tb/1e3/8:.2f} of 8)")
x2 = torch.randn(2, 128, 128, 128, device=dev); dy2 = torch.randn(2, 256, 256, 256, device=dev)
print("128^3 -> 256^3 B=2: bwd %.1f us" % (1e3 * timed(lambda: ops.trilinear_bwd(dy2, (128, 128, 128), False))))
