"""Dev aid: the cascade's 256^3 detail-enhancer convolution (64 -> 32 channels, k3 s1 p1; model_progressive.py:122) through the
implicit-GEMM path: forward, input gradient, weight gradient - time per call and effective TFLOP/s (for rocprofv3 --pmc too)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops, functional as HF
dev = torch.device("cuda:0"); torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cin, cout = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 32)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(1, S, S, S, cin, device=dev, dtype=torch.bfloat16)
w = (torch.randn(cout, cin, 3, 3, 3, device=dev) / (cin * 27) ** 0.5).requires_grad_(True)
b = torch.zeros(cout, device=dev)
geom = ops.ConvGeometry(1, cin, (S, S, S), (3, 3, 3), 1, (1, 1, 1))
w2d = HF.conv_weight_2d(w, torch.bfloat16, geom.Kp)
wt = HF.conv_weight_2d_t(w, torch.bfloat16)
dy = torch.randn(geom.M, cout, device=dev, dtype=torch.bfloat16)
gd = ops.ConvGeometry(1, cout, geom.out, geom.kernel, 1, (1, 1, 1))
flops = 2.0 * geom.M * cout * 27 * cin
def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for name, fn in (("forward  y = patches(x) W^T", lambda: ops.conv_gemm(x, w2d, geom, bias=b)),
                 ("dx       = patches(dy) W'^T (mirrored taps)", lambda: ops.conv_gemm(dy.view(1, S, S, S, cout), wt, gd, flip=True)),
                 ("dW       = dy^T patches(x) (split-K)", lambda: ops.conv_gemm_dw(x, dy, geom))):
    ms = timeit(fn)
    print(f"{name:48s} {S}^3 {cin}->{cout}: {ms:7.2f} ms  {flops / ms / 1e9:7.0f} TFLOP/s useful", flush=True)
