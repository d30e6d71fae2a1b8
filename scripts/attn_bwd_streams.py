"""Experiment: the dK/dV and dQ kernels of one attention backward on two HIP streams (they only share inputs) against the
usual back-to-back launches on one stream.  usage: python scripts/attn_bwd_streams.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (B, H, N, M, D) in ((2, 4, 32768, 32768, 64), (1, 8, 32768, 32768, 32)):
    q = torch.randn(B, N, H, D, device=dev, dtype=torch.bfloat16); k = torch.randn(B, M, H, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    p = 0.1
    o, lse = ops.attention_fwd(q, k, v, D ** -0.5, p, 7); do = torch.randn_like(o)
    dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
    ws = torch.empty((lib.hvc_attention_bwd_workspace(B, H, N, M, D),), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()

    def call(phases, stream):
        _lib.check(lib.hvc_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), ws.data_ptr(),
                                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, N, M, D, *ops._bnhd_strides(q), *ops._bnhd_strides(k),
                                         *ops._bnhd_strides(v), *ops._bnhd_strides(o), float(D ** -0.5), p, 7, phases, 1, stream.cuda_stream), "bwd")

    def serial():
        call(7, torch.cuda.current_stream())

    def overlapped():
        cur = torch.cuda.current_stream()
        call(1, cur)                       # delta
        side.wait_stream(cur)
        call(2, cur)                       # dK/dV
        call(4, side)                      # dQ beside it
        cur.wait_stream(side)

    for name, fn in (("serial", serial), ("two streams", overlapped), ("serial", serial), ("two streams", overlapped)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        t = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
        print(f"D={D} {name:12s} backward {min(t):.3f} ms", flush=True)
