"""Dev aid: per-shape GEMM time of one train step of a bench workload, against a bytes/flops floor."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
import bench
from hvc import ops

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "direct128"]
dev = torch.device("cuda:0")
model, crit, opt = bench.build(wl, dev)
xr, ct = bench.make_batch(wl, 0, dev)
params = [p for p in model.parameters() if p.requires_grad]
rec = []
orig = ops.gemm
def gemm(a, b, **kw):
    M, K = (a.shape[1], a.shape[0]) if kw.get("a_kmajor") else a.shape
    N = b.shape[1] if kw.get("b_kmajor") else b.shape[0]
    od = kw.get("out_dtype") or a.dtype
    esz = lambda t: 2 if t == torch.bfloat16 else 4
    bytes_ = M * K * esz(a.dtype) + N * K * esz(a.dtype) + M * N * esz(od)
    for name in ("aux", "zsave", "residual"):
        t = kw.get(name)
        if t is not None:
            bytes_ += t.numel() * t.element_size() * (1 if name != "aux" or kw.get("act") != ops.ACT_GELU else 1)
    flags = "".join(c for c, n in (("b", "bias"), ("g", "gate"), ("r", "residual"), ("z", "zsave"), ("x", "aux")) if kw.get(n) is not None)
    flags += f"a{kw.get('act', 0)}" + ("d" if kw.get("p_drop", 0) else "")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = orig(a, b, **kw); e.record()
    rec.append(((M, N, K, int(bool(kw.get("a_kmajor"))), int(bool(kw.get("b_kmajor"))), str(a.dtype)[6:], str(od)[6:], flags), bytes_, s, e))
    return out
for it in range(3):
    if it == 2:
        ops.gemm = gemm
        import hvc.functional as HF, hvc.stem as HS
    bench.train_step(model, params, crit, opt, xr, ct)
torch.cuda.synchronize()
ops.gemm = orig
agg = collections.OrderedDict()
for key, by, s, e in rec:
    d = agg.setdefault(key, [0, 0.0, by])
    d[0] += 1; d[1] += s.elapsed_time(e) * 1e3
tot = sum(d[1] for d in agg.values())
print(f"{'M':>7} {'N':>5} {'K':>6} akm bkm in   out  epi      n   avg_us  TF/s  floor_us(5TB/s|2.5PF)  x_floor   total_us")
for (M, N, K, akm, bkm, idt, odt, fl), (n, us, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    floor = max(by / 5e12, 2.0 * M * N * K / 2.5e15) * 1e6
    print(f"{M:7d} {N:5d} {K:6d} {akm:3d} {bkm:3d} {idt:4s} {odt:4s} {fl:8s} {n:3d} {us/n:8.1f} {2.0*M*N*K/(us/n)/1e6:6.0f} {floor:10.1f} {us/n/floor:16.2f} {us:10.0f}")
print(f"total GEMM time {tot/1e3:.2f} ms/step; sum of floors {sum(max(d[2]/5e12, 2.0*k[0]*k[1]*k[2]/2.5e15)*1e6*d[0] for k, d in agg.items())/1e3:.2f} ms")
