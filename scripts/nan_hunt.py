"""Dev aid: run train steps of a bench workload and report the first non-finite loss / gradient / parameter."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "direct128"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
model, crit, opt = bench.build(wl, dev)
xr, ct = bench.make_batch(wl, 0, dev)
params = [p for p in model.parameters() if p.requires_grad]
names = {id(p): n for n, p in model.named_parameters()}
torch.autograd.set_detect_anomaly(True, check_nan=True)
for s in range(steps):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(xr)
        loss = crit(pred.float(), ct)["total_loss"]
    loss.backward()
    bad_g = [names[id(p)] for p in params if p.grad is not None and not torch.isfinite(p.grad).all()]
    gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    bad_p = [names[id(p)] for p in params if not torch.isfinite(p).all()]
    print(f"step {s}: loss {loss.item():.6f} pred finite {bool(torch.isfinite(pred).all())} gnorm {gn.item():.4f} "
          f"bad grads {bad_g[:6]} ({len(bad_g)}) bad params {bad_p[:4]} ({len(bad_p)})", flush=True)
    if bad_g or bad_p or not torch.isfinite(loss):
        break
