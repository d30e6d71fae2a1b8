"""Dev aid / soak: 24 optimisation steps of the direct 128^3 model (the bench workload: 8-wavefront attention backward, persistent GEMMs)
on one synthetic batch (dropout 0.1, bf16): the loss must fall and stay finite."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
import bench
from direct_regression import train_direct_4gpu as T
dev = torch.device("cuda:0"); torch.manual_seed(0)
wl = bench.WORKLOADS["direct128"]
model, crit, opt = bench.build(wl, dev)
xr, ct = bench.make_batch(wl, 0, dev)
losses = []
t0 = time.time()
for it in range(24):
    out = T.train_step(model, crit, opt, None, xr, ct, 1.0)
    losses.append(float(out["total_loss"].detach()))
torch.cuda.synchronize()
print("24 steps on one batch of 2 (128^3, dropout 0.1, bf16): loss", " ".join(f"{l:.4f}" for l in losses[::3]), f"| {time.time()-t0:.1f} s")
assert all(l == l and l < 1e4 for l in losses) and losses[-1] < 0.6 * losses[0], losses
print("ok")
