"""Dev aid / soak: 60 optimisation steps of the direct 64^3 model on one synthetic batch (dropout 0.1, bf16): the loss must fall."""
import os, sys, json, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
from direct_regression import train_direct_4gpu as T
from hvc import synthetic
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = DirectCTRegression(volume_size=(64, 64, 64)).to(dev).train()
crit = DirectRegressionLoss(1.0, 0.5)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, fused=True)
xr, ct = synthetic.batch(0, 4, (64, 64, 64), 512)
xr, ct = xr.to(dev), ct.to(dev)
losses = []
t0 = time.time()
for it in range(60):
    out = T.train_step(m, crit, opt, None, xr, ct, 1.0)
    losses.append(float(out["total_loss"].detach()))
torch.cuda.synchronize()
print("60 steps on one batch of 4 (64^3, dropout 0.1, bf16): loss", " ".join(f"{l:.4f}" for l in losses[::6]), f"| {time.time()-t0:.1f} s")
assert all(l == l for l in losses) and losses[-1] < 0.8 * losses[0], "loss did not fall"
print("ok")
