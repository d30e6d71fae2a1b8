"""Full-size progressive-cascade training steps (BASELINE configs #4 / #5 geometry) on one MI355X: stage 2 at 128^3
(32768 tokens, 8 heads x 32) and stage 3 at 256^3, earlier stages frozen as in train_progressive_4gpu.py:221-232."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from direct_regression.progressive_cascade import ProgressiveCascadeModel, MultiScaleLoss
from hvc import synthetic
import torch.nn.functional as F

dev = torch.device("cuda:0")
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(0)
model = ProgressiveCascadeModel(use_gradient_checkpointing=(stage == 3)).to(dev).train()
for s in range(1, stage):
    model.freeze_stage(s)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
crit = MultiScaleLoss().to(dev)
size = {1: 64, 2: 128, 3: 256}[stage]
xr, ct = synthetic.batch(0, B, (64, 64, 64), 512)
xr = xr.to(dev)
target = F.interpolate(ct.to(dev), size=(size,) * 3, mode="trilinear", align_corners=False)
print(f"stage {stage}, B={B}, trainable params {sum(p.numel() for p in params)/1e6:.1f} M", flush=True)
for it in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(xr, return_intermediate=True, max_stage=stage)
        ld = crit(out[f"stage{stage}"], target, stage=stage, input_xrays=xr)
    ld["total_loss"].backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: {1e3*(time.perf_counter()-t0):.1f} ms  loss {ld['total_loss'].item():.4f}  "
          f"{ {k: round(v.item(), 4) for k, v in ld.items()} }  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
