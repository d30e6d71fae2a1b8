"""Dev aid (round 4): where does the DDP + hipGraph 64^3 step go non-finite?  One rank, nccl (= RCCL) world size 1.
Prints the loss of every eager warm-up step and every replay for a few DDP / zero_grad variants."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bucket-view", type=int, default=1)
    ap.add_argument("--static-graph", type=int, default=1)
    ap.add_argument("--set-to-none", type=int, default=1)
    ap.add_argument("--side-ctor", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=11)
    ap.add_argument("--clip", type=int, default=1)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(bench._free_port()))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", device_id=dev)
    wl = bench.WORKLOADS["direct64"]
    model, crit, opt = bench.build(wl, dev, capturable=True)
    params = [p for p in model.parameters() if p.requires_grad]
    kw = dict(device_ids=[0], gradient_as_bucket_view=bool(a.bucket_view), bucket_cap_mb=32, static_graph=bool(a.static_graph))
    if a.side_ctor:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            ddp = torch.nn.parallel.DistributedDataParallel(model, **kw)
        torch.cuda.current_stream(dev).wait_stream(side)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(model, **kw)
    xr, ct = bench.make_batch(wl, 0, dev)
    torch.manual_seed(1234)
    log = []

    def step(x, y):
        opt.zero_grad(set_to_none=bool(a.set_to_none))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(ddp(x).float(), y)["total_loss"]
        loss.backward()
        if a.clip:
            torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss

    from hvc.graph import GraphedStep
    orig = step

    def traced(x, y):
        out = orig(x, y)
        if not torch.cuda.is_current_stream_capturing():
            log.append(float(out.item()))
        return out
    g = GraphedStep(traced, [xr, ct], warmup=a.warmup)
    print("variant", vars(a))
    print(" eager warm-up losses:", ["%.4f" % v for v in log])
    rep = []
    for _ in range(5):
        rep.append(float(g(xr, ct).item()))
    print(" replay losses       :", ["%.4f" % v for v in rep])
    gn = torch.stack([p.grad.float().norm() for p in params if p.grad is not None]).norm().item()
    print(" grad norm after last replay:", gn, " params finite:", all(torch.isfinite(p).all().item() for p in params))
    g.close()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
