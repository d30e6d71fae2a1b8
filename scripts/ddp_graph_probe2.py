"""Dev aid (round 4): DDP + hipGraph on one stream.  DDP constructor, >= 11 eager warm-up iterations and the capture all on ONE side
stream (so the AccumulateGrad nodes DDP hooks into live on the capture stream); checks whether the capture pass itself executed
anything (weight checksum before / after) and what is non-finite after the first replay."""
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
    ap.add_argument("--ddp", type=int, default=1)
    ap.add_argument("--same-stream", type=int, default=1)
    ap.add_argument("--broadcast-buffers", type=int, default=1)
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
    xr, ct = bench.make_batch(wl, 0, dev)
    torch.manual_seed(1234)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    net = model

    def step(x, y):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(net(x).float(), y)["total_loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss

    def checksum():
        return float(torch.stack([p.detach().double().abs().sum() for p in params]).sum().item())

    with torch.cuda.stream(s):
        if a.ddp:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True, bucket_cap_mb=32,
                                                            static_graph=True, broadcast_buffers=bool(a.broadcast_buffers))
        warm = [float(step(xr, ct).item()) for _ in range(11)]
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    print("variant", vars(a))
    print(" eager losses:", ["%.4f" % v for v in warm[-4:]])
    c0 = checksum()
    g = torch.cuda.CUDAGraph()
    if a.same_stream:
        with torch.cuda.graph(g, stream=s):
            out = step(xr, ct)
    else:
        with torch.cuda.graph(g):
            out = step(xr, ct)
    torch.cuda.synchronize()
    c1 = checksum()
    print(" weight checksum before / after the capture pass:", c0, c1, "(equal: the capture executed nothing)" if c0 == c1 else "(DIFFERENT: work ran outside the capture)")
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        bad_g = sum(int(not torch.isfinite(p.grad).all().item()) for p in params if p.grad is not None)
        bad_p = sum(int(not torch.isfinite(p).all().item()) for p in params)
        print(f" replay {i}: loss {out.item():.4f}, non-finite grads {bad_g} / {len(params)}, non-finite params {bad_p}")
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
