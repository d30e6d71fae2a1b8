#!/usr/bin/env python3
"""Headline benchmark: CT volumes/s, forward + backward + optimizer step, direct model.

    python bench.py --gpus N --steps K --warmup W [--workload direct128|direct64]

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run); weak scaling: every
rank trains on its own batch, gradients are all-reduced over RCCL/xGMI by torch DDP.  A "step" is
the reference's training step (direct_regression/train_direct_4gpu.py:59-75): zero_grad -> autocast
forward -> L1 + 0.5 SSIM -> backward -> clip_grad_norm_(1.0) -> AdamW, train mode (dropout 0.1),
synthetic data resident in HBM, random-init weights.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  direct128 (default; the config the metric is quoted on, configs[2] per-GPU slice): 128^3, batch 2 per GPU,
            bf16, token grid 32^3 = 32768 tokens (A2-fix), no gradient checkpointing
            (activations fit easily in 288 GB, so the recompute pass is not needed).
  direct64  (configs[1]): 64^3, batch 4, bf16, 4096 tokens.
"""
import argparse
import math
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "direct128": dict(volume=(128, 128, 128), batch=2, name="Direct baseline 128^3, bf16, batch=2/GPU (BASELINE configs[2] per-GPU slice)"),
    "direct64": dict(volume=(64, 64, 64), batch=4, name="Direct baseline 64^3, bf16, batch=4 (BASELINE configs[1])"),
    "direct256": dict(volume=(256, 256, 256), batch=1, name="Direct model at 256^3 (north_star's third resolution), bf16, batch=1/GPU"),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense, /opt/skills/guides/MI355X_MICROARCH.md
ATTN_KERNELS = {"attn_fwd_kernel", "attn_fwd2_kernel", "attn_bwd_dkv_kernel", "attn_bwd_dq_kernel", "attn_delta_kernel"}


def fwd_flops_per_volume(model):
    """Algorithmic forward FLOPs per volume (2 FLOP / MAC), SURVEY.md §8(d) formula."""
    vb = model.vit_backbone
    N = vb.downsampled_size[0] * vb.downsampled_size[1] * vb.downsampled_size[2]
    C = vb.voxel_dim
    Cc = model.xray_encoder.embed_dim
    M = 64 * 64
    L = len(vb.blocks)
    block = 28 * N * C * C + 4 * N * C * (N + M) + 4 * M * Cc * C
    # conv stems: 2 * Cout * Cin * k^d * output positions
    xray = 2 * 2 * (64 * 1 * 49 * 256 * 256 + 128 * 64 * 9 * 128 * 128 + Cc * 128 * 9 * 64 * 64)
    vox, grid, cin = 0, list(model.volume_size), 1
    for layer in vb.voxel_embed:
        if isinstance(layer, torch.nn.Conv3d):
            grid = [(g - 1) // layer.stride[0] + 1 for g in grid]
            vox += 2 * layer.out_channels * layer.in_channels * 27 * grid[0] * grid[1] * grid[2]
    return xray + vox + L * block, dict(N=N, M=M, C=C, L=L)


def build(workload, device, capturable=False):
    from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss
    torch.manual_seed(0)
    model = DirectCTRegression(volume_size=workload["volume"], xray_img_size=512, voxel_dim=256, vit_depth=4,
                               num_heads=4, xray_feature_dim=512).to(device).train()
    crit = DirectRegressionLoss(1.0, 0.5)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, fused=True, capturable=capturable)
    return model, crit, opt


def make_batch(workload, rank, device):
    from hvc import synthetic
    xr, ct = synthetic.batch(1000 * rank, workload["batch"], workload["volume"], 512)
    return xr.to(device), ct.to(device)


def train_step(model, params, crit, opt, xr, ct, reducer=None):
    if reducer is not None:
        reducer.zero_grad()                   # gradients are views of the reducer's flat buffer: zeroed, not dropped
    else:
        opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(xr)
        loss = crit(pred.float(), ct)["total_loss"]
    loss.backward()
    if reducer is not None:
        reducer.finish()                      # the bucketed all-reduces were launched from inside backward; join them
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    return loss


def _host_cores():
    """(threads to use, description): the physical cores this process may run on (one GPU's share of the host on a shared box)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    model, phys, cur = "unknown", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur.get("processor") in allowed:
                    phys.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
                cur = {}
                continue
            k, v = (x.strip() for x in line.split(":", 1))
            if k == "model name":
                model = v
            if k in ("processor", "physical id", "core id"):
                cur[k] = int(v)
    except OSError:
        pass
    n = len(phys) if phys else len(allowed)
    desc = f"{model}: {n} physical cores of {len(allowed)} logical CPUs visible to this process"
    try:      # a container's CPU share (cgroup v2 quota): more threads than that only get throttled
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            share = max(1, int(int(quota) / int(period)))
            if share < n:
                desc += f"; cgroup CPU quota {share} cores"
                n = share
    except (OSError, ValueError):
        pass
    return max(1, n), desc


def cpu_baseline():
    """Oracle (CPU restatement pinned to the reference) timed on the host cores, protocol of BASELINE.md section 3: BASELINE
    config #1 (Direct 64^3, 2-view 512^2, batch 1, fp32, train mode), one step = zero_grad -> forward -> L1 + 0.5 SSIM ->
    backward -> clip_grad_norm_(1.0) -> AdamW(1e-4, wd 0.01); torch.set_num_threads(all physical cores); 2 warm-up + 5 timed
    steps, median (1 + 3 if a step takes longer than 12 s, so that the default run stays within a few minutes)."""
    from direct_regression.model_direct import DirectCTRegression
    from hvc import synthetic
    from oracle import hvc_oracle as O
    cores, host = _host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(64, 64, 64))
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k)
         for k, v in m.state_dict().items()}
    leaves = [v for v in P.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-4, weight_decay=0.01)
    xr, ct = synthetic.batch(0, 1, (64, 64, 64), 512)

    def one_step():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        pred = O.direct_ct_regression(xr, P, training=True, new_stats={}, p_drop=0.1)
        loss = O.direct_regression_loss(pred, ct)["total_loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 1.0)
        opt.step()
        return time.perf_counter() - t0
    first = one_step()
    warm, timed = (2, 5) if first <= 12.0 else (1, 3)
    for _ in range(warm - 1):
        one_step()
    times = sorted(one_step() for _ in range(timed))
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "volumes/s", "cores": cores, "kind": "port", "host": host,
            "torch": torch.__version__, "s_per_step_median": med, "s_per_step_all": times,
            "sample": f"{warm} warm-up + {timed} timed train steps of Direct 64^3, batch 1, fp32 (BASELINE config #1: fwd + L1+0.5*SSIM + "
                      f"bwd + clip + AdamW, train mode with the reference's dropout draws p=0.1) through oracle/hvc_oracle.py on "
                      f"{cores} threads, median step {med:.2f} s"}


def other_resolutions(main_workload):
    """north_star quotes volumes/s at 64^3, 128^3 and 256^3 "as absolute and as fraction of roofline": short runs of the other two
    resolutions (child processes of this one, 10 timed steps each, HIP events on their attention kernels), each with its own
    roofline object (dominant kernel, algorithmic flops / measured launch time / dense bf16 MFMA peak) beside the headline line."""
    import subprocess
    out = {}
    for wl in ("direct64", "direct128", "direct256"):
        if wl == main_workload:
            continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
                            "--no-extra"], capture_output=True, text=True, timeout=900)
        line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
        if r.returncode or line is None:
            out[wl] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        d = json.loads(line)
        out[wl] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "batch_per_gpu": d["config"]["batch_per_gpu"],
                   "tokens": d["config"]["tokens"], "launch": d["config"]["launch"], "steps": d["steps"],
                   "algorithmic_tflops_per_volume_fwd_bwd": d["algorithmic_tflops_per_volume_fwd_bwd"],
                   "achieved_model_tflops_per_gpu": d["achieved_model_tflops_per_gpu"],
                   "model_frac_of_mfma_peak": d["achieved_model_tflops_per_gpu"] / MFMA_BF16_PEAK_TFLOPS,
                   "roofline": d.get("roofline"),
                   "kernel_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in d.get("kernel_time_share", {}).items()}}
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n):
    """Run this script as n ranks under torch.distributed.run (one process per GPU); returns the launcher's exit code."""
    import subprocess
    from hvc.dist_env import ensure_rccl_env
    env = ensure_rccl_env(dict(os.environ))                 # dmabuf IPC for RCCL: one definition for every launcher (hvc/dist_env.py)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="direct128", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--graph", default="auto", choices=("auto", "on", "off"),
                    help="replay the step as one captured hipGraph (auto: for the launch-bound direct64 workload, single GPU)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short 64^3 / 256^3 runs reported under other_resolutions")
    ap.add_argument("--torch-ddp", action="store_true",
                    help="eager multi-rank steps through torch DistributedDataParallel instead of hvc.reducer.BucketedGradReducer "
                         "(A/B: DDP's per-parameter hooks cost ~19 ms of host time per 128^3 step)")
    ap.add_argument("--ddp", action="store_true",
                    help="with --gpus 1: still initialise the nccl (= RCCL) process group (world size 1) and wrap the model in DDP, so that "
                         "the multi-GPU code path - RCCL init, bucketed all-reduce hooks on the HIP autograd Functions - runs on one GPU")
    args = ap.parse_args()
    if os.environ.get("HVC_FORCE_DDP") == "1":
        args.ddp = True
    from hvc.dist_env import ensure_rccl_env
    ensure_rccl_env()          # also when an external launcher started this rank; no HIP call has been made yet

    if args.gpus > 1 and "RANK" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (the reference does the same with mp.spawn,
        # direct_regression/train_direct_4gpu.py:311-339).  The parent has made no GPU call yet and makes none: it
        # only waits for the torch.distributed.run child and passes on its exit code (rank 0 prints the JSON line).
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world} of the launcher")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HVC hot path has no CPU fallback")
    # HVC_TEST_SINGLE_DEVICE / HVC_DIST_BACKEND exist only to rehearse the N > 1 code path on a one-GPU box
    # (all ranks on cuda:0, gloo); the real launch is one rank per GPU over "nccl" = RCCL / xGMI.
    single = os.environ.get("HVC_TEST_SINGLE_DEVICE") == "1"
    device = torch.device("cuda", 0 if single else local_rank)
    torch.cuda.set_device(device)
    distributed = world > 1 or args.ddp
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                        # --ddp on one GPU: no launcher has set the rendezvous
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("HVC_DIST_BACKEND", "nccl")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the one JSON line, so file
        # descriptor 1 points at stderr until the communicator exists (a barrier forces it)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=device)   # RCCL over xGMI
            else:
                torch.distributed.init_process_group(backend)
            torch.distributed.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from hvc import ops, stem
    wl = WORKLOADS[args.workload]
    # hipGraph replay of the whole step where the step is launch-bound (64^3: ~300 launches in ~10 ms of kernels); the
    # 128^3 / 256^3 steps are GPU-bound and run eagerly, which also lets HIP events bracket their kernels in the timed region
    # Under DDP the captured step contains the bucketed RCCL all-reduces (torch records NCCL collectives into a capture): DDP is
    # built with static_graph=True on the stream the step is warmed up (>= 11 eager iterations: DDP's reducer rebuilds its buckets
    # and settles its hook order over the first ones) and captured on.
    use_graph = args.graph == "on" or (args.graph == "auto" and args.workload == "direct64")
    if distributed and torch.distributed.get_backend() != "nccl":
        use_graph = False                                  # rehearsal backends (gloo) cannot be captured
    model, crit, opt = build(wl, device, capturable=use_graph)
    params = [p for p in model.parameters() if p.requires_grad]
    fwd_flops, geom = fwd_flops_per_volume(model)
    step_model = model
    graph_stream = None
    reducer = None
    if distributed:
        ddp_kw = dict(device_ids=[device.index], gradient_as_bucket_view=True, bucket_cap_mb=32, static_graph=use_graph)
        if use_graph:       # whole-step capture: DDP constructor, warm-up and capture on ONE side stream (hvc/graph.py: GraphedStep)
            graph_stream = torch.cuda.Stream(device=device)
            graph_stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(graph_stream):
                step_model = torch.nn.parallel.DistributedDataParallel(model, **ddp_kw)
            torch.cuda.current_stream(device).wait_stream(graph_stream)
        elif args.torch_ddp:   # eager: on the stream the steps run on (its AccumulateGrad nodes then sit on that stream too)
            step_model = torch.nn.parallel.DistributedDataParallel(model, **ddp_kw)
        else:
            # eager steps: bucketed all-reduce of a flat gradient buffer driven by one autograd hook per bucket (hvc/reducer.py) - DDP's
            # contract and overlap without its per-parameter host work, which makes the 128^3 step host-bound
            from hvc.reducer import BucketedGradReducer, broadcast_module_state
            broadcast_module_state(model)
            reducer = BucketedGradReducer(params, bucket_bytes=32 << 20)
    xr, ct = make_batch(wl, rank, device)
    # identical initial weights on every rank (seed 0 in build(); DDP broadcasts rank 0's anyway), but each
    # data-parallel replica draws its own dropout seeds, as the reference's per-process generators do
    torch.manual_seed(1234 + rank)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def eager_step():
        return train_step(step_model, params, crit, opt, xr, ct, reducer)
    step = eager_step
    if use_graph:
        from hvc.graph import GraphedStep
        graphed = GraphedStep(lambda a_, b_: train_step(step_model, params, crit, opt, a_, b_), [xr, ct],
                              warmup=max(args.warmup, 11 if distributed else 3), stream=graph_stream)
        step = lambda: graphed(xr, ct)                    # noqa: E731
    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region: K steps.  Eager mode: only the attention kernels (the dominant-kernel candidates, 32 launches per step)
    # are bracketed by HIP events here, which perturbs the step by < 1 %; the full per-kernel table comes from an extra,
    # untimed pass below (bracketing all ~300 launches per step costs 40 % at 64^3).  Graph mode: a replay hides the
    # individual launches, so both tables come from eager passes after the timed region.
    # Under DDP the events come from extra steps after the timed region as well: with the nccl process group alive, HIP-event
    # records inside the step cost 4 ms per 128^3 step (measured: 56.6 ms plain, 57.4 ms DDP without events, 61.1 ms DDP with events -
    # the process group's watchdog threads contend for the runtime's event lock), which would be charged to every N > 1 point of a
    # scaling curve but not to its N = 1 point.
    prof = None if (args.no_profile or use_graph or distributed) else []
    ops.PROFILE, ops.PROFILE_ONLY = prof, ATTN_KERNELS
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue_s = time.perf_counter() - t0          # host time to ENQUEUE the K steps (the launch thread runs ahead of the GPU)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.PROFILE, ops.PROFILE_ONLY = None, None
    if use_graph:
        loss = loss.detach().clone()          # the graph's static output
        graphed.close()                       # the eager passes below take their dropout seeds as passed again
    rank_ms = None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        rank_ms = [1e3 * x.item() / args.steps for x in every]
        elapsed = max(x.item() for x in every)               # the slowest rank's clock is the job's
    roofline_note = f"the {args.steps} timed steps (HIP events on the launch stream)"
    if (use_graph or distributed) and not args.no_profile:
        prof = []
        ops.PROFILE, ops.PROFILE_ONLY = prof, ATTN_KERNELS
        for _ in range(min(args.steps, 5)):      # every rank runs them: the steps contain DDP's collectives
            eager_step()
        torch.cuda.synchronize()
        ops.PROFILE, ops.PROFILE_ONLY = None, None
        roofline_note = (f"{min(args.steps, 5)} eager steps after the timed region (HIP events on the launch stream; the timed region "
                         + ("replays a hipGraph, which hides individual launches)" if use_graph else
                            "runs under DDP, where event records inside the step perturb it by 7 %)"))
    full = None
    if prof is not None and rank == 0 and world == 1:
        full = []
        ops.PROFILE = full
        extra = min(args.steps, 3)
        for _ in range(extra):
            eager_step()
        torch.cuda.synchronize()
        ops.PROFILE = None

    if not math.isfinite(float(loss.item())):
        raise RuntimeError(f"rank {rank}: non-finite training loss {loss.item()} after {args.warmup + args.steps} steps - the measurement is void")
    if rank == 0:
        vols = wl["batch"] * world * args.steps
        value = vols / elapsed
        out = {
            "metric": f"CT volumes/sec fwd+bwd @{wl['volume'][0]}^3 direct model",
            "value": value, "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl["name"], "volume": list(wl["volume"]), "batch_per_gpu": wl["batch"],
                       "global_batch": wl["batch"] * world, "tokens": geom["N"], "context_tokens": geom["M"],
                       "layers": geom["L"], "parallelism": f"dp{world}", "train_mode_dropout": 0.1,
                       "gradient_checkpointing": False, "optimizer": "AdamW(fused) + clip_grad_norm 1.0",
                       "launch": "hipGraph replay of the captured step" if use_graph else "eager (one launch per kernel)",
                       "loss": float(loss.item()), "stage_backends": stem.STAGE_BACKEND},
            "algorithmic_tflops_per_volume_fwd_bwd": 3 * fwd_flops / 1e12,
            "achieved_model_tflops_per_gpu": 3 * fwd_flops * wl["batch"] * args.steps / elapsed / 1e12,
            "host_enqueue_ms_per_step": 1e3 * enqueue_s / args.steps,      # rank 0's launch thread; close to ms_per_step = host-bound
        }
        if prof:
            torch.cuda.synchronize()
            agg = {}
            for name, work, s, e in prof:
                a = agg.setdefault(name, [0.0, 0.0, 0])
                a[0] += s.elapsed_time(e) * 1e-3
                a[1] += work
                a[2] += 1
            # dominant kernel by measured device time (HIP events bracketing each launch on the launch stream)
            dom = max(agg, key=lambda k: agg[k][0])
            tsec, work, n = agg[dom]
            traffic = None
            try:   # HBM bytes per launch from the newest committed rocprofv3 --pmc passes (profiles/), same kernel & shape
                import glob
                src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
                traffic = json.load(open(src)).get(args.workload, {}).get(dom)
                out["roofline_traffic_source"] = os.path.relpath(src, ROOT)
            except (OSError, ValueError, IndexError):
                pass
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": work / tsec / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": work / tsec / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                               "launches": n, "avg_launch_ms": 1e3 * tsec / n,
                               "algorithmic_flops_per_launch": work / n}
            out["roofline"]["measured_over"] = roofline_note
            share_src, share_steps = (full, min(args.steps, 3)) if full else (prof, min(args.steps, 5) if (use_graph or distributed) else args.steps)
            agg2 = {}
            for name, work, s_ev, e_ev in share_src:
                a = agg2.setdefault(name, [0.0, 0.0, 0])
                a[0] += s_ev.elapsed_time(e_ev) * 1e-3
                a[1] += work
                a[2] += 1
            out["kernel_time_share"] = {k: {"ms_per_step": 1e3 * v[0] / share_steps, "tflops": v[1] / v[0] / 1e12 if v[0] else None,
                                            "launches_per_step": v[2] / share_steps, "avg_launch_ms": 1e3 * v[0] / v[2]}
                                        for k, v in sorted(agg2.items())}
            out["kernel_time_share_note"] = "from an extra untimed pass with every attention / GEMM launch bracketed" if full else "timed region"
        if distributed:
            out["rccl_ranks"] = torch.distributed.get_world_size()
            out["dist_backend"] = torch.distributed.get_backend()
            out["ddp"] = {"bucket_cap_mb": 32, "gradient_as_bucket_view": True, "static_graph": bool(use_graph),
                          "captured_in_hipgraph": bool(use_graph),
                          "gradient_exchange": reducer.describe() if reducer is not None else "torch DistributedDataParallel"}
            if rank_ms is not None:
                out["rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms)}
            out["ipc_env"] = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
            # scaling comparator: the N = 1 point of a scaling curve is this same command with --gpus 1 (same launch mode)
            out["n1_comparator"] = f"python bench.py --workload {args.workload} --graph {args.graph}"
            try:
                out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:       # noqa: BLE001 - informational only
                out["nccl_version"] = None
        if world == 1 and not args.no_extra:
            out["other_resolutions"] = other_resolutions(args.workload)
        if not args.no_cpu_baseline and world == 1:      # host-CPU baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
