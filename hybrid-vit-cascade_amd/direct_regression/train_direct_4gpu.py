"""Data-parallel trainer for DirectCTRegression on one MI355X node -- counterpart of the reference's
direct_regression/train_direct_4gpu.py (setup_ddp :25, compute_psnr :40, train_epoch :49, validate :101,
train_ddp :135, main :311) with the same JSON config schema, checkpoint dict and console format.

One process per GPU; gradients are all-reduced by torch DDP over the "nccl" backend, which on ROCm IS RCCL over
xGMI.  MI355X specifics: autocast runs in bfloat16 (the HIP kernels' MFMA dtype; the reference's fp16 GradScaler is
kept as a no-op scale for checkpoint compatibility), gradient buckets are views (no extra copy) sized so the 61 MB
of gradients go out in two reductions overlapped with the backward pass.

    python train_direct_4gpu.py --config config_direct.json [--resume ckpt.pt] [--synthetic]
    python -m torch.distributed.run --nproc-per-node 8 train_direct_4gpu.py --config ...   (torchrun launch)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.utils.data import DataLoader, Subset
from torch.utils.data.distributed import DistributedSampler

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from direct_regression.model_direct import DirectCTRegression, DirectRegressionLoss  # noqa: E402
from hvc import functional as HF  # noqa: E402
from utils.dataset import PatientDRRDataset  # noqa: E402

BUCKET_CAP_MB = 32


def setup_ddp(rank, world_size, backend=None, port="12355"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", port)
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    dist.init_process_group(backend, rank=rank, world_size=world_size)
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())


def cleanup_ddp():
    if dist.is_initialized():
        dist.destroy_process_group()


def compute_psnr(pred, target):
    mse = torch.mean((pred.float() - target.float()) ** 2)
    if mse == 0:
        return float("inf")
    return (20 * torch.log10(2.0 / torch.sqrt(mse))).item()


def wrap_ddp(model, device_ids=None, find_unused_parameters=False):
    """DDP with gradient-as-bucket-view and a bucket size tuned for per-link xGMI rings."""
    return DDP(model, device_ids=device_ids, gradient_as_bucket_view=True, bucket_cap_mb=BUCKET_CAP_MB,
               find_unused_parameters=find_unused_parameters)


def _on_device(dataloader, rank):
    """Batches one ahead on a side stream (utils/prefetch.py) when the loader yields host tensors; anything else passes through."""
    if torch.cuda.is_available():
        from utils.prefetch import DevicePrefetcher
        return DevicePrefetcher(dataloader, torch.device("cuda", rank % max(torch.cuda.device_count(), 1)))
    return dataloader


def train_step(model, criterion, optimizer, scaler, xrays, ct_volume, gradient_clip, autocast_device="cuda",
               autocast_dtype=torch.bfloat16, reducer=None):
    """One optimisation step (reference train_epoch body, :62-75).  Returns the loss dict.
    autocast_dtype=None runs the step in fp32 (split-bf16 MFMA products), the mode the parity fixture uses.
    reducer: an hvc.reducer.BucketedGradReducer over the (unwrapped) model's parameters instead of a DDP-wrapped model."""
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)
    with torch.autocast(autocast_device, dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
        with HF.trace_range("forward"):
            predicted = model(xrays)
        with HF.trace_range("loss"):
            loss_dict = criterion(predicted, ct_volume)
        total_loss = loss_dict["total_loss"]
    if scaler is not None:
        with HF.trace_range("backward"):
            scaler.scale(total_loss).backward()
            if reducer is not None:
                reducer.finish()
        with HF.trace_range("optimizer"):
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_clip)
            scaler.step(optimizer)
            scaler.update()
    else:
        with HF.trace_range("backward"):
            total_loss.backward()
            if reducer is not None:
                reducer.finish()
        with HF.trace_range("optimizer"):
            torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_clip)
            optimizer.step()
    return loss_dict


def train_epoch(model, dataloader, criterion, optimizer, scaler, rank, epoch, config):
    model.train()
    sums = {"total": 0.0, "l1": 0.0, "ssim": 0.0}
    n = 0
    start = time.time()
    world = dist.get_world_size() if dist.is_initialized() else 1
    for batch_idx, batch in enumerate(_on_device(dataloader, rank)):
        xrays = batch["drr_stacked"].cuda(rank, non_blocking=True)
        ct_volume = batch["ct_volume"].cuda(rank, non_blocking=True)
        loss_dict = train_step(model, criterion, optimizer, scaler, xrays, ct_volume, config["training"]["gradient_clip"],
                               reducer=config.get("_grad_reducer"))
        sums["total"] += loss_dict["total_loss"].item()
        sums["l1"] += loss_dict["l1_loss"].item()
        sums["ssim"] += loss_dict["ssim_loss"].item()
        n += 1
        if rank == 0 and batch_idx % 10 == 0:
            sps = (batch_idx + 1) * config["training"]["batch_size"] * world / (time.time() - start)
            print(f"Epoch {epoch} [{batch_idx}/{len(dataloader)}] Loss: {loss_dict['total_loss'].item():.4f} | "
                  f"L1: {loss_dict['l1_loss'].item():.4f} | SSIM: {loss_dict['ssim_loss'].item():.4f} | {sps:.2f} samples/s")
    return {k: v / max(n, 1) for k, v in sums.items()}


def validate(model, dataloader, criterion, rank):
    model.eval()
    sums = {"total": 0.0, "l1": 0.0, "ssim": 0.0}
    n, psnr = 0, 0.0
    with torch.no_grad():
        for batch in dataloader:
            xrays = batch["drr_stacked"].cuda(rank, non_blocking=True)
            ct_volume = batch["ct_volume"].cuda(rank, non_blocking=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                predicted = model(xrays)
                loss_dict = criterion(predicted, ct_volume)
            psnr += compute_psnr(predicted, ct_volume)
            sums["total"] += loss_dict["total_loss"].item()
            sums["l1"] += loss_dict["l1_loss"].item()
            sums["ssim"] += loss_dict["ssim_loss"].item()
            n += 1
    return {k: v / max(n, 1) for k, v in sums.items()}, psnr / max(n, 1)


def save_checkpoint(path, epoch, model, optimizer, scheduler, val_psnr, best_psnr, config):
    """The reference's checkpoint dict (direct_regression/train_direct_4gpu.py:277-298): unwrapped model state, optimizer,
    scheduler, epoch, val_psnr, best_psnr, config."""
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "scheduler_state_dict": scheduler.state_dict(), "val_psnr": val_psnr, "best_psnr": best_psnr,
                "config": {k: v for k, v in config.items() if not k.startswith("_")}}, path)      # (run-time objects, e.g. the gradient reducer, stay out)


def load_checkpoint(path, model, optimizer, scheduler, map_location):
    """--resume (reference :177-189).  A missing file is an error, as torch.load makes it in the reference: silently
    restarting from scratch would later overwrite best_model.  Returns (start_epoch, best_psnr)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"--resume checkpoint not found: {path}")
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt["model_state_dict"])
    optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt["epoch"] + 1, ckpt.get("best_psnr", ckpt.get("val_psnr", 0))


def train_ddp(rank, world_size, config, resume_from=None, synthetic=False):
    launched_by_torchrun = "LOCAL_RANK" in os.environ
    if world_size > 1 or launched_by_torchrun:
        setup_ddp(rank, world_size)
    if rank == 0:
        print("\n" + "=" * 80 + "\nDIRECT CT REGRESSION (NO DIFFUSION) - MI355X data-parallel training\n" + "=" * 80)
    HF.set_fp8_attention(bool(config.get("mi355x", {}).get("fp8_attention", False)))      # opt-in fp8 (e4m3) attention products
    model = DirectCTRegression(**config["model"]).cuda(rank)
    if rank == 0:
        total = sum(p.numel() for p in model.parameters())
        print(f"\nModel parameters: {total:,} ({total / 1e6:.2f}M)")
    # mi355x.grad_exchange: "ddp" (default: torch DistributedDataParallel, as the reference) | "bucketed" (hvc.reducer: one autograd hook per
    # bucket - DDP's per-parameter hooks make the 128^3 step host-bound; module buffers are broadcast once, not every forward)
    ddp_model = model
    if dist.is_initialized() and config.get("mi355x", {}).get("grad_exchange", "ddp") == "bucketed":
        from hvc.reducer import BucketedGradReducer, broadcast_module_state
        broadcast_module_state(model)
        config["_grad_reducer"] = BucketedGradReducer([p for p in model.parameters() if p.requires_grad], bucket_bytes=BUCKET_CAP_MB << 20)
    elif dist.is_initialized():
        ddp_model = wrap_ddp(model, [rank])
    tr = config["training"]
    optimizer = torch.optim.AdamW(ddp_model.parameters(), lr=tr["learning_rate"], weight_decay=tr["weight_decay"], fused=True)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=tr["num_epochs"], eta_min=1e-6)
    scaler = torch.amp.GradScaler("cuda", enabled=False)     # bf16 needs no loss scaling; object kept for API parity
    criterion = DirectRegressionLoss(l1_weight=tr["l1_weight"], ssim_weight=tr["ssim_weight"])

    start_epoch, best_psnr = 1, 0.0
    if resume_from is not None:
        if rank == 0:
            print(f"\nResuming from checkpoint: {resume_from}")
        start_epoch, best_psnr = load_checkpoint(resume_from, model, optimizer, scheduler, f"cuda:{rank}")
        if rank == 0:
            print(f"Resuming from epoch {start_epoch - 1}\nBest PSNR so far: {best_psnr:.2f} dB")

    data = config["data"]
    full = PatientDRRDataset(data_path=None if synthetic else data["dataset_path"], target_xray_size=config["model"]["xray_img_size"],
                             target_volume_size=tuple(config["model"]["volume_size"]), max_patients=data["max_patients"])
    n_train = int(len(full) * 0.8)
    train_ds, val_ds = Subset(full, range(n_train)), Subset(full, range(n_train, len(full)))
    train_sampler = DistributedSampler(train_ds, num_replicas=world_size, rank=rank, shuffle=True) if dist.is_initialized() else None
    val_sampler = DistributedSampler(val_ds, num_replicas=world_size, rank=rank, shuffle=False) if dist.is_initialized() else None
    train_loader = DataLoader(train_ds, batch_size=tr["batch_size"], sampler=train_sampler, shuffle=train_sampler is None,
                              num_workers=data["num_workers"], pin_memory=True)
    val_loader = DataLoader(val_ds, batch_size=tr["batch_size"], sampler=val_sampler, shuffle=False,
                            num_workers=data["num_workers"], pin_memory=True)
    save_dir = config["checkpoints"]["save_dir"]
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)

    for epoch in range(start_epoch, tr["num_epochs"] + 1):
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        if rank == 0:
            print(f"\n{'=' * 80}\nEpoch {epoch}/{tr['num_epochs']}\nLearning rate: {scheduler.get_last_lr()[0]:.6f}\n{'=' * 80}")
        losses = train_epoch(ddp_model, train_loader, criterion, optimizer, scaler, rank, epoch, config)
        val_losses, val_psnr = validate(ddp_model, val_loader, criterion, rank)
        if rank == 0:
            print(f"\nEpoch {epoch} Training Summary:\n  Total Loss: {losses['total']:.4f}\n  L1 Loss: {losses['l1']:.4f}\n"
                  f"  SSIM Loss: {losses['ssim']:.4f}\n\nValidation Results:\n  Total Loss: {val_losses['total']:.4f}\n"
                  f"  L1 Loss: {val_losses['l1']:.4f}\n  SSIM Loss: {val_losses['ssim']:.4f}\n  PSNR: {val_psnr:.2f} dB")
            if val_psnr > best_psnr:                     # file names as the reference writes them (:277, :290)
                best_psnr = val_psnr
                save_checkpoint(os.path.join(save_dir, "best_model.pt"), epoch, model, optimizer, scheduler, val_psnr, best_psnr, config)
                print(f"  ✓ New best model saved! PSNR: {best_psnr:.2f} dB")
            if epoch % config["checkpoints"]["save_every"] == 0:
                save_checkpoint(os.path.join(save_dir, f"checkpoint_epoch_{epoch}.pt"), epoch, model, optimizer, scheduler, val_psnr,
                                best_psnr, config)
                print(f"  ✓ Periodic checkpoint saved at epoch {epoch}")
        scheduler.step()
    if rank == 0:
        print(f"\n{'=' * 80}\nTraining Complete!\nBest PSNR: {best_psnr:.2f} dB\n{'=' * 80}")
    cleanup_ddp()


def main():
    from hvc.dist_env import ensure_rccl_env
    ensure_rccl_env()          # before any HIP call of this process and of the ranks it spawns
    ap = argparse.ArgumentParser(description="Direct CT regression, data-parallel on MI355X")
    ap.add_argument("--config", type=str, default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_direct.json"))
    ap.add_argument("--resume", type=str, default=None)
    ap.add_argument("--synthetic", action="store_true", help="train on the seeded synthetic phantoms")
    args = ap.parse_args()
    with open(args.config) as f:
        config = json.load(f)
    if "LOCAL_RANK" in os.environ:          # launched by torchrun: one process per GPU already exists
        train_ddp(int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"]), config, args.resume, args.synthetic)
        return
    world_size = torch.cuda.device_count()
    print(f"Using {world_size} GPUs")
    if world_size > 1:
        mp.spawn(train_ddp, args=(world_size, config, args.resume, args.synthetic), nprocs=world_size, join=True)
    else:
        train_ddp(0, 1, config, args.resume, args.synthetic)


if __name__ == "__main__":
    main()
