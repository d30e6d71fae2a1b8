"""Stage-wise data-parallel trainer for ProgressiveCascadeModel on one MI355X node -- counterpart of the
reference's direct_regression/progressive_cascade/train_progressive_4gpu.py (setup_ddp :31, resize_ct_volume :46,
train_epoch :60, validate :143, train_stage :200, main_worker :362, main :382) with the same JSON config schema,
per-stage checkpoint names (stage{n}_best.pth, stage{n}_epoch{e}.pth) and checkpoint dict.

Stage n loads stage n-1's best checkpoint (strict=False), freezes stages < n and trains stage n against the CT
resized to the stage resolution.  One process per GPU; gradients are all-reduced by torch DDP over "nccl" (= RCCL
over xGMI on ROCm).  MI355X specifics: bf16 autocast (no loss scaling needed; the GradScaler object is kept, disabled,
for API parity), gradient buckets as views; frozen stages keep no autograd graph (their parameters do not require
grad and their inputs are data), so DDP reduces only the trainable stage's gradients (find_unused_parameters stays
on, as in the reference, because the X-ray encoder heads of the other stages receive no gradient).

    python train_progressive_4gpu.py [--config config_progressive.json] [--stages 1 2 3] [--synthetic]
    python -m torch.distributed.run --nproc-per-node 8 train_progressive_4gpu.py ...      (torchrun launch)
"""
import argparse
import json
import os
import sys
import time
from datetime import datetime
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (_PKG, os.path.join(_PKG, "direct_regression")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from hvc import functional as HF
from hvc import stem  # noqa: E402
from progressive_cascade.model_progressive import ProgressiveCascadeModel  # noqa: E402
from progressive_cascade.loss_multiscale import MultiScaleLoss, compute_psnr, compute_ssim_metric  # noqa: E402
from utils.dataset import PatientDRRDataset  # noqa: E402

BUCKET_CAP_MB = 32
STAGE_SIZES = {1: (64, 64, 64), 2: (128, 128, 128), 3: (256, 256, 256)}


def setup_ddp(rank, world_size, backend=None, port="12366"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", port)
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    dist.init_process_group(backend, rank=rank, world_size=world_size)
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())


def cleanup_ddp():
    if dist.is_initialized():
        dist.destroy_process_group()


def resize_ct_volume(ct_volume, target_size):
    """(B,1,D,H,W) -> (B,1,D',H',W'), trilinear, align_corners=False (reference :46-57), on the HIP resize kernel."""
    if tuple(ct_volume.shape[2:]) == tuple(target_size):
        return ct_volume
    return stem.upsample_trilinear(ct_volume, tuple(target_size), align_corners=False)


def cascade_step(model, criterion, xrays, ct_volume_orig, max_stage):
    """Forward + loss of the stage being trained (reference train_epoch :93-114)."""
    target = resize_ct_volume(ct_volume_orig, STAGE_SIZES[max_stage])
    if max_stage == 1:
        pred = model(xrays, max_stage=1)
        return pred, target, criterion(pred, target, stage=1)
    outputs = model(xrays, return_intermediate=True, max_stage=max_stage)
    pred = outputs[f"stage{max_stage}"]
    if max_stage == 2:
        return pred, target, criterion(pred, target, stage=2)
    return pred, target, criterion(pred, target, stage=3, input_xrays=xrays)


def _on_device(dataloader, rank):
    """Batches one ahead on a side stream (utils/prefetch.py) when the loader yields host tensors; anything else passes through."""
    if torch.cuda.is_available():
        from utils.prefetch import DevicePrefetcher
        return DevicePrefetcher(dataloader, torch.device("cuda", rank % max(torch.cuda.device_count(), 1)))
    return dataloader


def train_step(model, criterion, optimizer, scaler, xrays, ct_volume_orig, max_stage, gradient_clip, autocast_device="cuda"):
    optimizer.zero_grad(set_to_none=True)
    with torch.autocast(autocast_device, dtype=torch.bfloat16):
        _, _, loss_dict = cascade_step(model, criterion, xrays, ct_volume_orig, max_stage)
        total_loss = loss_dict["total_loss"]
    params = [p for p in model.parameters() if p.requires_grad]
    if scaler is not None:
        scaler.scale(total_loss).backward()
        scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(params, gradient_clip)
        scaler.step(optimizer)
        scaler.update()
    else:
        total_loss.backward()
        torch.nn.utils.clip_grad_norm_(params, gradient_clip)
        optimizer.step()
    return loss_dict


def train_epoch(model, dataloader, criterion, optimizer, scaler, rank, epoch, stage, config, max_stage=1):
    model.train()
    epoch_losses = {"total": 0.0}
    num_batches = 0
    start = time.time()
    for batch_idx, batch in enumerate(_on_device(dataloader, rank)):
        xrays = batch["drr_stacked"].cuda(rank, non_blocking=True)
        ct = batch["ct_volume"].cuda(rank, non_blocking=True)
        loss_dict = train_step(model, criterion, optimizer, scaler, xrays, ct, max_stage, config["training"]["gradient_clip"])
        for k, v in loss_dict.items():
            epoch_losses[k] = epoch_losses.get(k, 0.0) + float(v)
        epoch_losses["total"] += float(loss_dict["total_loss"])
        num_batches += 1
        if rank == 0 and batch_idx % 10 == 0:
            bps = (batch_idx + 1) / (time.time() - start)
            print(f"Epoch {epoch} | Stage {max_stage} | Batch {batch_idx}/{len(dataloader)} | "
                  f"Loss: {float(loss_dict['total_loss']):.4f} | {bps:.2f} batch/s")
    return {k: v / max(num_batches, 1) for k, v in epoch_losses.items()}


def validate(model, dataloader, criterion, rank, stage, max_stage=1):
    model.eval()
    val_losses = {"total": 0.0}
    psnr = ssim = 0.0
    num_batches = 0
    with torch.no_grad():
        for batch in dataloader:
            xrays = batch["drr_stacked"].cuda(rank, non_blocking=True)
            ct = batch["ct_volume"].cuda(rank, non_blocking=True)
            target = resize_ct_volume(ct, STAGE_SIZES[max_stage])
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(xrays, max_stage=max_stage)
                loss_dict = criterion(pred, target, stage=max_stage, input_xrays=xrays) if max_stage == 3 else \
                    criterion(pred, target, stage=max_stage)
            for k, v in loss_dict.items():
                val_losses[k] = val_losses.get(k, 0.0) + float(v)
            val_losses["total"] += float(loss_dict["total_loss"])
            psnr += compute_psnr(pred, target)
            ssim += compute_ssim_metric(pred.float(), target)
            num_batches += 1
    out = {k: v / max(num_batches, 1) for k, v in val_losses.items()}
    out["psnr"], out["ssim"] = psnr / max(num_batches, 1), ssim / max(num_batches, 1)
    return out


def wrap_ddp(model, device_ids=None):
    return DDP(model, device_ids=device_ids, gradient_as_bucket_view=True, bucket_cap_mb=BUCKET_CAP_MB,
               find_unused_parameters=True)


def build_stage(config, stage, checkpoint_dir, device, rank=0):
    """Model for `stage` with the earlier stages loaded and frozen, the loss, optimizer, scheduler (reference :212-262)."""
    m = config["model"]
    model = ProgressiveCascadeModel(xray_img_size=m["xray_img_size"], xray_feature_dim=m["xray_feature_dim"],
                                    voxel_dim=m["voxel_dim"], use_gradient_checkpointing=(stage == 3)).to(device)
    if stage > 1:
        prev = Path(checkpoint_dir) / f"stage{stage - 1}_best.pth"
        if prev.exists():
            if rank == 0:
                print(f"Loading Stage {stage - 1} checkpoint: {prev}")
            ckpt = torch.load(prev, map_location=device, weights_only=False)
            model.load_state_dict(ckpt["model_state_dict"], strict=False)
            for prev_stage in range(1, stage):          # frozen only once their weights are loaded (reference :230-232):
                model.freeze_stage(prev_stage)          # freezing random weights would train the new stage on noise
        elif rank == 0:
            print(f"Warning: Stage {stage - 1} checkpoint not found!")
    criterion = MultiScaleLoss(config={k: config["loss"][k] for k in ("stage1", "stage2", "stage3")}).to(device)
    key = f"stage{stage}"
    lr = config["training"][key]["learning_rate"]
    optimizer = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr,
                                  weight_decay=config["training"]["weight_decay"], fused=device.type == "cuda")
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config["training"][key]["num_epochs"], eta_min=lr * 0.1)
    return model, criterion, optimizer, scheduler


def train_stage(rank, world_size, config, stage, checkpoint_dir, synthetic=False):
    launched_by_torchrun = "LOCAL_RANK" in os.environ
    if (world_size > 1 or launched_by_torchrun) and not dist.is_initialized():
        setup_ddp(rank, world_size)
    if rank == 0:
        print(f"\n{'=' * 60}\nTraining Stage {stage}\n{'=' * 60}\n")
    device = torch.device("cuda", rank % max(torch.cuda.device_count(), 1))
    # "mi355x": {"fp8_attention": true} runs the attention forward's Q K^T / P V products as fp8 (e4m3) MFMAs (BASELINE configs[4])
    HF.set_fp8_attention(bool(config.get("mi355x", {}).get("fp8_attention", False)))
    # "gradient_checkpointing": "auto" (default: recompute the stage-3 ViT only if its activations would not fit in HBM) | "on" | "off"
    HF.set_checkpoint_policy(config.get("mi355x", {}).get("gradient_checkpointing", "auto"))
    model, criterion, optimizer, scheduler = build_stage(config, stage, checkpoint_dir, device, rank)
    ddp_model = wrap_ddp(model, [device.index]) if dist.is_initialized() else model
    scaler = torch.amp.GradScaler("cuda", enabled=False)      # bf16: no loss scaling; object kept for API parity

    data, key = config["data"], f"stage{stage}"
    ds_kwargs = dict(root_dir=None if synthetic else data["dataset_path"], max_patients=data["max_patients"],
                     train_split=data["train_split"], val_split=data["val_split"],
                     target_xray_size=config["model"]["xray_img_size"], target_volume_size=STAGE_SIZES[stage])
    train_dataset = PatientDRRDataset(split="train", **ds_kwargs)
    val_dataset = PatientDRRDataset(split="val", **ds_kwargs)
    train_sampler = DistributedSampler(train_dataset, num_replicas=world_size, rank=rank) if dist.is_initialized() else None
    val_sampler = DistributedSampler(val_dataset, num_replicas=world_size, rank=rank) if dist.is_initialized() else None
    batch_size = config["training"][key]["batch_size"]
    train_loader = DataLoader(train_dataset, batch_size=batch_size, sampler=train_sampler, shuffle=train_sampler is None,
                              num_workers=data["num_workers"], pin_memory=True)
    val_loader = DataLoader(val_dataset, batch_size=batch_size, sampler=val_sampler, num_workers=data["num_workers"], pin_memory=True)

    num_epochs = config["training"][key]["num_epochs"]
    best_val_loss = float("inf")
    for epoch in range(1, num_epochs + 1):
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        train_losses = train_epoch(ddp_model, train_loader, criterion, optimizer, scaler, rank, epoch, stage, config, max_stage=stage)
        val_losses = validate(ddp_model.module if dist.is_initialized() else ddp_model, val_loader, criterion, rank, stage, max_stage=stage)
        scheduler.step()
        if rank == 0:
            print(f"\nEpoch {epoch}/{num_epochs} Summary:\nTrain Loss: {train_losses['total']:.4f}\n"
                  f"Val Loss: {val_losses['total']:.4f} | PSNR: {val_losses['psnr']:.2f} dB | SSIM: {val_losses['ssim']:.4f}")
            state = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                     "scheduler_state_dict": scheduler.state_dict(), "val_loss": val_losses["total"], "config": config}
            if val_losses["total"] < best_val_loss:
                best_val_loss = val_losses["total"]
                path = Path(checkpoint_dir) / f"stage{stage}_best.pth"
                torch.save(dict(state, val_psnr=val_losses["psnr"], val_ssim=val_losses["ssim"]), path)
                print(f"✓ Saved best checkpoint: {path}")
            if epoch % config["checkpoints"]["save_every"] == 0:
                path = Path(checkpoint_dir) / f"stage{stage}_epoch{epoch}.pth"
                torch.save(state, path)
                print(f"✓ Saved checkpoint: {path}")
    if dist.is_initialized():
        dist.barrier()


def main_worker(rank, world_size, config, stages=(1, 2, 3), synthetic=False):
    checkpoint_dir = Path(config["checkpoints"]["save_dir"])
    checkpoint_dir.mkdir(parents=True, exist_ok=True)
    for stage in stages:
        if rank == 0:
            print(f"\n{'#' * 60}\n# STAGE {stage} TRAINING\n{'#' * 60}\n")
        train_stage(rank, world_size, config, stage, checkpoint_dir, synthetic)
    cleanup_ddp()


def main():
    from hvc.dist_env import ensure_rccl_env
    ensure_rccl_env()          # before any HIP call of this process and of the ranks it spawns
    ap = argparse.ArgumentParser(description="Progressive cascade training, data-parallel on MI355X")
    ap.add_argument("--config", type=str, default=str(Path(__file__).parent / "config_progressive.json"))
    ap.add_argument("--stages", type=int, nargs="+", default=[1, 2, 3], choices=[1, 2, 3])
    ap.add_argument("--synthetic", action="store_true", help="train on the seeded synthetic phantoms")
    args = ap.parse_args()
    if not os.path.exists(args.config):
        print(f"Error: Config file not found: {args.config}\nPlease create config_progressive.json first!")
        return
    with open(args.config) as f:
        config = json.load(f)
    print("=" * 60 + "\nProgressive Multi-Scale CT Reconstruction Training\n" + "=" * 60)
    print(f"Start time: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\nConfig: {args.config}\n" + "=" * 60)
    if "LOCAL_RANK" in os.environ:          # launched by torchrun: one process per GPU already exists
        main_worker(int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"]), config, tuple(args.stages), args.synthetic)
        return
    world_size = torch.cuda.device_count()
    print(f"Using {world_size} GPUs")
    if world_size < 1:
        print("Error: No GPUs available!")
        return
    if world_size > 1:
        mp.spawn(main_worker, args=(world_size, config, tuple(args.stages), args.synthetic), nprocs=world_size, join=True)
    else:
        main_worker(0, 1, config, tuple(args.stages), args.synthetic)
    print("\n" + "=" * 60 + f"\nTraining Complete!\nEnd time: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n" + "=" * 60)


if __name__ == "__main__":
    main()
