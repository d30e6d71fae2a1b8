"""Progressive cascade (64^3 -> 128^3 -> 256^3): same exports as the reference package."""
from .model_progressive import (MultiScaleXrayEncoder, ProgressiveCascadeModel, Stage1Base64, Stage2Refiner128,
                                Stage3Refiner256)
from .loss_multiscale import (DRRReprojectionLoss, FrequencyLoss, MultiScaleLoss, SSIMLoss, Stage1Loss, Stage2Loss,
                              Stage3Loss, TotalVariationLoss, compute_psnr, compute_ssim_metric)

__all__ = ["MultiScaleXrayEncoder", "ProgressiveCascadeModel", "Stage1Base64", "Stage2Refiner128", "Stage3Refiner256",
           "DRRReprojectionLoss", "FrequencyLoss", "MultiScaleLoss", "SSIMLoss", "Stage1Loss", "Stage2Loss", "Stage3Loss",
           "TotalVariationLoss", "compute_psnr", "compute_ssim_metric"]
