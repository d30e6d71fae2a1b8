"""Patient DRR / CT dataset with the contract of the reference's utils/dataset.py.

Real data (reference utils/dataset.py:34-349): `data_path` holds one folder per patient, named by the patient id `pid`:
    pid/pid_pa_drr.*  | pid_pa.*  | pid_frontal.*      frontal (PA) DRR      -> view 0
    pid/pid_lat_drr.* | pid_lat.* | pid_lateral.*      lateral DRR           -> view 1
    pid/pid.nii.gz    | pid.nii   | pid.npy            CT volume in Hounsfield units
DRRs: bilinear resize (align_corners=False) to target_xray_size, divided by 255 when the image exceeds 1, mapped
affinely onto normalize_range.  CT: trilinear resize (align_corners=False) to target_volume_size, clamped to the soft
tissue window [-200, 200] HU, (v + 200) / 400, mapped onto normalize_range (:193-229).  Items are dicts with
'drr_frontal', 'drr_lateral' (1,S,S), 'drr_stacked' (2,1,S,S), 'ct_volume' (1,D,H,W), 'patient_id', 'aligned' (:285-349).

A `data_path` that is given but holds no valid patient folder raises ValueError, as in the reference (:78-79); a NIfTI
volume without nibabel, or a PNG without PIL, raises ImportError when that file is read.  Synthetic phantoms
(hvc.synthetic, same dict contract) are served ONLY on explicit request: data_path=None (the trainers' --synthetic
flag, the benchmark, the tests) -- never as a silent substitute for unusable real data.
"""
import fnmatch
import os

import numpy as np
import torch
import torch.nn.functional as F

from hvc import synthetic

HU_WINDOW = (-200.0, 200.0)                     # reference utils/dataset.py:218-223
ALIGNMENT_THRESHOLD = 0.5                       # reference :271

_FRONTAL = ("{pid}_pa_drr.*", "{pid}_pa.*", "{pid}_frontal.*")
_LATERAL = ("{pid}_lat_drr.*", "{pid}_lat.*", "{pid}_lateral.*")
_CT_EXT = (".nii.gz", ".nii", ".npy")


def _first_match(folder, patterns):
    """First file of `folder` (directory order, as pathlib's glob yields it) matching the first pattern that matches any."""
    pid = os.path.basename(folder.rstrip(os.sep))
    names = os.listdir(folder)
    for pat in patterns:
        hits = fnmatch.filter(names, pat.format(pid=pid))
        if hits:
            return os.path.join(folder, hits[0])
    return None


class PatientDRRDataset(torch.utils.data.Dataset):
    """Constructor as the reference (utils/dataset.py:34-45).  The reference's progressive trainer calls it with
    `root_dir=..., split=..., train_split=..., val_split=...` (train_progressive_4gpu.py:267-281), which its own class
    rejects; here `root_dir` is an alias of `data_path` and `split` selects the contiguous train / val / test range."""

    def __init__(self, data_path=None, target_xray_size=512, target_volume_size=(256, 256, 256), normalize_range=(-1, 1),
                 validate_alignment=True, augmentation=False, cache_in_memory=False, flip_drrs_vertical=False,
                 max_patients=None, root_dir=None, split=None, train_split=0.8, val_split=0.1):
        data_path = data_path if data_path is not None else root_dir
        self.data_path = data_path
        self.target_xray_size = target_xray_size
        self.target_volume_size = tuple(target_volume_size)
        self.normalize_range = tuple(normalize_range)
        self.validate_alignment = validate_alignment
        self.augmentation = augmentation
        self.flip_drrs_vertical = flip_drrs_vertical
        self.cache = {} if cache_in_memory else None
        self.alignment_stats = {"total": 0, "passed": 0, "failed": 0, "avg_error": 0.0}
        self.synthetic = data_path is None
        if self.synthetic:
            self.patient_folders = []
            self.index = list(range(int(max_patients) if max_patients else 64))
        else:
            self.patient_folders = self._scan(str(data_path), max_patients)
            if not self.patient_folders:
                raise ValueError(f"No valid patient folders found in {data_path}")
            print(f"Found {len(self.patient_folders)} valid patient datasets in {data_path}")
            self.index = list(range(len(self.patient_folders)))
        if split is not None:
            if split not in ("train", "val", "test"):
                raise ValueError(f"split must be 'train', 'val' or 'test', got {split!r}")
            n = len(self.index)
            a, b = int(train_split * n), int(train_split * n) + int(val_split * n)
            self.index = {"train": self.index[:a], "val": self.index[a:b], "test": self.index[b:]}[split]

    # ---- folder discovery (reference :67-79, :98-128) ------------------------------------------------
    @classmethod
    def _scan(cls, data_path, max_patients):
        found = []
        if os.path.isdir(data_path):
            for name in sorted(os.listdir(data_path)):
                folder = os.path.join(data_path, name)
                if not os.path.isdir(folder) or name.startswith("."):
                    continue
                if cls._validate_patient_folder(folder):
                    found.append(folder)
                    if max_patients is not None and len(found) >= max_patients:
                        break
        return found

    @classmethod
    def _validate_patient_folder(cls, folder):
        missing = [what for what, path in (("frontal/PA image", cls._find_file(folder, "drr_frontal")),
                                           ("lateral image", cls._find_file(folder, "drr_lateral")),
                                           ("CT volume", cls._find_file(folder, "ct_volume"))) if path is None]
        for what in missing:
            print(f"Warning: Missing {what} in {os.path.basename(folder)}")
        return not missing

    @staticmethod
    def _find_file(folder, file_type):
        """'drr_frontal' / 'drr_lateral' / 'ct_volume' -> path or None (reference :133-157)."""
        if file_type == "drr_frontal":
            return _first_match(folder, _FRONTAL)
        if file_type == "drr_lateral":
            return _first_match(folder, _LATERAL)
        if file_type == "ct_volume":
            pid = os.path.basename(folder.rstrip(os.sep))
            for ext in _CT_EXT:
                path = os.path.join(folder, pid + ext)
                if os.path.exists(path):
                    return path
        return None

    # ---- file decoding (reference :159-229) ------------------------------------------------------------
    def _to_range(self, unit):
        lo, hi = self.normalize_range
        return unit * (hi - lo) + lo

    def _load_image(self, filepath, target_size):
        if filepath.endswith(".npy"):
            img = np.load(filepath).astype(np.float32)
            if img.ndim == 2:
                img = img[None]
        else:
            try:
                from PIL import Image
            except ImportError as e:
                raise ImportError(f"reading {filepath} needs Pillow (PIL)") from e
            img = np.asarray(Image.open(filepath).convert("L"), dtype=np.float32)[None]
        t = torch.from_numpy(img)                                    # (1, H, W)
        if t.shape[1] != target_size or t.shape[2] != target_size:
            t = F.interpolate(t[None], size=(target_size, target_size), mode="bilinear", align_corners=False)[0]
        if t.max() > 1.0:                                            # 8-bit grey levels -> [0, 1]
            t = t / 255.0
        return self._to_range(t)

    def _load_volume(self, filepath, target_size):
        if filepath.endswith(".npy"):
            vol = np.load(filepath).astype(np.float32)
        else:
            try:
                import nibabel as nib
            except ImportError as e:
                raise ImportError(f"reading {filepath} needs nibabel (NIfTI); .npy volumes load without it") from e
            vol = nib.load(filepath).get_fdata().astype(np.float32)
        if vol.ndim == 3:
            vol = vol[None]
        t = torch.from_numpy(vol)                                    # (1, D, H, W), Hounsfield units
        if tuple(t.shape[1:]) != tuple(target_size):
            t = F.interpolate(t[None], size=tuple(target_size), mode="trilinear", align_corners=False)[0]
        lo, hi = HU_WINDOW
        return self._to_range((torch.clamp(t, lo, hi) - lo) / (hi - lo))

    def _validate_drr_ct_alignment(self, drr_frontal, drr_lateral, ct_volume, patient_id):
        """Mean squared distance between the DRRs and max-intensity projections of the CT (reference :231-283)."""
        size = (self.target_xray_size, self.target_xray_size)
        synth_frontal = F.interpolate(ct_volume.max(dim=1)[0][None], size=size, mode="bilinear", align_corners=False)[0]
        synth_lateral = F.interpolate(ct_volume.max(dim=3)[0][None], size=size, mode="bilinear", align_corners=False)[0]
        err = (F.mse_loss(drr_frontal, synth_frontal).item() + F.mse_loss(drr_lateral, synth_lateral).item()) / 2
        return err < ALIGNMENT_THRESHOLD, err

    def _apply_augmentation(self, drr_stacked, ct_volume):
        """Random left-right flip and intensity scale in [0.9, 1.1], clamped to normalize_range (reference :351-374)."""
        if torch.rand(1).item() > 0.5:
            drr_stacked, ct_volume = torch.flip(drr_stacked, [-1]), torch.flip(ct_volume, [-1])
        if torch.rand(1).item() > 0.5:
            scale = 0.9 + 0.2 * torch.rand(1).item()
            drr_stacked, ct_volume = drr_stacked * scale, ct_volume * scale
        lo, hi = self.normalize_range
        return torch.clamp(drr_stacked, lo, hi), torch.clamp(ct_volume, lo, hi)

    def get_alignment_report(self):
        s = self.alignment_stats
        n = s["total"]
        return {"total_validated": n, "passed": s["passed"], "failed": s["failed"],
                "pass_rate": s["passed"] / n if n else 0.0, "average_error": s["avg_error"] / n if n else 0.0}

    # ---- items (reference :285-349) ----------------------------------------------------------------------
    def __len__(self):
        return len(self.index)

    def __getitem__(self, idx):
        idx = self.index[idx]
        if self.synthetic:
            xr, ct = synthetic.sample(idx, self.target_volume_size, self.target_xray_size)
            return {"drr_frontal": xr[0], "drr_lateral": xr[1], "drr_stacked": xr, "ct_volume": ct,
                    "patient_id": f"synthetic_{idx:04d}", "aligned": True}
        if self.cache is not None and idx in self.cache:
            return self.cache[idx]
        folder = self.patient_folders[idx]
        pid = os.path.basename(folder)
        frontal = self._load_image(self._find_file(folder, "drr_frontal"), self.target_xray_size)
        lateral = self._load_image(self._find_file(folder, "drr_lateral"), self.target_xray_size)
        ct = self._load_volume(self._find_file(folder, "ct_volume"), self.target_volume_size)
        if self.flip_drrs_vertical:
            frontal, lateral = torch.flip(frontal, dims=[-2]), torch.flip(lateral, dims=[-2])
        aligned = True
        if self.validate_alignment:
            aligned, err = self._validate_drr_ct_alignment(frontal, lateral, ct, pid)
            s = self.alignment_stats
            s["total"] += 1
            s["passed" if aligned else "failed"] += 1
            s["avg_error"] += err
        stacked = torch.stack([frontal, lateral], dim=0)             # view 0 = frontal / PA, view 1 = lateral
        if self.augmentation:
            stacked, ct = self._apply_augmentation(stacked, ct)
        item = {"drr_frontal": frontal, "drr_lateral": lateral, "drr_stacked": stacked, "ct_volume": ct,
                "patient_id": pid, "aligned": aligned}
        if self.cache is not None:
            self.cache[idx] = item
        return item


def create_train_val_datasets(data_path=None, train_split=0.8, val_split=0.1, **dataset_kwargs):
    """(train, val, test) random split with the reference's fixed generator seed 42 (reference :393-428)."""
    full = PatientDRRDataset(data_path, **dataset_kwargs)
    n = len(full)
    n_train, n_val = int(train_split * n), int(val_split * n)
    parts = torch.utils.data.random_split(full, [n_train, n_val, n - n_train - n_val], generator=torch.Generator().manual_seed(42))
    print(f"\nDataset splits:\n  Train: {len(parts[0])} samples\n  Val:   {len(parts[1])} samples\n  Test:  {len(parts[2])} samples")
    return tuple(parts)
