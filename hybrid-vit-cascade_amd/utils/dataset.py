"""Dataset contract of the reference's utils/dataset.py (items: {'drr_stacked': (2,1,S,S), 'ct_volume': (1,D,H,W)},
both in [-1, 1]; reference :285-349).  The real loader needs nibabel / PIL (absent offline): when they or the data
directory are missing, PatientDRRDataset serves the seeded synthetic phantoms of hvc.synthetic with the same dict
contract, which is what the benchmark and the tests use."""
import os

import torch

from hvc import synthetic


class PatientDRRDataset(torch.utils.data.Dataset):
    """Constructor as the reference (utils/dataset.py:34-45: data_path, target_xray_size=512,
    target_volume_size=(256,256,256), ...).  The reference's progressive trainer calls it with
    `root_dir=..., split=..., train_split=..., val_split=...` (train_progressive_4gpu.py:267-281), which its own class
    rejects; here `root_dir` is an alias of `data_path` and `split` selects the contiguous train / val / test range."""

    def __init__(self, data_path=None, target_xray_size=512, target_volume_size=(256, 256, 256), normalize_range=(-1, 1),
                 validate_alignment=False, augmentation=False, cache_in_memory=False, flip_drrs_vertical=False,
                 max_patients=None, root_dir=None, split=None, train_split=0.8, val_split=0.1, **_unused):
        data_path = data_path if data_path is not None else root_dir
        self.data_path = data_path
        self.target_xray_size = target_xray_size
        self.target_volume_size = tuple(target_volume_size)
        self.synthetic = data_path is None or not os.path.isdir(data_path)
        if not self.synthetic:
            try:
                import nibabel  # noqa: F401
            except ImportError:
                print("[hvc] nibabel is not installed: serving synthetic phantoms with the PatientDRRDataset contract")
                self.synthetic = True
        if self.synthetic:
            total = int(max_patients) if max_patients else 64
            self.index = list(range(total))
        else:
            self.patients = sorted(d for d in os.listdir(data_path) if os.path.isdir(os.path.join(data_path, d)))
            if max_patients:
                self.patients = self.patients[:max_patients]
            self.index = list(range(len(self.patients)))
        if split is not None:
            if split not in ("train", "val", "test"):
                raise ValueError(f"split must be 'train', 'val' or 'test', got {split!r}")
            n = len(self.index)
            a, b = int(train_split * n), int(train_split * n) + int(val_split * n)
            self.index = {"train": self.index[:a], "val": self.index[a:b], "test": self.index[b:]}[split]
        self.n = len(self.index)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        idx = self.index[idx]
        if self.synthetic:
            xr, ct = synthetic.sample(idx, self.target_volume_size, self.target_xray_size)
            return {"drr_stacked": xr, "ct_volume": ct, "patient_id": f"synthetic_{idx:04d}"}
        return self._load_patient(idx)

    def _load_patient(self, idx):
        """NIfTI CT + two DRR images -> tensors in [-1, 1] (reference utils/dataset.py:285-349)."""
        import nibabel as nib
        import numpy as np
        from PIL import Image
        import torch.nn.functional as F
        pdir = os.path.join(self.data_path, self.patients[idx])
        files = sorted(os.listdir(pdir))
        ct_file = next(f for f in files if f.endswith((".nii", ".nii.gz")))
        vol = torch.from_numpy(np.asarray(nib.load(os.path.join(pdir, ct_file)).get_fdata(), dtype=np.float32))
        vol = F.interpolate(vol[None, None], size=self.target_volume_size, mode="trilinear", align_corners=False)[0]
        vol = (vol - vol.min()) / (vol.max() - vol.min() + 1e-8) * 2 - 1
        views = []
        for f in [f for f in files if f.lower().endswith((".png", ".npy"))][:2]:
            path = os.path.join(pdir, f)
            img = np.load(path).astype(np.float32) if f.endswith(".npy") else np.asarray(Image.open(path).convert("L"), dtype=np.float32)
            t = torch.from_numpy(img)[None, None]
            t = F.interpolate(t, size=(self.target_xray_size,) * 2, mode="bilinear", align_corners=False)[0]
            views.append((t - t.min()) / (t.max() - t.min() + 1e-8) * 2 - 1)
        return {"drr_stacked": torch.stack(views, 0), "ct_volume": vol, "patient_id": self.patients[idx]}


def create_train_val_datasets(data_path=None, train_split=0.8, **kwargs):
    full = PatientDRRDataset(data_path=data_path, **kwargs)
    n_train = int(len(full) * train_split)
    return torch.utils.data.Subset(full, range(n_train)), torch.utils.data.Subset(full, range(n_train, len(full)))
