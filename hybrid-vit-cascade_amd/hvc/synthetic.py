"""Seeded synthetic X-ray / CT pairs with the reference's dataset contract
(utils/dataset.py:285-349: {'drr_stacked': (2,1,S,S), 'ct_volume': (1,D,H,W)}, both in [-1,1]).

Recipe: CT = sum of 12 random anisotropic Gaussian ellipsoids + uniform noise
(+-0.02), min-max normalised to [-1,1]; X-rays = mean projections over D (AP) and W (lateral),
bilinear-resized (align_corners=False) to S x S and affinely mapped to [-1,1].
Pure torch on whatever device is asked for; used by tests, bench.py and the trainers' --synthetic mode.
"""
import torch
import torch.nn.functional as F


def sample(index, volume_size=(64, 64, 64), xray_size=512, device="cpu"):
    g = torch.Generator().manual_seed(1234 + int(index))
    D, H, W = volume_size
    zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, D), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    vol = torch.zeros(D, H, W)
    for _ in range(12):
        c = torch.rand(3, generator=g) * 1.4 - 0.7
        s = torch.rand(3, generator=g) * 0.35 + 0.08
        amp = torch.rand(1, generator=g).item() * 0.8 + 0.2
        vol += amp * torch.exp(-(((zz - c[0]) / s[0]) ** 2 + ((yy - c[1]) / s[1]) ** 2 + ((xx - c[2]) / s[2]) ** 2))
    vol += (torch.rand(D, H, W, generator=g) * 2 - 1) * 0.02
    vol = (vol - vol.min()) / (vol.max() - vol.min()) * 2 - 1
    ct = vol[None]                                          # (1,D,H,W)
    ap = ct.mean(dim=1, keepdim=True)                       # (1,1,H,W)
    lat = ct.mean(dim=3)[None]                              # (1,1,D,H)
    views = []
    for v in (ap, lat):
        v = F.interpolate(v, size=(xray_size, xray_size), mode="bilinear", align_corners=False)
        v = (v - v.min()) / (v.max() - v.min() + 1e-12) * 2 - 1
        views.append(v[0])
    xr = torch.stack(views, 0)                              # (2,1,S,S)
    return xr.to(device), ct.to(device)


def batch(start, n, volume_size=(64, 64, 64), xray_size=512, device="cpu"):
    xs, cs = zip(*(sample(start + i, volume_size, xray_size) for i in range(n)))
    return torch.stack(xs).to(device), torch.stack(cs).to(device)


class SyntheticPatientDataset(torch.utils.data.Dataset):
    """Stand-in for utils.dataset.PatientDRRDataset with the same item dict."""

    def __init__(self, n=64, volume_size=(64, 64, 64), xray_size=512):
        self.n, self.volume_size, self.xray_size = n, tuple(volume_size), xray_size

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        xr, ct = sample(i, self.volume_size, self.xray_size)
        return {"drr_stacked": xr, "ct_volume": ct}
