"""CPU-side checks (no GPU compute): the C-ABI library loads and exports exactly what include/hvc_hip.h declares,
and the host-side mirrors keep the reference's construction contract (state_dict keys / shapes / initial weights,
token-grid geometry, dataset dict contract)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hvc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hvc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hvc import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libhvc_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hvc_hip.h but not exported"
    # the ctypes table binds every declared entry point, and nothing that is not declared
    assert sorted(_lib.SIGNATURES) == declared
    h = _lib.load()
    assert h.hvc_abi_version() == _lib.ABI_VERSION
    # pure host-side queries work without a GPU
    assert h.hvc_gemm_workspace(768, 256, 16384) > 0 and h.hvc_gemm_workspace(16384, 768, 256) == 0
    assert h.hvc_layernorm_bwd_workspace(8192, 256, 4096) > 0
    assert h.hvc_ssim_l1_workspace(2, 16, 16, 16) >= 10 * 2 * 16 ** 3


def test_bad_arguments_are_rejected_before_any_launch():
    from hvc import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libhvc_hip.so not built")
    h = _lib.load()
    rc = h.hvc_attention_fwd(None, None, None, None, None, 1, 1, 8, 8, 64, *([0] * 12), 1.0, 0.0, 0, 0, None)
    assert rc == -1 and b"null operand" in h.hvc_last_error()
    rc = h.hvc_layernorm_fwd(1, 1, 1, None, None, 1, 1, 1, 4, 2048, 4, 1e-5, 0, None)
    assert rc == -2 and b"C <= 1024" in h.hvc_last_error()
    rc = h.hvc_drr_fwd(1, 1, 1, 2, 2, 2, 1, 1, 0.3, 1.0, 0.0, 0, 0, None)
    assert rc == -2 and b"axis" in h.hvc_last_error()


def test_strided_conv_dx_parity_classes_cover_every_tap_once():
    """The class-major tap order of the strided input gradient (hvc_conv_dx_class): the host-side C enumeration and the Python
    one that lays out the weights agree, and every kernel tap belongs to exactly one parity class (k3 s2 p1, k7 s2 p3, k5 s3,
    k1 s2 with its empty classes, 2-D layers as depth-1 volumes)."""
    import ctypes
    from hvc import _lib, ops
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libhvc_hip.so not built")
    h = _lib.load()
    for src, kernel, stride, pad in (((8, 8, 8), (3, 3, 3), 2, (1, 1, 1)), ((1, 16, 16), (1, 7, 7), 2, (0, 3, 3)), ((9, 9, 9), (5, 5, 5), 3, (2, 2, 2)),
                                     ((6, 6, 6), (1, 1, 1), 2, (0, 0, 0)), ((1, 12, 12), (1, 3, 3), 2, (0, 1, 1)), ((7, 5, 9), (4, 4, 4), 2, (1, 1, 1))):
        geom = ops.ConvGeometry(1, 8, src, kernel, stride, pad)
        classes = ops.conv_dx_class_taps(geom)
        flat = [t for _, taps in classes for t in taps]
        assert sorted(flat) == list(range(geom.taps)), (kernel, stride, pad)
        col = 0
        for (cd, ch, cw), taps in classes:
            first, n = ctypes.c_int(-1), ctypes.c_int(-1)
            assert h.hvc_conv_dx_class_columns(*kernel, stride, *pad, cd, ch, cw, ctypes.byref(first), ctypes.byref(n)) == 0
            assert (first.value, n.value) == (col, len(taps)), (kernel, stride, (cd, ch, cw))
            col += len(taps)
    assert h.hvc_conv_dx_class_columns(3, 3, 3, 2, 1, 1, 1, 2, 0, 0, None, None) != 0 and b"bad geometry" in h.hvc_last_error()


def test_cpu_tensors_raise_instead_of_falling_back():
    from hvc import ops
    from models.hybrid_vit_backbone import HybridViT3D
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(torch.randn(4, 8), torch.ones(8), torch.zeros(8))
    m = HybridViT3D(volume_size=(8, 8, 8), voxel_dim=64, depth=1, num_heads=2, context_dim=16, cond_dim=8)
    with pytest.raises(RuntimeError, match="HIP device only"):
        m(torch.randn(1, 1, 8, 8, 8), torch.randn(1, 4, 16), torch.randn(1, 8))


def test_direct_model_reproduces_reference_construction(golden):
    """Seeded construction gives the reference's state_dict keys, shapes and parameter count, and the
    oracle run on those initial weights reproduces the reference's full-size known answers."""
    from direct_regression.model_direct import DirectCTRegression
    from oracle import hvc_oracle as O
    z = golden("direct_kat64").z
    torch.manual_seed(0)
    m = DirectCTRegression(volume_size=(64, 64, 64)).eval()
    assert list(m.state_dict().keys()) == list(z["keys"])
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == list(z["shapes"])
    assert sum(p.numel() for p in m.parameters()) == int(z["nparams"]) == 15300481
    x = torch.randn(1, 2, 1, 512, 512, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = O.direct_ct_regression(x, {k: v for k, v in m.state_dict().items()})
    assert abs(y.mean().item() - float(z["mean"])) < 1e-6 and abs(y.std().item() - float(z["std"])) < 1e-6
    assert abs(y.abs().sum().item() - float(z["abssum"])) < 0.5
    for idx, val in zip(z["idx"], z["vals"]):
        assert abs(y[tuple(idx)].item() - float(val)) < 1e-5
    assert abs(float(z["mean"]) - 0.1782742) < 1e-6        # the value SURVEY.md measured on the reference


def test_token_grid_geometry_is_bit_exact():
    from models.hybrid_vit_backbone import HybridViT3D
    kw = dict(voxel_dim=64, depth=1, num_heads=2, context_dim=16, cond_dim=8)
    assert HybridViT3D(volume_size=(64, 64, 64), **kw).downsampled_size == (16, 16, 16)
    assert HybridViT3D(volume_size=(256, 256, 256), in_channels=32, **kw).downsampled_size == (32, 32, 32)
    assert HybridViT3D(volume_size=(64, 32, 32), **kw).downsampled_size == (16, 8, 8)
    assert HybridViT3D(volume_size=(8, 8, 8), **kw).downsampled_size == (8, 8, 8)
    # 128^3: the reference builds a 25^3 pos_embed for a 32^3 stem output and raises; the build follows the stem (A2-fix)
    m = HybridViT3D(volume_size=(128, 128, 128), **kw)
    assert m.downsampled_size == (32, 32, 32) and m.pos_embed.shape == (1, 32768, 64)
    assert HybridViT3D(volume_size=(128, 128, 128), token_grid=16, **kw).downsampled_size == (16, 16, 16)
    # token order n = (d*H' + h)*W' + w is what reshape(B, 1, D', H', W') of the head inverts
    Dd, Hd, Wd = 4, 3, 2
    n = torch.arange(Dd * Hd * Wd).reshape(Dd, Hd, Wd)
    assert n[2, 1, 1].item() == (2 * Hd + 1) * Wd + 1


def test_cascade_parameter_budget_matches_reference():
    from direct_regression.progressive_cascade import ProgressiveCascadeModel
    m = ProgressiveCascadeModel()
    count = lambda mod: sum(p.numel() for p in mod.parameters())
    assert count(m.xray_encoder) == 8933504 and count(m.stage1) == 22382977 and count(m.stage3) == 31669252
    # stage 2 differs from the reference's 21,708,034 only by the A2-fix pos_embed (32768 - 15625 tokens x 256)
    assert count(m.stage2) == 21708034 + (32768 - 15625) * 256


def test_conv_geometry_and_weight_layout():
    from hvc import ops
    g = ops.ConvGeometry(2, 64, (32, 32, 32), (3, 3, 3), 2, (1, 1, 1))
    assert g.out == (16, 16, 16) and g.Kp == 27 * 64 and g.M == 2 * 16 ** 3
    g1 = ops.ConvGeometry(4, 1, (1, 512, 512), (1, 7, 7), 2, (0, 3, 3))
    assert g1.out == (1, 256, 256) and g1.Kp == 56          # 49 taps padded to a multiple of 8


def test_synthetic_dataset_contract():
    from utils.dataset import PatientDRRDataset
    ds = PatientDRRDataset(data_path=None, target_xray_size=64, target_volume_size=(16, 16, 16), max_patients=3)
    assert len(ds) == 3
    item = ds[1]
    assert item["drr_stacked"].shape == (2, 1, 64, 64) and item["ct_volume"].shape == (1, 16, 16, 16)
    for k in ("drr_stacked", "ct_volume"):
        assert -1.0001 <= item[k].min().item() and item[k].max().item() <= 1.0001
    assert torch.equal(ds[1]["ct_volume"], item["ct_volume"])      # seeded: reproducible


def test_dataset_split_ranges_of_the_progressive_trainer():
    """train_progressive_4gpu.py:267-281 constructs the dataset with root_dir / split / train_split / val_split."""
    from utils.dataset import PatientDRRDataset
    kw = dict(root_dir=None, max_patients=10, train_split=0.8, val_split=0.1, target_xray_size=32, target_volume_size=(8, 8, 8))
    parts = {s: PatientDRRDataset(split=s, **kw) for s in ("train", "val", "test")}
    assert [len(parts[s]) for s in ("train", "val", "test")] == [8, 1, 1]
    ids = [parts[s][i]["patient_id"] for s in ("train", "val", "test") for i in range(len(parts[s]))]
    assert ids == [f"synthetic_{i:04d}" for i in range(10)]          # disjoint, ordered, complete
    with pytest.raises(ValueError):
        PatientDRRDataset(split="holdout", **kw)
    import inspect
    sig = inspect.signature(PatientDRRDataset.__init__)
    assert sig.parameters["target_volume_size"].default == (256, 256, 256)   # reference utils/dataset.py:38


def test_progressive_trainer_module_surface():
    """Counterpart of direct_regression/progressive_cascade/train_progressive_4gpu.py: same entry points and config keys."""
    import importlib, json, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, os.path.join(here, "hybrid-vit-cascade_amd", "direct_regression"))
    mod = importlib.import_module("progressive_cascade.train_progressive_4gpu")
    for name in ("setup_ddp", "cleanup_ddp", "resize_ct_volume", "train_epoch", "validate", "train_stage", "main_worker", "main"):
        assert callable(getattr(mod, name)), name
    cfg = json.load(open(os.path.join(here, "hybrid-vit-cascade_amd", "direct_regression", "progressive_cascade", "config_progressive.json")))
    assert cfg["training"]["stage2"] == {"num_epochs": 30, "batch_size": 2, "learning_rate": 5e-05, "target_resolution": [128, 128, 128]}
    assert cfg["loss"]["stage3"]["drr"] == 0.3 and cfg["model"]["voxel_dim"] == 256
    assert mod.STAGE_SIZES == {1: (64,) * 3, 2: (128,) * 3, 3: (256,) * 3}


def test_bench_rejects_world_size_mismatch_before_touching_the_gpu():
    """`bench.py --gpus N` inside a launcher whose WORLD_SIZE disagrees must exit non-zero (it used to be
    accepted silently); the check runs before any GPU call, so it is testable here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_spawn_command_is_a_fresh_torchrun_child(monkeypatch):
    """Plain `python bench.py --gpus N` starts N ranks as a torch.distributed.run CHILD process (never an exec of the
    current one) on 127.0.0.1 and hands back its exit code."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("hvc_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _write_patient(root, pid, rng, ct_shape=(6, 5, 4), frontal_name=None, lateral_name=None, ct_ext=".npy"):
    import numpy as np
    d = root / pid
    d.mkdir()
    ct = rng.uniform(-1000, 1500, size=ct_shape).astype(np.float32)              # Hounsfield units, beyond the window both ways
    frontal = rng.uniform(0, 255, size=(8, 8)).astype(np.float32)                # 8-bit grey levels
    lateral = rng.uniform(0, 1, size=(8, 8)).astype(np.float32) * 0.9            # already in [0, 1]
    np.save(d / (frontal_name or f"{pid}_pa_drr.npy"), frontal)
    np.save(d / (lateral_name or f"{pid}_lat_drr.npy"), lateral)
    np.save(d / f"{pid}{ct_ext}", ct)
    return ct, frontal, lateral


def test_real_data_loader_follows_the_reference_contract(tmp_path):
    """utils/dataset.py of the reference (:98-229, :285-349) on a fake patient tree of .npy files: view order (frontal / PA
    is view 0 although `*_lat_*` sorts first), the CT file is `pid.npy` and is not mistaken for a view, HU window
    [-200, 200] -> [-1, 1], DRRs / 255 only when they exceed 1, resize modes, vertical flip, item keys."""
    import numpy as np
    import torch.nn.functional as F
    from utils.dataset import PatientDRRDataset
    rng = np.random.default_rng(7)
    ct, frontal, lateral = _write_patient(tmp_path, "p001", rng)
    _write_patient(tmp_path, "p002", rng, frontal_name="p002_frontal.npy", lateral_name="p002_lateral.npy")
    (tmp_path / "p003").mkdir()                                                   # no files: skipped with warnings
    (tmp_path / ".hidden").mkdir()
    ds = PatientDRRDataset(str(tmp_path), target_xray_size=8, target_volume_size=(6, 5, 4))
    assert len(ds) == 2 and [os.path.basename(p) for p in ds.patient_folders] == ["p001", "p002"]
    item = ds[0]
    assert set(item) == {"drr_frontal", "drr_lateral", "drr_stacked", "ct_volume", "patient_id", "aligned"}
    assert item["patient_id"] == "p001" and item["drr_stacked"].shape == (2, 1, 8, 8) and item["ct_volume"].shape == (1, 6, 5, 4)
    exp_ct = (np.clip(ct, -200, 200) + 200) / 400 * 2 - 1
    assert np.allclose(item["ct_volume"][0].numpy(), exp_ct, atol=1e-6)
    assert np.allclose(item["drr_stacked"][0, 0].numpy(), frontal / 255 * 2 - 1, atol=1e-6)       # view 0 = frontal, / 255
    assert np.allclose(item["drr_stacked"][1, 0].numpy(), lateral * 2 - 1, atol=1e-6)             # view 1 = lateral, max <= 1: no / 255
    assert torch.equal(item["drr_frontal"], item["drr_stacked"][0]) and torch.equal(item["drr_lateral"], item["drr_stacked"][1])
    assert ds[1]["patient_id"] == "p002"                                         # *_frontal / *_lateral naming
    # resize paths: bilinear / trilinear, align_corners=False, applied BEFORE the window / scaling
    big = PatientDRRDataset(str(tmp_path), target_xray_size=16, target_volume_size=(12, 10, 8), normalize_range=(0, 1), validate_alignment=False)[0]
    exp = F.interpolate(torch.from_numpy(ct)[None, None], size=(12, 10, 8), mode="trilinear", align_corners=False)[0]
    assert torch.allclose(big["ct_volume"], (exp.clamp(-200, 200) + 200) / 400, atol=1e-6)
    expf = F.interpolate(torch.from_numpy(frontal)[None, None], size=(16, 16), mode="bilinear", align_corners=False)[0] / 255
    assert torch.allclose(big["drr_frontal"], expf, atol=1e-6)
    flipped = PatientDRRDataset(str(tmp_path), target_xray_size=8, target_volume_size=(6, 5, 4), flip_drrs_vertical=True)[0]
    assert torch.equal(flipped["drr_frontal"], torch.flip(item["drr_frontal"], dims=[-2]))
    rep = ds.get_alignment_report()
    assert rep["total_validated"] == 2 and rep["passed"] + rep["failed"] == 2
    # max_patients stops the scan
    assert len(PatientDRRDataset(str(tmp_path), target_xray_size=8, target_volume_size=(6, 5, 4), max_patients=1)) == 1


def test_unusable_data_path_raises_instead_of_training_on_phantoms(tmp_path):
    """A data_path that is given but unusable is an error, as in the reference (utils/dataset.py:78-79); synthetic phantoms
    are served only for data_path=None."""
    from utils.dataset import PatientDRRDataset
    with pytest.raises(ValueError, match="No valid patient folders"):
        PatientDRRDataset(str(tmp_path / "does_not_exist"))
    (tmp_path / "empty_patient").mkdir()
    with pytest.raises(ValueError, match="No valid patient folders"):
        PatientDRRDataset(str(tmp_path))
    assert PatientDRRDataset(None, max_patients=2, target_xray_size=32, target_volume_size=(8, 8, 8)).synthetic


def test_nifti_volume_without_nibabel_raises_at_read_time(tmp_path):
    import numpy as np
    from utils.dataset import PatientDRRDataset
    rng = np.random.default_rng(8)
    _write_patient(tmp_path, "q1", rng)
    os.rename(tmp_path / "q1" / "q1.npy", tmp_path / "q1" / "q1.nii.gz")          # a NIfTI name: found, but unreadable without nibabel
    ds = PatientDRRDataset(str(tmp_path), target_xray_size=8, target_volume_size=(6, 5, 4))
    try:
        import nibabel  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="nibabel"):
            ds[0]


def test_train_val_test_split_uses_the_reference_seed():
    from utils.dataset import create_train_val_datasets
    tr, va, te = create_train_val_datasets(None, max_patients=20, target_xray_size=32, target_volume_size=(8, 8, 8))
    assert (len(tr), len(va), len(te)) == (16, 2, 2)
    ref = torch.utils.data.random_split(range(20), [16, 2, 2], generator=torch.Generator().manual_seed(42))
    assert list(tr.indices) == list(ref[0].indices) and sorted(tr.indices + va.indices + te.indices) == list(range(20))


def test_checkpoint_resume_round_trip(tmp_path):
    """save_checkpoint / load_checkpoint of the direct trainer (reference train_direct_4gpu.py:177-189, :273-298): the dict
    holds the reference's keys, a resumed model + optimizer + scheduler continue bit for bit, a missing file raises."""
    from direct_regression import train_direct_4gpu as T

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
        o = torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.01)
        return m, o, torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=10, eta_min=1e-6)

    def step(m, o, s, x):
        o.zero_grad()
        m(x).pow(2).mean().backward()
        o.step()
        s.step()
    x = torch.randn(4, 6)
    m, o, s = make()
    for _ in range(3):
        step(m, o, s, x)
    path = tmp_path / "checkpoint_epoch_3.pt"
    T.save_checkpoint(path, 3, m, o, s, 21.5, 22.0, {"training": {"num_epochs": 10}})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "val_psnr", "best_psnr", "config"}
    m2, o2, s2 = make()
    start, best = T.load_checkpoint(str(path), m2, o2, s2, "cpu")
    assert (start, best) == (4, 22.0) and s2.get_last_lr() == s.get_last_lr()
    step(m, o, s, x)
    step(m2, o2, s2, x)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    with pytest.raises(FileNotFoundError):
        T.load_checkpoint(str(tmp_path / "nope.pt"), m2, o2, s2, "cpu")


def test_every_optimizer_step_invalidates_the_weight_cast_cache():
    """AdamW(fused=True) updates parameters WITHOUT bumping their version counters, so the cached bf16 weight copies are
    keyed by an epoch that every optimizer step advances (hvc/functional.py); without it training silently runs on the
    pre-step weights."""
    from hvc import functional as HF
    p = torch.nn.Parameter(torch.randn(4, 4))
    for kw in (dict(fused=True), dict(foreach=True), dict()):
        opt = torch.optim.AdamW([p], lr=1e-3, **kw)
        p.grad = torch.randn(4, 4)
        key0 = HF._cache_key(p, torch.bfloat16)
        opt.step()
        assert HF._cache_key(p, torch.bfloat16) != key0, kw


def test_checkpoint_policy_accepts_json_booleans_and_decides_once():
    """"gradient_checkpointing": true / false in the JSON config map to 'on' / 'off'; anything else still raises."""
    from hvc import functional as HF
    try:
        HF.set_checkpoint_policy(True)
        assert HF.CHECKPOINT_POLICY == "on" and HF.use_checkpoint(True, 1, "cpu") and not HF.use_checkpoint(False, 1, "cpu")
        HF.set_checkpoint_policy(False)
        assert HF.CHECKPOINT_POLICY == "off" and not HF.use_checkpoint(True, 1 << 50, "cpu")
        with pytest.raises(ValueError):
            HF.set_checkpoint_policy("sometimes")
    finally:
        HF.set_checkpoint_policy("auto")


def test_options_are_declared_in_the_header_and_bound():
    """hvc_set_option / hvc_get_option replace per-launch getenv() reads: declared in include/hvc_hip.h, exported by the
    library, unknown names refused, values round-trip, the environment only seeds the initial value."""
    import ctypes as C
    from hvc import _lib, ops
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "hvc_hip.h")).read()
    for name in ("HVC_ATTN_FWD_ROWS", "HVC_ATTN_FWD_WAVES", "HVC_ATTN_BWD_WAVES", "HVC_ATTN_PIPE", "HVC_GEMM_PERSISTENT",
                 "HVC_GEMM_STAGGER", "HVC_GEMM_HALF_TILE", "HVC_FP8_MX", "HVC_CONV_FORCE_ADDR64", "HVC_ATTN_EXTRA_LDS", "HVC_LOSS_FUSED"):
        assert name in hdr, name
        old = ops.get_option(name)
        with ops.options(**{name: 4}):
            assert ops.get_option(name) == 4
        assert ops.get_option(name) == old
    assert lib.hvc_set_option(b"HVC_NO_SUCH_SWITCH", 1) != 0
    assert lib.hvc_set_option(b"HVC_GEMM_STAGGER", -3) != 0          # a negative stagger used to disable the persistent form silently
    v = C.c_int(-1)
    assert lib.hvc_get_option(b"HVC_GEMM_PERSISTENT", C.byref(v)) == 0 and v.value == 1
    csrc = os.path.join(ROOT, "hybrid-vit-cascade_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip") and f != "capi.hip":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f}: launch paths must read hvc::option(), not the environment"
