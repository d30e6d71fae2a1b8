import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hybrid-vit-cascade_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
PROBE_LIMIT = 32768
PROBE_N = 512


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def probe_indices(n):
    """Same sampler as tests/golden/make_golden.py."""
    return np.random.default_rng(12345).integers(0, n, size=PROBE_N)


def _report(key, metric, err, tol):
    """HVC_TEST_REPORT=<file>: append every golden comparison's measured error and bound (how much margin a tolerance has)."""
    path = os.environ.get("HVC_TEST_REPORT")
    if path:
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]
        with open(path, "a") as f:
            f.write(f"{test}\t{key}\t{metric}\t{err:.3e}\t{tol:.1e}\n")


class Golden:
    """Read access to one golden .npz (see tests/golden/make_golden.py for the layout)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def t(self, key, dtype=torch.float32):
        return torch.from_numpy(np.asarray(self.z[key])).to(dtype)

    def group(self, prefix, dtype=torch.float32):
        pre = prefix + "/"
        return {k[len(pre):]: torch.from_numpy(np.asarray(self.z[k])).to(dtype if self.z[k].dtype.kind == "f" else torch.int64)
                for k in self.z.files if k.startswith(pre) and "#" not in k}

    def keys(self, prefix):
        pre = prefix + "/"
        return sorted({k[len(pre):].split("#")[0] for k in self.z.files if k.startswith(pre)})

    def is_noise(self, prefix, name, floor=1e-6):
        """True when the stored reference tensor is all rounding noise (a mathematically zero gradient,
        e.g. a conv bias ahead of a normalisation that removes per-channel means)."""
        key = f"{prefix}/{name}" if prefix else name
        a = self.z[key] if key in self.z.files else self.z[key + "#probe"]
        return float(np.abs(a).max()) < floor

    @staticmethod
    def _err(got, ref, metric):
        # floor: exactly-zero gradients (conv bias ahead of BatchNorm) hold only rounding noise
        if metric == "l2":     # relative Frobenius error: the right yardstick for bf16 gradients
            return np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-6 * np.sqrt(ref.size))
        return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)

    def check(self, prefix, name, value, rtol, atol_scale=1.0, metric="max"):
        """Compare `value` with the stored array (full, or probes + sums for compacted entries).
        metric 'max': max|got-ref| / max|ref| ; 'l2': ||got-ref|| / ||ref||."""
        key = f"{prefix}/{name}" if prefix else name
        v = value.detach().double().cpu().numpy()
        tol = rtol * atol_scale
        if key in self.z.files:
            ref = self.z[key].astype(np.float64)
            assert ref.shape == v.shape, f"{key}: shape {v.shape} vs golden {ref.shape}"
            err = self._err(v, ref, metric)
            _report(key, metric, err, tol)
            assert err <= tol, f"{key}: {metric} err {err:.3e} > {tol:.1e}"
            return err
        ref = self.z[key + "#probe"].astype(np.float64)
        stats = self.z[key + "#stats"]
        assert v.size == int(stats[2]), f"{key}: size mismatch"
        got = v.reshape(-1)[probe_indices(v.size)]
        err = self._err(got, ref, metric)
        _report(key + "#probe", metric, err, tol)
        assert err <= tol, f"{key} (probe): {metric} err {err:.3e} > {tol:.1e}"
        # stats[1] = sum |x| bounds the rounding of the plain sum
        assert abs(v.sum() - stats[0]) <= 10 * tol * stats[1] + 1e-12, f"{key}: checksum mismatch"
        return err


    def ref_values(self, prefix, name, like=None):
        """Stored reference values as a flat float64 array: the whole tensor, or its probes for compacted entries.
        Returns (ref, picker) where picker(flat_tensor) selects the matching elements of a same-shaped tensor."""
        key = f"{prefix}/{name}" if prefix else name
        if key in self.z.files:
            return self.z[key].astype(np.float64).reshape(-1), (lambda v: v.reshape(-1))
        ref = self.z[key + "#probe"].astype(np.float64)
        idx = probe_indices(int(self.z[key + "#stats"][2]))
        return ref, (lambda v: v.reshape(-1)[idx])

    def check_step_move(self, before_prefix, after_prefix, name, before, after, lr, frac_tol, move_tol=0.05):
        """Optimizer-step comparison on the MOVE of a parameter (after - before), in units of the learning rate: AdamW
        moves every weight by <= ~lr whatever the gradient's size, so weights are compared by how far they moved.
        At most `frac_tol` of the elements may differ by more than move_tol * lr (elements whose gradient is ~0 move
        by a sign-like +-lr on the first step, so rounding may flip them); returns that fraction."""
        ref_after, pick = self.ref_values(after_prefix, name)
        ref_before = self.z[f"{before_prefix}/{name}"].astype(np.float64)
        ref_move = ref_after - pick(ref_before)
        got_move = pick((after.detach().double() - before.detach().double()).cpu().numpy())
        bad = np.abs(got_move - ref_move) > move_tol * lr
        frac = float(bad.mean())
        assert frac <= frac_tol, f"{after_prefix}/{name}: {frac:.2%} of the weights moved differently (> {move_tol} lr)"
        return frac


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


@pytest.fixture
def hvc_option():
    """Setter for the library's run-time switches (hvc_set_option); every switch touched is restored after the test.  Replaces
    monkeypatch.setenv on HVC_* names: the library reads its switches from atomics, not from the environment, at launch."""
    from hvc import ops
    old = {}

    def set_(name, value):
        if name not in old:
            old[name] = ops.get_option(name)
        ops.set_option(name, int(value))
    yield set_
    for name, value in old.items():
        ops.set_option(name, value)
