"""pytest configuration: `gpu` marker + shared helpers for loading golden fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_box():
    """True when this machine exposes an AMD GPU (/dev/kfd).  Decided WITHOUT the HIP library on purpose: on a GPU
    box a missing / unloadable libhudiff_hip.so must make the gpu tests FAIL (there is no CPU fallback), while on a
    machine without any GPU a plain `pytest tests` skips them instead of failing with HD_ERR_NO_DEVICE."""
    return os.path.exists("/dev/kfd") or os.environ.get("HUDIFF_REQUIRE_GPU") == "1"


def pytest_collection_modifyitems(config, items):
    if _gpu_box():
        return
    skip = pytest.mark.skip(reason="no AMD GPU on this machine (/dev/kfd absent); run with -m gpu on an MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_cfg(kind):
    raw = np.load(os.path.join(GOLDEN, f"micro_{kind}_config.npz"))
    cfg = {}
    for k, v in raw.items():
        v = v.item() if v.ndim == 0 else v
        cfg[k] = str(v) if isinstance(v, (str, np.str_)) else v
    return cfg


def load_weights(kind):
    return dict(np.load(os.path.join(GOLDEN, f"micro_{kind}_weights.npz")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def load_deep(kind):
    """tests/golden/deep_{kind}.npz (oracle/make_golden_deep.py): production depth / heads at small width.  Weights are
    regenerated from the recorded seed and checked against the recorded SHA-256.  -> (fixture, config, state_dict)"""
    import hashlib
    from hudiff_amd import synthetic as S
    z = np.load(os.path.join(GOLDEN, f"deep_{kind}.npz"))
    cfg = {}
    for k, v in zip(z["config_keys"], z["config_vals"]):
        v = str(v)
        cfg[str(k)] = v if k == "activation" else (float(v) if k == "dropout" else int(v))
    sd = S.random_state_dict(kind, cfg, seed=int(z["weight_seed"]))
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes())
    assert h.hexdigest() == str(z["weight_sha256"]), "regenerated weights differ from the ones the reference ran with"
    return z, cfg, sd


def unpack_masks(z, key):
    shape = tuple(int(x) for x in z[key.replace("masks", "shape")])
    return np.unpackbits(z[key])[: int(np.prod(shape))].reshape(shape)


def prec(model, **expect):
    """Assert a subset of hd_precision_report (model.precision_info()): e.g. prec(m, precision="split", range_fallbacks=0)."""
    info = model.precision_info()
    bad = {k: (info.get(k), v) for k, v in expect.items() if info.get(k) != v}
    assert not bad, (bad, info)
    return info


def env_forces_route():
    """True when the environment overrides the library's default precision route (the nested whole-suite run does)."""
    return any(os.environ.get(k) not in (None, "") for k in ("HUDIFF_PRECISION", "HUDIFF_X3", "HUDIFF_ATTN_X3"))


def chain_or_none(z):
    return z["chain"] if z["chain"].size else None


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
