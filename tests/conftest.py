import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth(seed, m, n, sy=60, sx=60):
    """Same generator as oracle/gen_golden.py::synth (seeded inputs that are
    too large to commit; verified against the stored SHA-256)."""
    rng = np.random.default_rng(seed)
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    return exp, dic


@pytest.fixture(scope="session")
def synth_inputs():
    g = load_golden("di_synth.npz")
    exp, dic = synth(int(g["seed"]), int(g["m"]), int(g["n"]))
    assert sha(exp) == str(g["exp_sha"]) and sha(dic) == str(g["dic_sha"]), (
        "numpy Generator stream differs from the one the goldens were made with"
    )
    return exp, dic, g


@pytest.fixture(scope="session")
def config1_inputs():
    g = load_golden("config1_ni.npz")
    exp = g["exp"]
    rng = np.random.default_rng(int(g["seed"]))
    base = exp.reshape(9, 60, 60).astype(np.float32) / 255.0
    noise = rng.random((1000, 60, 60)).astype(np.float32)
    wgt = (np.float32(0.35) + np.float32(0.5) * rng.random(1000).astype(np.float32))
    dic = base[np.arange(1000) % 9] * wgt[:, None, None] + noise * (np.float32(1) - wgt)[:, None, None]
    dic[0:999:111] = base
    dic = dic.astype(np.float32)
    return exp, dic, g


def gpu_available():
    """True when the HIP runtime sees a device (no torch involved)."""
    try:
        from kikuchipy_amd import _lib

        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _no_engine_left_behind():
    """`dictionary_indexing` keeps the engine of a finished call for the next one (kikuchipy_amd._lib: engine pool); a test
    must not inherit another test's engine (stand-ins, patched factories, environment switches read at creation)."""
    yield
    try:
        from kikuchipy_amd import _lib

        _lib.clear_engine_cache()
    except Exception:  # noqa: BLE001 - the library may be unbuildable in a test of exactly that
        pass
