import os
import sys

import numpy as np
import pytest

# torch ships its own HIP runtime: import it before the tracer library is loaded so that one
# process never ends up with the runtime copies initialised in the other order (torch then
# reports "No HIP GPUs are available" when a test allocates its first tensor).
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_scene(name):
    import raytracer_amd as ra
    return ra.SceneArrays.load(os.path.join(GOLD, "scenes", name + ".npz"))


def golden_cases():
    return sorted(f[len("render_"):-4] for f in os.listdir(GOLD) if f.startswith("render_") and f.endswith(".npz"))


def load_case(case):
    """-> (SceneArrays configured as rendered, roi, golden dict)"""
    z = np.load(os.path.join(GOLD, f"render_{case}.npz"), allow_pickle=False)
    sname = bytes(z["scene"]).decode()
    w, h, spp, depth, proj, seed = (int(v) for v in z["params"])
    s = load_scene(sname).copy(width=w, height=h, spp=spp, max_depth=depth, proj=proj, seed=seed)
    roi = tuple(int(v) for v in z["roi"])
    return s, roi, {k: z[k] for k in ("rgb", "rgbf", "counts")}


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def kat():
    z = np.load(os.path.join(GOLD, "kat.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def tracer():
    """One GPU context for the whole session (GPU tests only)."""
    import raytracer_amd as ra
    t = ra.Tracer(device=0)
    yield t
    t.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_float_image(a, b):
    """bit equality, treating any NaN as equal to any NaN"""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def hip_runtime_version():
    """HIP_VERSION of the runtime this process runs on (torch's bundled one when torch was imported first)"""
    import ctypes
    v = ctypes.c_int(0)
    rc = ctypes.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(ctypes.byref(v))     # the copy already loaded, by soname
    assert rc == 0
    return v.value
