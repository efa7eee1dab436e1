"""pytest configuration.

Markers:  gpu -- needs a real MI355X; everything else runs on CPU.
The oracle (oracle/) is imported ONLY from tests (and smoke/bench's cpu_baseline); the product
package rmcl_amd never sees it.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()  # builds librmcl_oracle.so if needed
    return oracle


@pytest.fixture(scope="session")
def ra():
    import rmcl_amd
    rmcl_amd._capi.lib()  # raises if librmclhip.so is missing: no silent fallback
    return rmcl_amd


@pytest.fixture(scope="session")
def ctx(ra):
    """HIP context; a missing device is a hard failure for -m gpu tests (never a skip)."""
    return ra.Context(0)


@pytest.fixture(scope="session")
def meshes():
    from rmcl_amd import synthetic as syn
    cache = {}

    def get(name):
        if name not in cache:
            if name == "cube":
                cache[name] = syn.cube_room()
            elif name == "sphere100k":
                cache[name] = syn.uv_sphere(100000)
            elif name == "sphere20k":
                cache[name] = syn.uv_sphere(20000)
            elif name == "room30k":
                cache[name] = syn.noisy_room(30000)
            elif name == "room100k":
                cache[name] = syn.noisy_room(100000)
            elif name == "sphere1m":
                cache[name] = syn.uv_sphere(1000000)
            else:
                raise KeyError(name)
        return cache[name]

    return get


def golden_path(name):
    return os.path.join(GOLDEN, name)


def assert_close_rel(a, b, rtol=1e-5, atol=0.0, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "%s: NaN pattern differs" % what
    ok = np.abs(a - b) <= atol + rtol * np.abs(b)
    ok |= nan_a
    assert ok.all(), "%s: max rel err %.3g at %d of %d" % (
        what, np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)), (~ok).sum(), ok.size)
