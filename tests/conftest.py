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
    config.addinivalue_line("markers", "lab: test drives an EXPERIMENT of librmclhip_lab.so (a rejected traversal kind, an older "
                                       "particle-filter kernel); also marked gpu.  -m 'gpu and not lab' = the product alone")


# traversal kinds librmclhip.so builds (lab_hooks.h: find_kind_in_product); 15 = the automatic rule
PRODUCT_FIND_KINDS = (0, 2, 15, 23, 24, 32)


def find_kinds(*kinds):
    """parametrize list of find traversal kinds: kinds of the experiments library carry the `lab` marker"""
    return [k if k in PRODUCT_FIND_KINDS else pytest.param(k, marks=pytest.mark.lab) for k in kinds]


def pf_variant_needs_lab(v):
    """rmclhip_pf_set_variant: rounds (bits 4..6 = 0), the round kernels' traversals (bits 0..1), full nodes (bit 7) and the
    round-2 kernel (bit 8) are experiments"""
    return ((v >> 4) & 7) == 0 or (v & 3) != 0 or (v & 128) != 0 or (v & 256) != 0


def pf_variants(*variants):
    return [pytest.param(v, marks=pytest.mark.lab) if pf_variant_needs_lab(v) else v for v in variants]


@pytest.fixture(autouse=True)
def _experiments_library(request):
    """tests marked `lab` get librmclhip_lab.so loaded (once per process); nothing else ever loads it"""
    if request.node.get_closest_marker("lab") is not None:
        import rmcl_amd
        rmcl_amd.load_lab()


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
            elif name == "sphere10m":
                cache[name] = syn.uv_sphere(10000000)
            elif name == "chain200":
                cache[name] = syn.exp_chain(200, 1.5)
            elif name == "chain2000":
                cache[name] = syn.exp_chain(2000, 1.05)
            elif name == "nested200":
                cache[name] = syn.nested_triangles(200, 1.2, 1e-3)
            elif name == "fan200k":
                cache[name] = syn.sliver_fan(200000)
            elif name == "fan20k":
                cache[name] = syn.sliver_fan(20000)
            elif name == "cadmix20k":
                cache[name] = syn.cad_mix(20000)
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
