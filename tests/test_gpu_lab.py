"""The seam between the product (librmclhip.so) and the experiments (librmclhip_lab.so, include/rmclhip_lab.h): without the
experiments library the product refuses their kernel variants loudly; with it loaded they run and give the product's results."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_PRODUCT_ONLY = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
assert ra._capi._lab is None
ctx = ra.Context(0)
v, f = syn.cube_room()
hm = ra.import_hip_map(ctx, v, f)
rcc = ra.RCCHipSpherical(hm)
rcc.setTsb(T.identity())
rcc.setModel(syn.model_c1())
for kind in (0, 2, 15, 23, 24, 32):              # the product's own kinds
    rcc.set_traversal(kind)
    rcc.find(syn.pose_c2_truth())
for kind in (4, 5, 6, 13, 17, 19, 20, 22, 31):    # experiments: refused at set_variant
    try:
        rcc.set_traversal(kind)
        raise SystemExit("kind %%d accepted without the experiments library" %% kind)
    except ra.RmclHipError as e:
        assert "librmclhip_lab.so" in str(e), str(e)
try:
    rcc.debug_wave_clock(syn.pose_c2_truth())
    raise SystemExit("clocked launch ran without the experiments library")
except ra.RmclHipError as e:
    assert "librmclhip_lab.so" in str(e), str(e)
upd = ra.PCDSensorUpdaterHip(hm)
upd.init()
for variant in (0, 2, 64 | 256, 48 | 128):       # round kernels / the round-2 kernel: experiments
    try:
        upd.set_variant(variant)
        raise SystemExit("pf variant %%d accepted without the experiments library" %% variant)
    except ra.RmclHipError as e:
        assert "librmclhip_lab.so" in str(e), str(e)
upd.set_variant(64)
print("PRODUCT-ONLY-OK")
"""


def test_product_alone_refuses_experiments():
    """fresh process that never loads librmclhip_lab.so"""
    out = subprocess.run([sys.executable, "-c", _PRODUCT_ONLY % ROOT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "PRODUCT-ONLY-OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.lab
def test_experiments_load_and_match_the_product(ra, ctx, meshes):
    from rmcl_amd import synthetic as syn, types as T
    lab = ra.load_lab()
    assert b"experiments" in lab.rmclhip_lab_version()
    v, f = meshes("room30k")
    hm = ra.import_hip_map(ctx, v, f)
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(syn.tsb_offset())
    rcc.setModel(syn.model_c1())
    pose = T.transform_from_rpy((1.0, -1.5, 1.2), (0.0, 0.0, 0.5))
    rcc.set_traversal(23)
    rcc.find(pose)
    ref = rcc.modelView()
    for kind in (1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 20, 21, 22):
        rcc.set_traversal(kind)
        rcc.find(pose)
        mv = rcc.modelView()
        for k in ("hits", "ranges", "face_ids", "points", "normals"):
            assert mv[k].tobytes() == ref[k].tobytes(), (kind, k)
    clocks = rcc.debug_wave_clock(pose)
    assert clocks.shape[1] == 8 and (clocks[:, 1] != 0).any()
    rcc.close()
