"""Manual measurement helper (not a test): traversal work per ray of the PRODUCT's BVH4 on the C2 scan, counted by
the oracle's instrumented walk.  Usage: python tests/bvh_quality.py [mesh] [stride]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as orc  # noqa: E402
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402


def measure(nodes, tris, O, dirs, tfar, mode):
    L = orc.lib()
    L.orc_trace_bvh4_ordered.argtypes = [C.c_void_p, C.c_void_p, orc.Vec3, orc.Vec3, C.c_float, C.c_float, C.c_int,
                                         C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    cnt = np.zeros(5, np.uint64)
    t, f = C.c_float(), C.c_uint32()
    hits = 0
    for d in dirs:
        hits += L.orc_trace_bvh4_ordered(nodes.ctypes.data, tris.ctypes.data, orc.Vec3(*O), orc.Vec3(*d), 0.0, tfar, mode,
                                         cnt.ctypes.data, C.byref(t), C.byref(f))
    n = len(dirs)
    return dict(nodes=cnt[0] / n, dead=cnt[1] / n, leaves=cnt[2] / n, tris=cnt[3] / n, stack=int(cnt[4]), hits=hits / n)


if __name__ == "__main__":
    mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere100k"
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 37
    v, f = {"sphere100k": lambda: syn.uv_sphere(100000), "room100k": lambda: syn.noisy_room(100000)}[mesh]()[:2]
    info, nodes, tris = ra.build_bvh_host(v, f)
    model = syn.model_c2()
    dirs = syn.model_directions(model)[::stride]
    pose = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation()) if mesh == 'sphere100k' else T.transform_from_rpy((0.5, -0.3, 1.2), (0.02, -0.03, 0.4))
    Tsm = pose
    O = [float(Tsm["t"][k]) for k in "xyz"]
    rot = T.transform([float(Tsm["R"][k]) for k in "xyzw"], (0, 0, 0))
    dm = np.array([orc.tapply(rot, d) for d in dirs], np.float32)
    print(mesh, info)
    for mode in (0, 1, 3):
        print("mode", mode, measure(nodes, tris, O, dm, float(model.range.max), mode))


def per_ray_counts(nodes, tris, O, dm, tfar, mode):
    L = orc.lib()
    L.orc_trace_bvh4_ordered.argtypes = [C.c_void_p, C.c_void_p, orc.Vec3, orc.Vec3, C.c_float, C.c_float, C.c_int,
                                         C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    out = np.zeros((len(dm), 5), np.uint64)
    t, f = C.c_float(), C.c_uint32()
    for i, d in enumerate(dm):
        L.orc_trace_bvh4_ordered(nodes.ctypes.data, tris.ctypes.data, orc.Vec3(*O), orc.Vec3(*d), 0.0, tfar, mode,
                                 out[i].ctypes.data, C.byref(t), C.byref(f))
    return out
