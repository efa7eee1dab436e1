#!/usr/bin/env python
"""CPU model of what bounds a single-scan find: wave-level step counts of the product's traversal (tools/wavesim.c: orc_wavesim_ww) for every 8x8 tile of the C2 scan.  The launch ends with its slowest wave, so the figure of merit is the
MAX over waves of (node iterations, leaf rounds), not the mean per ray.   usage: python tools/wavesim.py [sphere|room] [modes]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402


def simlib():
    """tools/libwavesim.so (tools/wavesim.c: the models moved out of the parity oracle in round 3)"""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-s", "-C", here])
    return C.CDLL(os.path.join(here, "libwavesim.so"))


def rays_c2(mesh):
    model = syn.model_c2()
    pose = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    dirs = syn.model_directions(model)
    rot = T.transform([float(pose["R"][k]) for k in "xyzw"], (0, 0, 0))
    from scipy.spatial.transform import Rotation
    R = Rotation.from_quat([float(pose["R"][k]) for k in "xyzw"]).as_matrix()
    dm = (dirs.astype(np.float64) @ R.T).astype(np.float32)
    O = np.array([float(pose["t"][k]) for k in "xyz"], np.float32)
    return model, O, dm


def simulate(nodes, tris, model, O, dm, mode, tile=(8, 8), stride=1, seeded=False):
    L = simlib()
    L.orc_wavesim_seed.argtypes = [C.c_void_p]
    L.orc_wavesim_ww.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    H, W = model.phi.size, model.theta.size
    th, tw = tile
    img = dm.reshape(H, W, 3)
    res = []
    k = 0
    for ty in range(0, H, th):
        for tx in range(0, W, tw):
            k += 1
            if k % stride:
                continue
            d = np.ascontiguousarray(img[ty:ty + th, tx:tx + tw].reshape(-1, 3))
            o = np.ascontiguousarray(np.tile(O, (len(d), 1)))
            out = np.zeros(8, np.uint64)
            if seeded:
                # tracking mode: every ray starts with the hit distance of the previous (identical) scan as its best_t
                t_prev = np.zeros(len(d), np.float32)
                L.orc_wavesim_seed(None)
                L.orc_wavesim_ww(nodes.ctypes.data, tris.ctypes.data, o.ctypes.data, d.ctypes.data, len(d), float(model.range.max),
                                 mode, out.ctypes.data, t_prev.ctypes.data, None)
                L.orc_wavesim_seed(t_prev.ctypes.data)
            L.orc_wavesim_ww(nodes.ctypes.data, tris.ctypes.data, o.ctypes.data, d.ctypes.data, len(d), float(model.range.max),
                             mode, out.ctypes.data, None, None)
            L.orc_wavesim_seed(None)
            res.append(out.copy())
    return np.array(res, dtype=np.float64)


def report(tag, r):
    nrays = 64
    print("%-28s node iters/wave mean %5.1f p95 %5.1f max %3d | leaf rounds mean %4.1f max %2d | tri iters mean %4.1f max %2d | "
          "lane visits/ray %5.2f max ray %3d | lane efficiency %.0f%%" %
          (tag, r[:, 0].mean(), np.percentile(r[:, 0], 95), r[:, 0].max(), r[:, 1].mean(), r[:, 1].max(), r[:, 2].mean(), r[:, 2].max(),
           r[:, 3].mean() / nrays, r[:, 4].max(), 100 * r[:, 3].sum() / (r[:, 0].sum() * nrays)))


if __name__ == "__main__":
    mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
    modes = [int(a) for a in sys.argv[2:]] or [2, 3, 6, 7]
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    info, nodes, tris = ra.build_bvh_host(v, f)
    print(mesh, {k: info[k] for k in ("n_nodes", "max_depth", "stack_need")})
    model, O, dm = rays_c2(mesh)
    for mode in modes:
        report("mode %d" % mode, simulate(nodes, tris, model, O, dm, mode, stride=4))
        report("mode %d seeded" % mode, simulate(nodes, tris, model, O, dm, mode, stride=4, seeded=True))
        L = simlib()
        L.orc_wavesim_frontier_cands.restype = C.c_uint64
        for depth in (1, 2, 3, 4):
            L.orc_wavesim_frontier(depth)
            r = simulate(nodes, tris, model, O, dm, mode, stride=4)
            report("mode %d frontier depth %d" % (mode, depth), r)
            print("      candidates accepted per ray: %.2f" % (L.orc_wavesim_frontier_cands() / (64.0 * len(r))))
        L.orc_wavesim_frontier(0)
