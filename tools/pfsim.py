#!/usr/bin/env python
"""CPU model of the particle-filter kernel's schedule (tools/pfsim.c) on sampled blocks of config C4.
usage: python tools/pfsim.py [sphere|room] [max_leaf] [nblocks]"""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn  # noqa: E402


class Out(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("issue", "lane_issue", "node_issue", "leaf_issue", "refill_issue", "loop_issue",
                                          "node_lane", "leaf_lane", "refill_lane", "makespan", "nvisit", "lvisit", "rays", "deep_steps", "node_steps", "max_sp")]


def lib():
    subprocess.check_call(["make", "-s", "-C", HERE])
    L = C.CDLL(os.path.join(HERE, "libpfsim.so"))
    L.pfsim_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.POINTER(Out)]
    L.pfsim_costs.argtypes = [C.c_void_p]
    return L


def block_rays(poses, dirs, order=None):
    """rays of one block in queue order: particle-major, beams in `order`"""
    from scipy.spatial.transform import Rotation
    q = np.stack([poses["R"][k] for k in "xyzw"], -1).astype(np.float64)
    t = np.stack([poses["t"][k] for k in "xyz"], -1).astype(np.float32)
    R = Rotation.from_quat(q).as_matrix()
    d = dirs if order is None else dirs[order]
    D = np.einsum("pij,bj->pbi", R, d.astype(np.float64)).astype(np.float32)
    O = np.broadcast_to(t[:, None, :], D.shape)
    return np.ascontiguousarray(O.reshape(-1, 3)), np.ascontiguousarray(D.reshape(-1, 3))


def run(L, nodes, tris, poses, dirs, pb, nblocks, slots, thr, tail, order=None, nwaves=4):
    tot = np.zeros(16)
    for b in range(nblocks):
        O, D = block_rays(poses[b * pb:(b + 1) * pb], dirs, order)
        o = Out()
        L.pfsim_block(nodes.ctypes.data, tris.ctypes.data, O.ctypes.data, D.ctypes.data, len(O), 1e30, slots, thr, tail, nwaves, C.byref(o))
        tot += np.array([getattr(o, n) for n, _ in Out._fields_])
    return dict(zip([n for n, _ in Out._fields_], tot))


def coherent_order(dirs, group=64):
    """beam order that makes every run of `group` beams a compact bundle: sort by azimuth sector, then elevation"""
    az = np.arctan2(dirs[:, 1], dirs[:, 0])
    el = np.arcsin(np.clip(dirs[:, 2], -1, 1))
    nsec = max(1, len(dirs) // group)
    sec = np.minimum(((az + math.pi) / (2 * math.pi) * nsec).astype(int), nsec - 1)
    return np.lexsort((el, az, sec))


if __name__ == "__main__":
    mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
    max_leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    L = lib()
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    info, nodes, tris = ra.build_bvh_host(v, f)
    if max_leaf == 2:
        info, nodes, _ = ra.build_bvh_host_pf(v, f)
    print(mesh, "max_leaf", max_leaf, {k: info[k] for k in ("n_nodes", "max_depth", "stack_need")})
    if mesh == "sphere":
        poses, _ = syn.uniform_particles(16 * nblocks * 2, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    else:
        poses, _ = syn.uniform_particles(16 * nblocks * 2, seed=42, bb_min=(-9, -9, 0.3, 0, 0, -math.pi), bb_max=(9, 9, 3, 0, 0, math.pi))
    dirs = syn.model_directions(syn.model_pf16()).reshape(-1, 3)
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    coh = coherent_order(dirs)

    def show(tag, r, pb):
        nb_total = 100000 / pb
        ms = r["issue"] / nblocks * nb_total * 4 / 1024 / 2.4e6
        print("%-44s issues/ray %7.1f lanes %.3f | node %.3f(%4.1f%%) leaf %.3f(%4.1f%%) refill %.3f(%4.1f%%) | visits/ray %5.2f leaves/ray %4.2f | est %.2f ms" % (
            tag, r["issue"] / r["rays"] * 64 / 64, r["lane_issue"] / r["issue"],
            r["node_lane"] / max(r["node_issue"], 1), 100 * r["node_issue"] / r["issue"],
            r["leaf_lane"] / max(r["leaf_issue"], 1), 100 * r["leaf_issue"] / r["issue"],
            r["refill_lane"] / max(r["refill_issue"], 1), 100 * r["refill_issue"] / r["issue"],
            r["nvisit"] / r["rays"], r["lvisit"] / r["rays"], ms))

    import ctypes
    def costs(node=125, tri=75, leaf_fixed=12, ev=110, setup=200, loop=14, sn=14, sl=14):
        a = (ctypes.c_double * 8)(node, tri, leaf_fixed, ev, setup, loop, sn, sl)
        L.pfsim_costs(a)

    def go(tag, pb, slots, thr, tail, order=None):
        nb = nblocks * 8 // pb
        r = run(L, nodes, tris, poses, dirs, pb, nb, slots, thr, tail, order)
        nb_total = 100000 / pb
        merge = 44.0 * 256 * pb / 8 / (pb * 256)   # issues per ray of the 8-lane in-order merge (one wave)
        ms = (r["issue"] / nb + merge * pb * 256) * nb_total * 4 / 1024 / 2.4e6
        print("   wave node steps that leave the 20 LDS stack rows: %.2f %%  (max sp %d)" % (100 * r["deep_steps"] / max(r["node_steps"], 1), r["max_sp"] / nb))
        print("%-50s issues/ray %6.1f lanes %.3f | node %.3f(%4.1f%%) leaf %.3f(%4.1f%%) refill %.3f(%4.1f%%) | visits %5.2f | est %.2f ms (with merge)" % (
            tag, r["issue"] / r["rays"], r["lane_issue"] / r["issue"],
            r["node_lane"] / max(r["node_issue"], 1), 100 * r["node_issue"] / r["issue"],
            r["leaf_lane"] / max(r["leaf_issue"], 1), 100 * r["leaf_issue"] / r["issue"],
            r["refill_lane"] / max(r["refill_issue"], 1), 100 * r["refill_issue"] / r["issue"],
            r["nvisit"] / r["rays"], ms))

    costs()
    go("r2 kernel refill>=48 tail<=8", 8, 1, 48, 8)
    print("-- eval out of the loop (25) + prepared next ray (swap 15; + 170 x 512 / 51 per wave dense = +3.3 issues/ray not shown)")
    costs(ev=25, setup=15)
    for thr in (4, 8, 16, 24, 32, 48):
        go("1 slot swap-in refill>=%d tail<=8" % thr, 8, 1, thr, 8)
    for thr in (8, 16, 24):
        go("1 slot swap-in refill>=%d tail<=8 PB=16" % thr, 16, 1, thr, 8)
    print("-- multi-slot upper bound: no slot overhead")
    costs(ev=25, setup=170, sn=0, sl=0)
    for slots in (2, 3):
        go("%d slots refill>=56 ratio 100" % slots, 8, slots, 56, 100)
