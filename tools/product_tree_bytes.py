#!/usr/bin/env python
"""profiles/traversal_product_tree.json: what the PRODUCT's tree moves for the C2 scan -- node visits x 128 B (Node4) + triangle
records x 64 B (TriRec), counted by the CPU model of the product's traversal (tools/wavesim.c: BVH4 of the product's own builder,
frontier start at BFS depth 4, nearest-first order, leaf trigger) on the rays of the timed scan.  bench.py's extras.traversal_view
reports it next to SURVEY.md 8(d)'s nominal BVH2 / one-triangle-leaf figure.   usage: python tools/product_tree_bytes.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn  # noqa: E402
import wavesim as ws  # noqa: E402

out = {}
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    info, nodes, tris = ra.build_bvh_host(v, f)
    model, O, dm = ws.rays_c2(mesh)
    L = ws.simlib()
    L.orc_wavesim_frontier(4)
    r = ws.simulate(nodes, tris, model, O, dm, 0x40000 | 2, tile=(4, 16), stride=1)     # the product's tile shape for kind 23
    L.orc_wavesim_frontier(0)
    n_rays = model.phi.size * model.theta.size
    node_visits, records = float(r[:, 3].sum()), float(r[:, 7].sum())
    out["c2_%s100k" % mesh] = {
        "rays": n_rays, "node_visits_per_ray": round(node_visits / n_rays, 3), "records_per_ray": round(records / n_rays, 3),
        "bytes_per_scan": int(node_visits * 128 + records * 64), "bytes_per_ray": round((node_visits * 128 + records * 64) / n_rays, 1),
        "tree": {k: info[k] for k in ("n_nodes", "max_depth", "stack_need")},
        "model": "tools/wavesim.c mode 0x40002 (nearest-first with far-distance tie-break, leaf trigger), frontier depth 4, 16x4 tiles; "
                 "the frontier table itself (<= 256 x 32 B per wave, cooperative) is not included"}
    print(mesh, out["c2_%s100k" % mesh])
with open(os.path.join(ROOT, "profiles", "traversal_product_tree.json"), "w") as fh:
    json.dump(out, fh, indent=1)
