#!/usr/bin/env python
"""A/B of the particle filter's forms (round 5; rmclhip_pf_set_variant):
   sorted  bit 12 (rounds 3 / 4): children sorted by entry distance, beam errors in global scratch, dense likelihood pass, one lane per
           particle walking the in-order Gaussian1D chain
   slot    bits 11 + 12: the ray's SLOT order instead of the sorting network (bvh_build.cpp orders a node's children along an axis)
   accum   the default since round 5: order-independent likelihood accumulation -- fixed-point accumulators in LDS, closed-form merge
           weights, no per-beam storage, no chain
Config C4 on sphere-100k and room-100k, uniform and converged clouds, and the C5 shard (125 000 particles, 1 M triangles).
sorted == slot bit for bit; accum agrees to float rounding (max relative difference printed).
   usage: python tools/pf_forms_ab.py [sorted|slot|accum]     (one form only: the PMC passes of tools/pmc_sets.sh)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

forms = {"sorted": 64 | 4096, "slot": 64 | 2048 | 4096, "accum": 64}
only = sys.argv[1] if len(sys.argv) > 1 else None
ctx = ra.Context(0)
for mesh, n, bb, centre in (("sphere100k", 100000, ((-5, -5, -1), (5, 5, 1)), T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4))),
                            ("room100k", 100000, ((-9, -9, 0.3), (9, 9, 3)), T.transform_from_rpy((1.5, -2.0, 1.6), (0, 0, 0.4))),
                            ("sphere1m", 125000, ((-5, -5, -1), (5, 5, 1)), None)):
    v, f = syn.noisy_room(100000) if mesh.startswith("room") else syn.uv_sphere(1000000 if mesh.endswith("1m") else 100000)
    hm = ra.import_hip_map(ctx, v, f)
    row = {}
    for name, var in forms.items():
        if only and name != only:
            continue
        ms, _ = bench._pf_c4(ra, syn, T, np, ctx, hm, n, 256, iters=3, bb=bb, variant=var)
        mc = bench._pf_c4(ra, syn, T, np, ctx, hm, n, 256, iters=3, converged_at=centre, variant=var)[0] if centre is not None else float("nan")
        row[name] = (ms, mc)
    print("%-10s " % mesh + "   ".join("%s: uniform %.4f ms converged %.4f ms" % (k, a, b) for k, (a, b) in row.items()), flush=True)
    if not only:
        # identical results
        poses, attrs = syn.uniform_particles(20000, seed=1, bb_min=bb[0] + (0, 0, -3.14), bb_max=bb[1] + (0, 0, 3.14))
        beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
        outs = []
        for var in forms.values():
            upd = ra.PCDSensorUpdaterHip(hm)
            upd.init()
            upd.set_variant(var)
            upd.setInput(beams, T.identity())
            d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
            upd.update(d_p, d_a)
            outs.append(d_a.download())
            upd.close()
        m0, m2 = outs[0]["likelihood"]["mean"].astype(np.float64), outs[2]["likelihood"]["mean"].astype(np.float64)
        s0, s2 = outs[0]["likelihood"]["sigma"].astype(np.float64), outs[2]["likelihood"]["sigma"].astype(np.float64)
        print("           sorted == slot: %s   accum vs sorted: n_meas equal %s, max rel diff mean %.2e, sigma %.2e (abs %.2e)" % (
            outs[0].tobytes() == outs[1].tobytes(), np.array_equal(outs[0]["likelihood"]["n_meas"], outs[2]["likelihood"]["n_meas"]),
            np.max(np.abs(m2 - m0) / np.abs(m0)), np.max(np.abs(s2 - s0) / np.maximum(np.abs(s0), 1e-30)), np.max(np.abs(s2 - s0))), flush=True)
    hm.release()
