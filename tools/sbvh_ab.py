#!/usr/bin/env python
"""Spatial splits (SBVH, bvh_build.cpp SplitBuilder) A/B: maps built with RMCLHIP_SBVH_ALPHA = 0 (object splits only, rounds 1-5) and with the
default budget, in child processes (the knob is read once per process): build time, records, find time of a 64x512 / 128x1024 scan.
usage (GPU box): python tools/sbvh_ab.py"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, os, time, json, math
sys.path.insert(0, %r)
import numpy as np
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
f32 = np.float32
ctx = ra.Context(0)
out = {}
H, W = 64, 512
cases = (
  ("sliver_fan_200k", lambda: syn.sliver_fan(200000), T.spherical_model(f32(-1.5), f32(3.0 / (H - 1)), H, f32(-math.pi), f32(2 * math.pi / W), W, f32(0.01), f32(1e6)), T.transform_from_rpy((1.0, 2.0, 3.0), (0.1, 0.2, 0.3))),
  ("sliver_fan_20k", lambda: syn.sliver_fan(20000), T.spherical_model(f32(-1.5), f32(3.0 / (H - 1)), H, f32(-math.pi), f32(2 * math.pi / W), W, f32(0.01), f32(1e6)), T.transform_from_rpy((1.0, 2.0, 3.0), (0.1, 0.2, 0.3))),
  ("cadmix_100k", lambda: syn.cad_mix(100000), syn.model_c2(), T.transform_from_rpy((-6.0, 0.5, 1.5), (0.3, 0.1, -1.0))),
  ("cadmix_100k_8_beams_turned", lambda: syn.cad_mix(100000, beam_yaw_deg=35.0, beam_tilt_deg=12.0), syn.model_c2(), T.transform_from_rpy((-6.0, 0.5, 1.5), (0.3, 0.1, -1.0))),
  ("cadmix_100k_200_beams_turned", lambda: syn.cad_mix(100000, beam_yaw_deg=35.0, beam_tilt_deg=12.0, n_beams=200), syn.model_c2(), T.transform_from_rpy((-6.0, 0.5, 1.5), (0.3, 0.1, -1.0))),
  ("cadmix_100k_2000_beams_turned", lambda: syn.cad_mix(100000, beam_yaw_deg=35.0, beam_tilt_deg=12.0, n_beams=2000), syn.model_c2(), T.transform_from_rpy((-6.0, 0.5, 1.5), (0.3, 0.1, -1.0))),
  ("cadmix_20k", lambda: syn.cad_mix(20000), syn.model_c2(), T.transform_from_rpy((-6.0, 0.5, 1.5), (0.3, 0.1, -1.0))),
  ("sphere_100k", lambda: syn.uv_sphere(100000), syn.model_c2(), syn.pose_c2_truth()),
  ("room_100k", lambda: syn.noisy_room(100000), syn.model_c2(), T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))),
)
for name, gen, model, pose in cases:
    v, f = gen()
    t0 = time.perf_counter()
    hm = ra.import_hip_map(ctx, v, f)
    build = time.perf_counter() - t0
    info = hm.info()
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    first = rcc.time_find(pose, 1)
    ms = sorted(rcc.time_find(pose, 20 if first < 1.0 else 2) for _ in range(5))[2]
    rcc.find(pose)
    mv = rcc.modelView()
    import hashlib
    out[name] = dict(build_s=round(build, 2), faces=info["n_faces"], records=info["n_tri_records"], spatial_splits=info["spatial_splits"], nodes=info["n_nodes"],
                     stack_need=info["stack_need"], find_us=round(ms * 1e3, 2), hits=int(mv["hits"].sum()),
                     face_ids_sha=hashlib.sha256(mv["face_ids"].tobytes()).hexdigest()[:16], ranges_sha=hashlib.sha256(mv["ranges"].tobytes()).hexdigest()[:16])
    rcc.close(); hm.release()
print(json.dumps(out))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for alpha in ("0", "1"):
    env = dict(os.environ, RMCLHIP_SBVH_ALPHA=alpha)
    res[alpha] = json.loads(subprocess.check_output([sys.executable, "-c", CHILD % root], env=env).decode().strip().splitlines()[-1])
print("%-18s %9s %9s %8s | %9s %9s %8s %8s | %s" % ("map", "build s", "find us", "records", "build s", "find us", "records", "splits", "same results"))
for name in res["0"]:
    a, b = res["0"][name], res["1"][name]
    print("%-18s %9.2f %9.2f %8d | %9.2f %9.2f %8d %8d | %s" % (name, a["build_s"], a["find_us"], a["records"], b["build_s"], b["find_us"], b["records"], b["spatial_splits"],
          "yes" if (a["face_ids_sha"], a["ranges_sha"], a["hits"]) == (b["face_ids_sha"], b["ranges_sha"], b["hits"]) else "NO"))
