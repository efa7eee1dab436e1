#!/usr/bin/env python
"""pose estimate of a 100 000-particle cloud through the multi-GPU C ABI on one GPU, 50 calls (for rocprofv3)."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

v, f = syn.uv_sphere(100000)
shp = ra.ShardedParticleFilterHip(v, f, devices=(0,))
poses, attrs = syn.uniform_particles(100000, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
attrs["likelihood"]["mean"] = np.random.RandomState(1).uniform(0, 1, len(attrs))
shp.set_particles(poses, attrs)
shp.pose_estimate()
t0 = time.perf_counter()
for _ in range(50):
    shp.pose_estimate()
print("pose estimate: %.1f us per call" % ((time.perf_counter() - t0) / 50 * 1e6))
shp.close()
