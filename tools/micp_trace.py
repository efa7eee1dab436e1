#!/usr/bin/env python
"""C3 schedule (R), moment form, direct launches: 300 corrections back to back (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
rcc = ra.RCCHipSpherical(hm)
rcc.set_variant(15)
rcc.setTsb(T.identity())
rcc.setModel(syn.model_c2())
rcc.find(syn.pose_c2_truth())
rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
est = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation())
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if len(sys.argv) > 2:
    rcc.set_micp_fast(int(sys.argv[2]))   # 1 = moments in the find's epilogue (default), 3 = in a pass of their own
for _ in range(300):
    rcc.correct_once(est, T.identity(), n_iter, 0.0, False)
print(rcc.micp_fast_info())
