#!/usr/bin/env python
"""How often does the {sequence, xor} completion tag actually catch something?  (profiles/r05_tag_handoff.txt)
Round 3 saw the host accept a tag and read the previous call's result block, 1 in ~10^4 calls of find_async + computeCrossStatistics.
The library counts, per process, every poll that found ITS sequence number in the tag while the words it read did not add up to the
tag's checksum (rmclhip_debug_tag_retries).  This drives the three tagged result paths N times each and prints the count:
  (1) find_async + streaming computeCrossStatistics (64-B CrossStatistics block + tag, two allocations), results compared bit for bit
  (2) correct_once, moment form: the 37 KB host block of k_micp_publish
  (3) correct_once, device loop (MicpState + status blocks)
   usage: python tools/tag_retries.py [calls per path = 300000]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
L = ra._capi.lib()


def retries():
    out = C.c_ulonglong(0)
    ra._capi.check(L.rmclhip_debug_tag_retries(C.byref(out)))
    return out.value


ctx = ra.Context(0)
v, f = syn.uv_sphere(20000)
hm = ra.import_hip_map(ctx, v, f)
model = syn.model_vlp16_900(0.0)
rcc = ra.RCCHipSpherical(hm)
rcc.setTsb(T.identity())
rcc.setModel(model)
rcc.find(T.identity())
mv0 = rcc.modelView()
rcc.set_dataset(mv0["points"].reshape(-1, 3), mv0["hits"].reshape(-1))
rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
poses = (T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2)), T.transform((0, 0, 0, 1), (0.0, 0.0, 0.15)))
# (1)
rcc.set_micp_fast(0)
ref = []
for P in poses:
    rcc.find(P)
    ref.append(rcc.computeCrossStatistics(T.identity()).tobytes())
r0, bad, t0 = retries(), 0, time.perf_counter()
for i in range(n):
    rcc.find_async(poses[i & 1])
    bad += rcc.computeCrossStatistics(T.identity()).tobytes() != ref[i & 1]
print("(1) find_async + streaming computeCrossStatistics: %d calls, %d wrong results, %d checksum retries, %.1f us / call"
      % (n, bad, retries() - r0, (time.perf_counter() - t0) / n * 1e6), flush=True)
# (2), (3)
for mode, what in ((1, "(2) correct_once, host moment form"), (4, "(3) correct_once, device loop")):
    rcc.set_micp_fast(mode)
    ident = T.identity()
    refs = [rcc.correct_once(P, ident, 5, 0.0, False)[0].tobytes() for P in poses]
    refs = [rcc.correct_once(P, ident, 5, 0.0, False)[0].tobytes() for P in poses]
    r0, bad, t0 = retries(), 0, time.perf_counter()
    for i in range(n):
        bad += rcc.correct_once(poses[i & 1], ident, 5, 0.0, False)[0].tobytes() != refs[i & 1]
    print("%s: %d calls, %d wrong results, %d checksum retries, %.1f us / call" % (what, n, bad, retries() - r0, (time.perf_counter() - t0) / n * 1e6), flush=True)
