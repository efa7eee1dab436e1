import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
ctx = ra.Context(0)
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    t0 = time.perf_counter(); hm = ra.import_hip_map(ctx, v, f); print(mesh, "map", round(time.perf_counter() - t0, 3), "s")
    model = syn.model_c2()
    truth = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    est = T.mult(truth, syn.pose_c2_perturbation())
    rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(model); rcc.find(truth); mv = rcc.modelView()
    for variant in (15,):
        cpc = ra.CPCHip(hm); cpc.setTsb(T.identity()); cpc.params.max_dist = 1.0
        cpc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
        t0 = time.perf_counter(); cpc.find(est); print("  first find (builds the grid)", round((time.perf_counter() - t0) * 1e3, 2), "ms")
        for label, tr, gr in (("cold + grid", False, True), ("cold bare", False, False), ("tracking", True, True)):
            cpc.set_tracking(tr); cpc.set_grid(gr); cpc.find(est)
            t1 = time.perf_counter()
            for _ in range(30): cpc.find(est)
            print("  %-12s %.4f ms" % (label, (time.perf_counter() - t1) / 30 * 1e3))
        cpc.close()
