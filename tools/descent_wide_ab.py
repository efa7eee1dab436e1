import os, sys
sys.path.insert(0, os.getcwd())
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T, _capi
ctx = ra.Context(0)
for mesh in ("sphere", "room", "sphere1m"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else (syn.noisy_room(100000) if mesh == "room" else syn.uv_sphere(1000000))
    hm = ra.import_hip_map(ctx, v, f)
    base = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4)) if mesh == "room" else syn.pose_c2_truth()
    rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(syn.model_c2())
    def t(kind, cap=64, lev=24):
        rcc.set_traversal(kind)
        _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, lev))
        return sorted(rcc.time_find(base, 30) for _ in range(7))[3] * 1e3
    print("%s: kind 23 %.2f | kind 31 narrow cap 64 %.2f cap 12 %.2f | wide cap 64 %.2f cap 32 %.2f cap 16 %.2f cap 12 %.2f" % (
        mesh, t(23), t(31, 64, 24 | (1 << 31)), t(31, 12, 24 | (1 << 31)), t(31, 64), t(31, 32), t(31, 16), t(31, 12)), flush=True)
    rcc.close(); hm.release()
