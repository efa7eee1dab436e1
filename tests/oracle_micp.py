"""Oracle-side (CPU) restatement of the callers of the hot path, used only by tests and the golden
generator: dataset construction (MICPSphericalSensorCPU.cpp:181-233), the MICP-L inner loop
(micp_localization.cpp:900-964 + MICPSensor.hpp:146-184) and the v1 batch corrector loop
(lidar_corrector_embree_benchmark.cpp:127-135).  Built from oracle primitives only.
"""
import numpy as np

import oracle as orc


def directions(model):
    """rmagine SphericalModel::getDirection for every (vid, hid), from the C oracle (libm cosf/sinf)."""
    return orc.spherical_directions(model)


def dataset_from_ranges(model, ranges):
    """unpackMessage: point = dir * range; mask = 0 iff range < range.min or range > range.max."""
    r = np.asarray(ranges, dtype=np.float32).reshape(-1)
    pts = (directions(model) * r[:, None]).astype(np.float32)
    mask = np.where((r < np.float32(model.range.min)) | (r > np.float32(model.range.max)), 0, 1).astype(np.uint8)
    return pts, mask


def compute_cross_statistics_b(sim, ds_points, ds_mask, Tsb, T_bnew_bold, max_dist):
    """MICPSensor_::computeCrossStatistics (MICPSensor.hpp:159-184), stats in the base frame."""
    T_snew_sold = orc.tmult(orc.tmult(orc.tinv(Tsb), T_bnew_bold), Tsb)
    stats_s = orc.statistics_p2l_exact(T_snew_sold, ds_points, ds_mask, sim["points"], sim["normals"], sim["hits"], max_dist)
    return orc.cs_transform(Tsb, stats_s)


def correct_once(mesh, model, Tsb, Tbo, Tom, ds_points, ds_mask, n_iter, max_dist, adaptive_min=None,
                 convergence_progress=0.0, refind=False, nthreads=1):
    """Returns (T_onew_oold, merged stats in odom frame, list of T_onew_oold after each iteration)."""
    if adaptive_min is None:
        adaptive_min = max_dist
    md = orc.adaptive_max_dist(max_dist, adaptive_min, convergence_progress)
    ident = orc.transform()
    T_onew_oold = ident
    traj = []
    merged = orc.cs_identity()
    sim = None
    for i in range(n_iter):
        if sim is None or refind:
            Tom_cur = orc.tmult(Tom, T_onew_oold) if refind else Tom
            sim = mesh.simulate_spherical(model, Tsb, orc.tmult(Tom_cur, Tbo), bvh=True, nthreads=nthreads)
        T_delta = ident if refind else T_onew_oold
        T_bnew_bold = orc.tmult(orc.tmult(orc.tinv(Tbo), T_delta), Tbo)
        Cs_b = compute_cross_statistics_b(sim, ds_points, ds_mask, Tsb, T_bnew_bold, md)
        Cs_o = orc.cs_transform(Tbo, Cs_b)
        merged = orc.cs_merge(orc.cs_identity(), Cs_o)
        T_inner = orc.umeyama(merged)
        T_onew_oold = orc.tmult(T_onew_oold, T_inner)
        traj.append(T_onew_oold.copy())
    return T_onew_oold, merged, traj


def correct_batch(mesh, model, Tsb, Tbm, ds_points, ds_mask, max_dist, nthreads=1):
    """v1 SphereCorrector::correct: per pose raycast + reduce (Tpre = I) + Umeyama;
    Tdelta_b = Tsb * T_s * ~Tsb."""
    Tbm = np.asarray(Tbm, dtype=orc.TRANSFORM).reshape(-1)
    out = np.zeros(len(Tbm), dtype=orc.TRANSFORM)
    stats = np.zeros(len(Tbm), dtype=orc.CROSS_STATISTICS)
    ident = orc.transform()
    for i in range(len(Tbm)):
        sim = mesh.simulate_spherical(model, Tsb, Tbm[i], bvh=True, nthreads=nthreads)
        s = orc.statistics_p2l_exact(ident, ds_points, ds_mask, sim["points"], sim["normals"], sim["hits"], max_dist)
        Ts = orc.umeyama(s)
        out[i] = orc.tmult(orc.tmult(Tsb, Ts), orc.tinv(Tsb))
        stats[i] = s
    return out, stats
