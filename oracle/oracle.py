"""ctypes front-end of the CPU oracle (oracle/rmcl_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by anything under rmcl_amd/.
PARITY UNPINNED: see oracle/rmcl_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librmcl_oracle.so")

# numpy mirrors of the POD structs (identical layout to rmcl_amd.types)
VEC3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
QUAT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
TRANSFORM = np.dtype([("R", QUAT), ("t", VEC3), ("stamp", "<u4")])
CROSS_STATISTICS = np.dtype([("dataset_mean", VEC3), ("model_mean", VEC3),
                             ("covariance", "<f4", (9,)), ("n_meas", "<u4")])
GAUSSIAN1D = np.dtype([("mean", "<f4"), ("sigma", "<f4"), ("n_meas", "<u4")])
PARTICLE_ATTRIBUTES = np.dtype([("likelihood", GAUSSIAN1D), ("state_sigma", "<f4", (6,))])
RANGE_MEASUREMENT = np.dtype([("orig", VEC3), ("dir", VEC3), ("range", "<f4"), ("cov", "<f4", (9,))])
assert TRANSFORM.itemsize == 32 and CROSS_STATISTICS.itemsize == 64
assert PARTICLE_ATTRIBUTES.itemsize == 36 and RANGE_MEASUREMENT.itemsize == 64


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Quat(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Transform(C.Structure):
    _fields_ = [("R", Quat), ("t", Vec3), ("stamp", C.c_uint32)]


class DiscreteInterval(C.Structure):
    _fields_ = [("min", C.c_float), ("inc", C.c_float), ("size", C.c_uint32)]


class Interval(C.Structure):
    _fields_ = [("min", C.c_float), ("max", C.c_float)]


class SphericalModel(C.Structure):
    _fields_ = [("phi", DiscreteInterval), ("theta", DiscreteInterval), ("range", Interval)]


class CrossStatistics(C.Structure):
    _fields_ = [("dataset_mean", Vec3), ("model_mean", Vec3), ("covariance", C.c_float * 9),
                ("n_meas", C.c_uint32)]


class Gaussian1D(C.Structure):
    _fields_ = [("mean", C.c_float), ("sigma", C.c_float), ("n_meas", C.c_uint32)]


class Filter1D(C.Structure):
    _fields_ = [("skip_begin", C.c_uint32), ("skip_end", C.c_uint32), ("increment", C.c_uint32)]


class GladiatorConfig(C.Structure):
    _fields_ = [("min_noise_tx", C.c_float), ("min_noise_ty", C.c_float), ("min_noise_tz", C.c_float),
                ("min_noise_roll", C.c_float), ("min_noise_pitch", C.c_float), ("min_noise_yaw", C.c_float),
                ("likelihood_forget_per_meter", C.c_float), ("likelihood_forget_per_radian", C.c_float),
                ("trans_dist_metric", C.c_uint32)]


class LikelihoodStats(C.Structure):
    _fields_ = [("sum", C.c_float), ("max", C.c_float)]


class PFParams(C.Structure):
    _fields_ = [("dist_sigma", C.c_float), ("real_hit_sim_miss_error", C.c_float),
                ("real_miss_sim_hit_error", C.c_float), ("real_miss_sim_miss_error", C.c_float),
                ("sensor_range", Interval), ("max_n_meas", C.c_uint32),
                ("correspondence_type", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("rays", C.c_uint64)]


def build(force=False):
    """Compile the C restatement (gcc). Called by __graft_entry__.build()."""
    src = os.path.join(_HERE, "rmcl_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                    os.path.getmtime(os.path.join(_HERE, "rmcl_oracle.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "librmcl_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    with open("/proc/cpuinfo") as f:
        if " fma" not in f.read():
            raise RuntimeError("oracle is compiled with -mfma but this CPU has no FMA")
    L = C.CDLL(_LIB_PATH)
    vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
    L.orc_quat_mult.restype = Quat
    L.orc_quat_mult.argtypes = [Quat, Quat]
    L.orc_quat_rotate.restype = Vec3
    L.orc_quat_rotate.argtypes = [Quat, Vec3]
    L.orc_transform_mult.restype = Transform
    L.orc_transform_mult.argtypes = [Transform, Transform]
    L.orc_transform_inv.restype = Transform
    L.orc_transform_inv.argtypes = [Transform]
    L.orc_transform_apply.restype = Vec3
    L.orc_transform_apply.argtypes = [Transform, Vec3]
    L.orc_euler_to_quat.restype = Quat
    L.orc_euler_to_quat.argtypes = [f32, f32, f32]
    L.orc_mesh_create.restype = vp
    L.orc_mesh_create.argtypes = [vp, u32, vp, u32, u32]
    L.orc_mesh_destroy.argtypes = [vp]
    L.orc_mesh_num_nodes.restype = u32
    L.orc_mesh_num_nodes.argtypes = [vp]
    L.orc_mesh_face_normals.argtypes = [vp, vp]
    L.orc_intersect_brute.restype = i32
    L.orc_intersect_brute.argtypes = [vp, Vec3, Vec3, f32, f32, C.POINTER(f32), C.POINTER(u32)]
    L.orc_intersect_bvh.restype = i32
    L.orc_intersect_bvh.argtypes = [vp, Vec3, Vec3, f32, f32, C.POINTER(f32), C.POINTER(u32), vp]
    L.orc_trace_bvh4.restype = i32
    L.orc_trace_bvh4.argtypes = [vp, u32, vp, u32, Vec3, Vec3, f32, f32, C.POINTER(f32), C.POINTER(u32)]
    L.orc_mesh_tri_records.argtypes = [vp, vp]
    L.orc_spherical_directions.argtypes = [vp, vp]
    L.orc_simulate_spherical.restype = i32
    L.orc_simulate_spherical.argtypes = [vp, vp, vp, vp, u32, i32, i32,
                                         vp, vp, vp, vp, vp, vp]
    L.orc_simulate_pinhole.restype = i32
    L.orc_simulate_pinhole.argtypes = [vp, u32, u32, Interval, vp, vp, vp, vp, u32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.orc_simulate_ondn.restype = i32
    L.orc_simulate_ondn.argtypes = [vp, u32, u32, Interval, vp, vp, vp, vp, u32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.orc_pinhole_directions.argtypes = [u32, u32, vp, vp, vp]
    L.orc_simulate_o1dn.restype = i32
    L.orc_simulate_o1dn.argtypes = [vp, u32, u32, Interval, Vec3, vp, vp, vp, u32, i32, i32,
                                    vp, vp, vp, vp, vp, vp]
    L.orc_statistics_p2l_f32.argtypes = [vp, vp, vp, vp, vp, vp, u32, f32, vp]
    L.orc_statistics_p2l_f64.argtypes = [vp, vp, vp, vp, vp, vp, u32, f32, vp, vp]
    L.orc_statistics_p2l_fast.restype = None
    L.orc_statistics_p2l_fast.argtypes = [vp, vp, vp, vp, vp, vp, u32, f32, C.c_int, vp]
    L.orc_adaptive_max_dist.restype = f32
    L.orc_adaptive_max_dist.argtypes = [f32, f32, C.c_double]
    L.orc_cross_statistics_merge.restype = CrossStatistics
    L.orc_cross_statistics_merge.argtypes = [CrossStatistics, CrossStatistics]
    L.orc_cross_statistics_transform.restype = CrossStatistics
    L.orc_cross_statistics_transform.argtypes = [Transform, CrossStatistics]
    L.orc_umeyama_transform.restype = Transform
    L.orc_umeyama_transform.argtypes = [vp]
    L.orc_svd3.argtypes = [vp, vp, vp, vp]
    L.orc_gaussian1d_add.restype = Gaussian1D
    L.orc_gaussian1d_add.argtypes = [Gaussian1D, Gaussian1D]
    L.orc_evaluate_rcc.restype = f32
    L.orc_evaluate_rcc.argtypes = [vp, vp, vp, i32]
    L.orc_pf_update.restype = i32
    L.orc_pf_update.argtypes = [vp, vp, vp, u32, vp, u32, vp, vp, i32, i32, vp]
    L.orc_closest_point.restype = i32
    L.orc_closest_point.argtypes = [vp, Vec3, i32, C.POINTER(f32), C.POINTER(Vec3), C.POINTER(u32)]
    L.orc_cpc_find.argtypes = [vp, vp, vp, vp, u32, f32, i32, vp, vp, vp, vp, vp]
    L.orc_pf_motion_update.argtypes = [vp, vp, vp, u32, vp, C.c_double, u32, i32]
    L.orc_pointcloud2_unpack.argtypes = [vp, u32, u32, u32, u32, u32, u32, u32, u32, Filter1D, Filter1D, f32, f32,
                                         C.POINTER(u32), C.POINTER(u32), vp, vp, vp, vp, C.POINTER(u32)]
    L.orc_philox4x32_10.argtypes = [vp, vp, vp]
    L.orc_likelihood_stats_compute.restype = LikelihoodStats
    L.orc_likelihood_stats_compute.argtypes = [vp, u32]
    L.orc_quat_to_euler.argtypes = [Quat, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
    L.orc_gladiator_resample.argtypes = [vp, vp, u32, vp, vp, u32, u32, C.POINTER(GladiatorConfig), C.c_uint64, u32]
    L.orc_residual_resample.restype = u32
    L.orc_residual_resample.argtypes = [vp, vp, u32, vp, vp, u32, C.POINTER(GladiatorConfig), C.c_uint64, u32, C.c_uint64, C.POINTER(C.c_uint64)]
    _lib = L
    return L


# ---------------------------------------------------------------- helpers --
def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def transform(q=(0, 0, 0, 1), t=(0, 0, 0)):
    T = np.zeros((), dtype=TRANSFORM)
    T["R"]["x"], T["R"]["y"], T["R"]["z"], T["R"]["w"] = q
    T["t"]["x"], T["t"]["y"], T["t"]["z"] = t
    return T


def _ct_T(T):
    T = np.ascontiguousarray(T, dtype=TRANSFORM).reshape(())
    return Transform.from_buffer_copy(T.tobytes())


def _np_T(ct):
    return np.frombuffer(bytes(ct), dtype=TRANSFORM)[0].copy()


def euler_to_quat(roll, pitch, yaw):
    q = lib().orc_euler_to_quat(roll, pitch, yaw)
    return (q.x, q.y, q.z, q.w)


def transform_from_rpy(t, rpy):
    return transform(euler_to_quat(*rpy), t)


def tmult(a, b):
    return _np_T(lib().orc_transform_mult(_ct_T(a), _ct_T(b)))


def tinv(a):
    return _np_T(lib().orc_transform_inv(_ct_T(a)))


def tapply(T, p):
    v = lib().orc_transform_apply(_ct_T(T), Vec3(*[float(x) for x in p]))
    return np.array([v.x, v.y, v.z], dtype=np.float32)


def _ct_cs(s):
    s = np.ascontiguousarray(s, dtype=CROSS_STATISTICS).reshape(())
    return CrossStatistics.from_buffer_copy(s.tobytes())


def _np_cs(ct):
    return np.frombuffer(bytes(ct), dtype=CROSS_STATISTICS)[0].copy()


def cs_identity():
    return np.zeros((), dtype=CROSS_STATISTICS)


def cs_merge(a, b):
    return _np_cs(lib().orc_cross_statistics_merge(_ct_cs(a), _ct_cs(b)))


def cs_transform(T, s):
    return _np_cs(lib().orc_cross_statistics_transform(_ct_T(T), _ct_cs(s)))


def umeyama(s):
    s = np.ascontiguousarray(s, dtype=CROSS_STATISTICS).reshape(1)
    return _np_T(lib().orc_umeyama_transform(_p(s)))


def svd3(A):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(9)
    U = np.zeros(9); w = np.zeros(3); V = np.zeros(9)
    lib().orc_svd3(_p(A), _p(U), _p(w), _p(V))
    return U.reshape(3, 3), w, V.reshape(3, 3)


def adaptive_max_dist(max_dist, adaptive_min, p):
    return float(lib().orc_adaptive_max_dist(max_dist, adaptive_min, p))


def spherical_model(phi_min, phi_inc, phi_n, theta_min, theta_inc, theta_n, range_min, range_max):
    m = SphericalModel()
    m.phi.min, m.phi.inc, m.phi.size = phi_min, phi_inc, phi_n
    m.theta.min, m.theta.inc, m.theta.size = theta_min, theta_inc, theta_n
    m.range.min, m.range.max = range_min, range_max
    return m


def pf_params(dist_sigma=2.0, real_hit_sim_miss_error=100.0, real_miss_sim_hit_error=100.0,
              real_miss_sim_miss_error=0.0, range_min=0.05, range_max=80.0, max_n_meas=10000,
              correspondence_type=0):
    """Defaults: PCDSensorUpdaterEmbree.cpp:122-134."""
    p = PFParams()
    p.dist_sigma = dist_sigma
    p.real_hit_sim_miss_error = real_hit_sim_miss_error
    p.real_miss_sim_hit_error = real_miss_sim_hit_error
    p.real_miss_sim_miss_error = real_miss_sim_miss_error
    p.sensor_range.min, p.sensor_range.max = range_min, range_max
    p.max_n_meas = max_n_meas
    p.correspondence_type = correspondence_type
    return p


class Mesh:
    def __init__(self, verts, faces, max_leaf=4, build_bvh=True):
        """build_bvh=False: records only (ORC_MESH_NO_BVH) -- every query is brute force whatever `bvh=` says; for 10^7-face maps
        that are checked on a sample of rays against every triangle"""
        self.verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        self.h = lib().orc_mesh_create(_p(self.verts), len(self.verts), _p(self.faces), len(self.faces),
                                       int(max_leaf) | (0 if build_bvh else 0x80000000))
        if not self.h:
            raise ValueError("orc_mesh_create failed")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_mesh_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def num_nodes(self):
        return lib().orc_mesh_num_nodes(self.h)

    def face_normals(self):
        out = np.zeros((len(self.faces), 3), dtype=np.float32)
        lib().orc_mesh_face_normals(self.h, _p(out))
        return out

    def tri_records(self):
        out = np.zeros((len(self.faces), 15), dtype=np.float32)
        lib().orc_mesh_tri_records(self.h, _p(out))
        return out

    def intersect(self, O, D, tnear=0.0, tfar=np.inf, bvh=False):
        t = C.c_float(0)
        f = C.c_uint32(0)
        Ov = Vec3(*[float(x) for x in O])
        Dv = Vec3(*[float(x) for x in D])
        if bvh:
            r = lib().orc_intersect_bvh(self.h, Ov, Dv, tnear, tfar, C.byref(t), C.byref(f), None)
        else:
            r = lib().orc_intersect_brute(self.h, Ov, Dv, tnear, tfar, C.byref(t), C.byref(f))
        return (True, t.value, f.value) if r > 0 else (False, None, None)

    def _alloc(self, n, want):
        out = {}
        out["hits"] = np.zeros(n, dtype=np.uint8) if "hits" in want else None
        out["ranges"] = np.zeros(n, dtype=np.float32) if "ranges" in want else None
        out["points"] = np.zeros((n, 3), dtype=np.float32) if "points" in want else None
        out["normals"] = np.zeros((n, 3), dtype=np.float32) if "normals" in want else None
        out["face_ids"] = np.zeros(n, dtype=np.uint32) if "face_ids" in want else None
        return out

    def simulate_spherical(self, model, Tsb, Tbm, bvh=True, nthreads=1,
                           want=("hits", "ranges", "points", "normals", "face_ids"), counters=False, out=None):
        """out: reuse the output arrays of an earlier call of the same size (timing loops: fresh arrays cost
        first-touch page faults in every worker thread)."""
        Tbm = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        n = model.phi.size * model.theta.size * len(Tbm)
        if out is None:
            out = self._alloc(n, want)
        cnt = Counters()
        lib().orc_simulate_spherical(self.h, C.byref(model), _p(Tsb), _p(Tbm), len(Tbm), int(bvh), nthreads,
                                     _p(out["hits"]), _p(out["ranges"]), _p(out["points"]), _p(out["normals"]),
                                     _p(out["face_ids"]), C.addressof(cnt) if counters else None)
        if counters:
            out["counters"] = dict(nodes_visited=cnt.nodes_visited, tris_tested=cnt.tris_tested, rays=cnt.rays)
        return out

    def simulate_o1dn(self, width, height, range_min, range_max, orig, dirs, Tsb, Tbm, bvh=True, nthreads=1,
                      want=("hits", "ranges", "points", "normals", "face_ids"), counters=False):
        Tbm = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        assert len(dirs) == width * height
        n = width * height * len(Tbm)
        out = self._alloc(n, want)
        cnt = Counters()
        rng = Interval(range_min, range_max)
        lib().orc_simulate_o1dn(self.h, width, height, rng, Vec3(*[float(x) for x in orig]), _p(dirs),
                                _p(Tsb), _p(Tbm), len(Tbm), int(bvh), nthreads,
                                _p(out["hits"]), _p(out["ranges"]), _p(out["points"]), _p(out["normals"]),
                                _p(out["face_ids"]), C.addressof(cnt) if counters else None)
        if counters:
            out["counters"] = dict(nodes_visited=cnt.nodes_visited, tris_tested=cnt.tris_tested, rays=cnt.rays)
        return out

    def simulate_pinhole(self, width, height, range_min, range_max, f, c, Tsb, Tbm, bvh=True, nthreads=1,
                         want=("hits", "ranges", "points", "normals", "face_ids")):
        Tbm = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        fa, ca = np.asarray(f, dtype=np.float32), np.asarray(c, dtype=np.float32)
        out = self._alloc(width * height * len(Tbm), want)
        lib().orc_simulate_pinhole(self.h, width, height, Interval(range_min, range_max), _p(fa), _p(ca), _p(Tsb), _p(Tbm),
                                   len(Tbm), int(bvh), nthreads, _p(out["hits"]), _p(out["ranges"]), _p(out["points"]),
                                   _p(out["normals"]), _p(out["face_ids"]), None)
        return out

    def simulate_ondn(self, width, height, range_min, range_max, origs, dirs, Tsb, Tbm, bvh=True, nthreads=1,
                      want=("hits", "ranges", "points", "normals", "face_ids")):
        Tbm = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        origs = np.ascontiguousarray(origs, dtype=np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        assert len(origs) == len(dirs) == width * height
        out = self._alloc(width * height * len(Tbm), want)
        lib().orc_simulate_ondn(self.h, width, height, Interval(range_min, range_max), _p(origs), _p(dirs), _p(Tsb), _p(Tbm),
                                len(Tbm), int(bvh), nthreads, _p(out["hits"]), _p(out["ranges"]), _p(out["points"]),
                                _p(out["normals"]), _p(out["face_ids"]), None)
        return out

    def closest_point(self, P, bvh=False):
        d, cp, f = C.c_float(0), Vec3(), C.c_uint32(0)
        r = lib().orc_closest_point(self.h, Vec3(*[float(x) for x in P]), int(bvh), C.byref(d), C.byref(cp), C.byref(f))
        return (d.value, np.array([cp.x, cp.y, cp.z], np.float32), f.value) if r > 0 else None

    def cpc_find(self, Tsb, Tbm, dataset_points, max_dist, bvh=True):
        """CPCEmbree::find: dict(hits, ranges (= distance), points, normals, face_ids), sensor frame."""
        Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        Tbm = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(1)
        dp = np.ascontiguousarray(dataset_points, dtype=np.float32).reshape(-1, 3)
        out = self._alloc(len(dp), ("hits", "ranges", "points", "normals", "face_ids"))
        lib().orc_cpc_find(self.h, _p(Tsb), _p(Tbm), _p(dp), len(dp), float(max_dist), int(bvh), _p(out["hits"]),
                           _p(out["ranges"]), _p(out["points"]), _p(out["normals"]), _p(out["face_ids"]))
        return out

    def pf_motion_update(self, poses, attrs, T_bnew_bold, forget_rate, collision=True, max_n_meas=10000, bvh=True):
        """In-place TFMotionUpdaterCPU inner loop (poses, attrs are modified)."""
        assert poses.dtype == TRANSFORM and attrs.dtype == PARTICLE_ATTRIBUTES
        T = np.ascontiguousarray(T_bnew_bold, dtype=TRANSFORM).reshape(1)
        lib().orc_pf_motion_update(self.h if collision else None, _p(poses), _p(attrs), len(poses), _p(T), float(forget_rate),
                                   int(max_n_meas), int(bvh))

    def pf_update(self, poses, attrs, beams, Tsb, params, bvh=True, nthreads=1, want_errors=False):
        """In-place update of attrs (returns errors if asked)."""
        poses = np.ascontiguousarray(poses, dtype=TRANSFORM).reshape(-1)
        assert attrs.dtype == PARTICLE_ATTRIBUTES and attrs.flags.c_contiguous
        beams = np.ascontiguousarray(beams, dtype=RANGE_MEASUREMENT).reshape(-1)
        Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        err = np.zeros((len(poses), len(beams)), dtype=np.float32) if want_errors else None
        lib().orc_pf_update(self.h, _p(poses), _p(attrs), len(poses), _p(beams), len(beams), _p(Tsb),
                            C.byref(params), int(bvh), nthreads, _p(err))
        return err


def statistics_p2l(Tpre, dataset_points, dataset_mask, model_points, model_normals, model_mask, max_dist):
    Tpre = np.ascontiguousarray(Tpre, dtype=TRANSFORM).reshape(1)
    dp = np.ascontiguousarray(dataset_points, dtype=np.float32).reshape(-1, 3)
    mp = np.ascontiguousarray(model_points, dtype=np.float32).reshape(-1, 3)
    mn = np.ascontiguousarray(model_normals, dtype=np.float32).reshape(-1, 3)
    dm = None if dataset_mask is None else np.ascontiguousarray(dataset_mask, dtype=np.uint8)
    mm = None if model_mask is None else np.ascontiguousarray(model_mask, dtype=np.uint8)
    out = np.zeros(1, dtype=CROSS_STATISTICS)
    lib().orc_statistics_p2l_f32(_p(Tpre), _p(dp), _p(dm), _p(mp), _p(mn), _p(mm), len(dp), max_dist, _p(out))
    return out[0].copy()


def statistics_p2l_fast(Tpre, dataset_points, dataset_mask, model_points, model_normals, model_mask, max_dist, nthreads=1):
    """one pass of raw double sums on `nthreads` workers (bench.py's cpu_baseline form; inputs are used as they are when already
    contiguous float32 / uint8)"""
    Tpre = np.ascontiguousarray(Tpre, dtype=TRANSFORM).reshape(1)
    dp = np.ascontiguousarray(dataset_points, dtype=np.float32).reshape(-1, 3)
    mp = np.ascontiguousarray(model_points, dtype=np.float32).reshape(-1, 3)
    mn = np.ascontiguousarray(model_normals, dtype=np.float32).reshape(-1, 3)
    dm = None if dataset_mask is None else np.ascontiguousarray(dataset_mask, dtype=np.uint8)
    mm = None if model_mask is None else np.ascontiguousarray(model_mask, dtype=np.uint8)
    out = np.zeros(1, dtype=CROSS_STATISTICS)
    lib().orc_statistics_p2l_fast(_p(Tpre), _p(dp), _p(dm), _p(mp), _p(mn), _p(mm), len(dp), max_dist, int(nthreads), _p(out))
    return out[0].copy()


def statistics_p2l_f64(Tpre, dataset_points, dataset_mask, model_points, model_normals, model_mask, max_dist):
    Tpre = np.ascontiguousarray(Tpre, dtype=TRANSFORM).reshape(1)
    dp = np.ascontiguousarray(dataset_points, dtype=np.float32).reshape(-1, 3)
    mp = np.ascontiguousarray(model_points, dtype=np.float32).reshape(-1, 3)
    mn = np.ascontiguousarray(model_normals, dtype=np.float32).reshape(-1, 3)
    dm = None if dataset_mask is None else np.ascontiguousarray(dataset_mask, dtype=np.uint8)
    mm = None if model_mask is None else np.ascontiguousarray(model_mask, dtype=np.uint8)
    out = np.zeros(15, dtype=np.float64)
    n = np.zeros(1, dtype=np.uint32)
    lib().orc_statistics_p2l_f64(_p(Tpre), _p(dp), _p(dm), _p(mp), _p(mn), _p(mm), len(dp), max_dist, _p(out), _p(n))
    return dict(dataset_mean=out[0:3].copy(), model_mean=out[3:6].copy(), covariance=out[6:15].reshape(3, 3).copy(),
                n_meas=int(n[0]))


def gaussian1d_add(a, b):
    r = lib().orc_gaussian1d_add(Gaussian1D(*a), Gaussian1D(*b))
    return (r.mean, r.sigma, r.n_meas)


def trace_bvh4(nodes, tris, O, D, tnear=0.0, tfar=np.inf):
    """Closest hit through the PRODUCT's exported BVH4 arrays, with the oracle's intersector."""
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    tris = np.ascontiguousarray(tris, dtype=np.uint32)
    t = C.c_float(0)
    f = C.c_uint32(0)
    r = lib().orc_trace_bvh4(_p(nodes), nodes.size // 32, _p(tris), tris.size // 16, Vec3(*[float(x) for x in O]),
                             Vec3(*[float(x) for x in D]), tnear, tfar, C.byref(t), C.byref(f))
    if r < 0:
        raise RuntimeError("malformed BVH4 (code %d)" % r)
    return (True, t.value, f.value) if r > 0 else (False, None, None)


def spherical_directions(model):
    """(H*W, 3) float32 sensor-frame directions, exactly the values the simulate restatement uses."""
    out = np.zeros((model.phi.size * model.theta.size, 3), dtype=np.float32)
    lib().orc_spherical_directions(C.byref(model), _p(out))
    return out


def statistics_p2l_exact(Tpre, dataset_points, dataset_mask, model_points, model_normals, model_mask, max_dist):
    """statistics_p2l evaluated in double (two-pass) and rounded once to the f32 CrossStatistics struct:
    the order-independent value of the reference's formula on the given f32 inputs.  This is the parity
    authority for reductions; statistics_p2l() (f32 sequential merge, the single-thread order of the
    reference) drifts by ~1e-4 over 10^5 elements, as do the reference's own OpenMP / CUDA orders."""
    r = statistics_p2l_f64(Tpre, dataset_points, dataset_mask, model_points, model_normals, model_mask, max_dist)
    s = np.zeros((), dtype=CROSS_STATISTICS)
    for i, k in enumerate("xyz"):
        s["dataset_mean"][k] = r["dataset_mean"][i]
        s["model_mean"][k] = r["model_mean"][i]
    s["covariance"] = r["covariance"].reshape(9)
    s["n_meas"] = r["n_meas"]
    return s


def pinhole_directions(width, height, f, c):
    """(H*W, 3) sensor-frame directions of a pinhole model (rmagine PinholeModel::getDirection)."""
    fa, ca = np.asarray(f, dtype=np.float32), np.asarray(c, dtype=np.float32)
    out = np.zeros((width * height, 3), dtype=np.float32)
    lib().orc_pinhole_directions(width, height, _p(fa), _p(ca), _p(out))
    return out


# ------------------------------------------------------------ resampling --
def gladiator_config(min_noise_tx=0.03, min_noise_ty=0.03, min_noise_tz=0.0, min_noise_roll=0.0, min_noise_pitch=0.0,
                     min_noise_yaw=0.01, likelihood_forget_per_meter=0.3, likelihood_forget_per_radian=0.2,
                     trans_dist_metric=0):
    """GladiatorResamplerGPU::updateParams defaults (GladiatorResamplerGPU.cpp:34-44)."""
    return GladiatorConfig(min_noise_tx, min_noise_ty, min_noise_tz, min_noise_roll, min_noise_pitch, min_noise_yaw,
                           likelihood_forget_per_meter, likelihood_forget_per_radian, trans_dist_metric)


def philox4x32_10(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def likelihood_stats(attrs):
    assert attrs.dtype == PARTICLE_ATTRIBUTES
    r = lib().orc_likelihood_stats_compute(_p(attrs), len(attrs))
    return {"sum": r.sum, "max": r.max}


def quat_to_euler(q):
    r, p, y = C.c_float(), C.c_float(), C.c_float()
    xyzw = [float(v) for v in q] if isinstance(q, (tuple, list)) else [float(q[k]) for k in "xyzw"]
    lib().orc_quat_to_euler(Quat(*xyzw), C.byref(r), C.byref(p), C.byref(y))
    return r.value, p.value, y.value


def gladiator_resample(poses, attrs, cfg, seed, step, first=0, count=None):
    """returns (poses_new, attrs_new) of champions first..first+count-1."""
    poses = np.ascontiguousarray(poses, dtype=TRANSFORM).reshape(-1)
    assert attrs.dtype == PARTICLE_ATTRIBUTES and len(attrs) == len(poses)
    if count is None:
        count = len(poses) - first
    pn, an = np.zeros(count, TRANSFORM), np.zeros(count, PARTICLE_ATTRIBUTES)
    lib().orc_gladiator_resample(_p(poses), _p(attrs), len(poses), _p(pn), _p(an), int(first), int(count),
                                 C.byref(cfg), int(seed), int(step))
    return pn, an


def residual_resample(poses, attrs, cfg, seed, step, n_new=None, max_draws=None):
    """ResidualResamplerCPU::update: (poses_new, attrs_new, n_filled, n_draws)"""
    n_new = len(poses) if n_new is None else int(n_new)
    max_draws = (1 << 40) if max_draws is None else int(max_draws)
    pn, an = np.zeros(n_new, TRANSFORM), np.zeros(n_new, PARTICLE_ATTRIBUTES)
    nd = C.c_uint64(0)
    filled = lib().orc_residual_resample(_p(poses), _p(attrs), len(poses), _p(pn), _p(an), n_new, C.byref(cfg), int(seed), int(step),
                                         max_draws, C.byref(nd))
    return pn, an, int(filled), int(nd.value)


# ------------------------------------------------------------ wire formats --
def pointcloud2_unpack(data, width, height, point_step, row_step, off_x, off_y, off_z, datatype=7,
                       filter_h=(0, 0, 1), filter_w=(0, 0, 1), range_min=0.0, range_max=1e30):
    """PointCloud2 bytes -> {width, height, dirs, ranges, points, mask, n_valid} (O1Dn model + dataset)."""
    buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1))
    fh, fw = Filter1D(*filter_h), Filter1D(*filter_w)
    ow, oh, nv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    args = [_p(buf), width, height, point_step, row_step, off_x, off_y, off_z, datatype, fh, fw, range_min, range_max]
    rc = lib().orc_pointcloud2_unpack(*args, C.byref(ow), C.byref(oh), None, None, None, None, C.byref(nv))
    if rc != 0:
        raise ValueError("pointcloud2_unpack: rc %d" % rc)
    n = ow.value * oh.value
    dirs, pts = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    rng, mask = np.zeros(n, np.float32), np.zeros(n, np.uint8)
    lib().orc_pointcloud2_unpack(*args, C.byref(ow), C.byref(oh), _p(dirs), _p(rng), _p(pts), _p(mask), C.byref(nv))
    return {"width": ow.value, "height": oh.value, "dirs": dirs, "ranges": rng, "points": pts, "mask": mask,
            "n_valid": nv.value}


def sample_beams_pointcloud2(data, width, height, point_step, row_step, off_x, off_y, off_z, samples, seed, datatype=7):
    """PCDSensorUpdaterEmbree.cpp:276-327 with the pinned stream (MT19937(seed), index = draw % n_points)."""
    buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8)) if isinstance(data, (bytes, bytearray, memoryview)) \
        else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    out = np.zeros(int(samples), dtype=RANGE_MEASUREMENT)
    n = C.c_uint32(0)
    L = lib()
    L.orc_sample_beams_pointcloud2.argtypes = [C.c_void_p] + [C.c_uint32] * 10 + [C.c_void_p, C.POINTER(C.c_uint32)]
    rc = L.orc_sample_beams_pointcloud2(buf.ctypes.data, width, height, point_step, row_step, off_x, off_y, off_z, datatype,
                                        samples, seed & 0xFFFFFFFF, out.ctypes.data, C.byref(n))
    assert rc == 0
    return out[: n.value].copy()


def mt19937_draw(seed, n_skip=0):
    L = lib()
    L.orc_mt19937_draw.restype = C.c_uint32
    L.orc_mt19937_draw.argtypes = [C.c_uint32, C.c_uint32]
    return int(L.orc_mt19937_draw(seed & 0xFFFFFFFF, n_skip))


def estimate_stats(poses, attrs, max_induction_particles=None):
    """RmclNode::estimateStats (rmcl_ros/src/nodes/rmcl_localization.cpp:642-731) restated in double precision numpy.
    rm::markley_mean / rm::covariance are EXTERNAL (rmagine): restated as the published algorithms -- Markley et al. 2007:
    the mean rotation is the eigenvector of the largest eigenvalue of sum w q q^T, the mean translation sum w t; the 6x6
    covariance is sum w d d^T with d = (translation, roll, pitch, yaw) of ~Tbm * T_i (parity unpinned)."""
    n = len(poses) if max_induction_particles is None else min(len(poses), int(max_induction_particles))
    P, A = poses[:n], attrs[:n]
    L = A["likelihood"]["mean"].astype(np.float64)
    L_sum, L_n = L.sum(), float(n)
    L_mean = L_sum / L_n
    t = np.stack([P["t"][k] for k in "xyz"], 1).astype(np.float64)
    q = np.stack([P["R"][k] for k in "xyzw"], 1).astype(np.float64)
    out = {"likelihood": {"mean": L_mean, "sigma": float(np.sqrt(max((L * L).sum() / L_n - L_mean * L_mean, 0.0))),
                          "min": float(L.min()), "max": float(max(L.max(), 0.0))},
           "trans_bb_min": t.min(0), "trans_bb_max": t.max(0), "nparticles": n}
    w = L / L_sum
    M = (q * w[:, None]).T @ q
    ev, evec = np.linalg.eigh(M)
    qm = evec[:, -1]
    if qm[3] < 0:
        qm = -qm
    tm = (t * w[:, None]).sum(0)
    Tbm = transform(qm, tm)
    out["pose"] = Tbm
    Tmb = tinv(Tbm)
    d = np.zeros((n, 6))
    for i in range(n):
        Td = tmult(Tmb, P[i])
        d[i, :3] = [Td["t"][k] for k in "xyz"]
        d[i, 3:] = quat_to_euler(Td["R"])
    out["covariance"] = (d * w[:, None]).T @ d
    return out
