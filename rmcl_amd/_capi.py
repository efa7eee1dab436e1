"""ctypes binding of librmclhip.so (include/rmclhip.h).

The library is the product; this module only loads it.  It fails loudly when the
shared object is missing -- there is no Python/CPU fallback of the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librmclhip.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED = range(6)


class RmclHipError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (micp_localization.cpp:613)."""

    def __init__(self, status, msg):
        super().__init__("rmclhip status %d: %s" % (status, msg))
        self.status = status


class NoDeviceError(RmclHipError):
    pass


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Interval(C.Structure):
    _fields_ = [("min", C.c_float), ("max", C.c_float)]


class DiscreteInterval(C.Structure):
    _fields_ = [("min", C.c_float), ("inc", C.c_float), ("size", C.c_uint32)]


class SphericalModel(C.Structure):
    _fields_ = [("phi", DiscreteInterval), ("theta", DiscreteInterval), ("range", Interval)]


class PFParams(C.Structure):
    _fields_ = [("dist_sigma", C.c_float), ("real_hit_sim_miss_error", C.c_float),
                ("real_miss_sim_hit_error", C.c_float), ("real_miss_sim_miss_error", C.c_float),
                ("sensor_range", Interval), ("max_n_meas", C.c_uint32),
                ("correspondence_type", C.c_uint32)]


class MicpFastInfo(C.Structure):
    _fields_ = [("attempts", C.c_uint32), ("done", C.c_uint32), ("cap_exits", C.c_uint32), ("overflows", C.c_uint32),
                ("last_code", C.c_uint32), ("last_uncertain", C.c_uint32), ("last_rho", C.c_float), ("last_tau", C.c_float),
                ("rho_cap", C.c_float), ("tau_cap", C.c_float), ("last_setup_clocks", C.c_uint32), ("last_loop_clocks", C.c_uint32),
                ("host_loops", C.c_uint32)]


class CcsInfo(C.Structure):
    _fields_ = [("calls", C.c_uint32), ("from_moments", C.c_uint32), ("passes", C.c_uint32), ("speculative_finds", C.c_uint32)]


class PointCloud2Layout(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("point_step", C.c_uint32), ("row_step", C.c_uint32),
                ("offset_x", C.c_uint32), ("offset_y", C.c_uint32), ("offset_z", C.c_uint32), ("datatype", C.c_uint32)]


class Filter1D(C.Structure):
    _fields_ = [("skip_begin", C.c_uint32), ("skip_end", C.c_uint32), ("increment", C.c_uint32)]


class GladiatorConfig(C.Structure):
    _fields_ = [("min_noise_tx", C.c_float), ("min_noise_ty", C.c_float), ("min_noise_tz", C.c_float),
                ("min_noise_roll", C.c_float), ("min_noise_pitch", C.c_float), ("min_noise_yaw", C.c_float),
                ("likelihood_forget_per_meter", C.c_float), ("likelihood_forget_per_radian", C.c_float),
                ("trans_dist_metric", C.c_uint32)]


class LikelihoodStats(C.Structure):
    _fields_ = [("sum", C.c_float), ("max", C.c_float)]


class PoseEstimate(C.Structure):
    _fields_ = [("pose", C.c_float * 8), ("covariance", C.c_double * 36), ("likelihood_mean", C.c_double),
                ("likelihood_sigma", C.c_double), ("likelihood_min", C.c_double), ("likelihood_max", C.c_double),
                ("trans_bb_min", C.c_float * 3), ("trans_bb_max", C.c_float * 3), ("n_particles", C.c_uint32),
                ("reserved", C.c_uint32)]


class MapInfo(C.Structure):
    _fields_ = [("n_faces", C.c_uint32), ("n_vertices", C.c_uint32), ("n_nodes", C.c_uint32),
                ("n_tri_records", C.c_uint32), ("max_depth", C.c_uint32), ("stack_need", C.c_uint32),
                ("device_bytes", C.c_uint64), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("height_fallbacks", C.c_uint32), ("guarded_nodes", C.c_uint32), ("spatial_splits", C.c_uint32), ("reserved", C.c_uint32)]


class BundleViews(C.Structure):
    """rmclhip_bundle_views: a rmagine Bundle over caller-owned device memory, one nullable pointer per attribute"""
    _fields_ = [("hits_dev", C.c_void_p), ("ranges_dev", C.c_void_p), ("points_xyz_dev", C.c_void_p),
                ("normals_xyz_dev", C.c_void_p), ("face_ids_dev", C.c_void_p)]


OUT_HITS, OUT_RANGES, OUT_POINTS, OUT_NORMALS, OUT_FACE_IDS = 1, 2, 4, 8, 16
OUT_ALL = 31
OUT_MICP = OUT_HITS | OUT_POINTS | OUT_NORMALS   # Correspondences_::model_buffers_ (Correspondences.hpp:81-85)


class Mesh(C.Structure):
    _fields_ = [("vertices_xyz", C.c_void_p), ("faces_ijk", C.c_void_p), ("n_vertices", C.c_uint32), ("n_faces", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("mesh", C.c_uint32), ("reserved", C.c_uint32 * 3)]


# name -> (restype, argtypes); every symbol declared in include/rmclhip.h
_vp, _u32, _f32, _i32, _sz, _dbl = C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_size_t, C.c_double
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "rmclhip_last_error": (C.c_char_p, []),
    "rmclhip_version": (C.c_char_p, []),
    "rmclhip_ctx_create": (_i32, [_i32, _pp]),
    "rmclhip_ctx_destroy": (None, [_vp]),
    "rmclhip_ctx_device_name": (_i32, [_vp, C.c_char_p, _sz]),
    "rmclhip_map_create": (_i32, [_vp, _vp, _u32, _vp, _u32, _pp]),
    "rmclhip_map_create_scene": (_i32, [_vp, _vp, _u32, _vp, _u32, _pp]),
    "rmclhip_map_scene_instances": (_i32, [_vp, _vp, _sz, C.POINTER(_u32)]),
    "rmclhip_map_scene_locate": (_i32, [_vp, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "rmclhip_scene_flatten_host": (_i32, [_vp, _u32, _vp, _u32, _vp, _sz, _vp, _sz, _vp, _sz, C.POINTER(_u32),
                                           C.POINTER(_u32)]),
    "rmclhip_map_retain": (_i32, [_vp]),
    "rmclhip_map_release": (None, [_vp]),
    "rmclhip_map_get_info": (_i32, [_vp, C.POINTER(MapInfo)]),
    "rmclhip_bvh_build_host": (_i32, [_vp, _u32, _vp, _u32, C.POINTER(MapInfo), _vp, _sz, _vp, _sz]),
    "rmclhip_bvh_build_host_quantised": (_i32, [_vp, _u32, _vp, _u32, _vp, _sz]),
    "rmclhip_bvh_build_host_pf": (_i32, [_vp, _u32, _vp, _u32, C.POINTER(MapInfo), _vp, _sz, _vp, _sz]),
    "rmclhip_rcc_create": (_i32, [_vp, _vp, _pp]),
    "rmclhip_rcc_destroy": (None, [_vp]),
    "rmclhip_rcc_set_tsb": (_i32, [_vp, _vp]),
    "rmclhip_rcc_set_model_spherical": (_i32, [_vp, C.POINTER(SphericalModel)]),
    "rmclhip_rcc_set_model_o1dn": (_i32, [_vp, _u32, _u32, Interval, Vec3, _vp]),
    "rmclhip_rcc_set_model_pinhole": (_i32, [_vp, _u32, _u32, Interval, _f32, _f32, _f32, _f32]),
    "rmclhip_rcc_set_model_ondn": (_i32, [_vp, _u32, _u32, Interval, _vp, _vp]),
    "rmclhip_rcc_set_params": (_i32, [_vp, _f32, _f32]),
    "rmclhip_rcc_set_dataset": (_i32, [_vp, _vp, _vp, _u32, _i32]),
    "rmclhip_rcc_set_dataset_view": (_i32, [_vp, _vp, _vp, _u32]),
    "rmclhip_rcc_set_dataset_from_ranges": (_i32, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "rmclhip_rcc_find": (_i32, [_vp, _vp]),
    "rmclhip_rcc_find_async": (_i32, [_vp, _vp]),
    "rmclhip_rcc_sync": (_i32, [_vp]),
    "rmclhip_rcc_find_cpc": (_i32, [_vp, _vp]),
    "rmclhip_rcc_set_input_pointcloud2": (_i32, [_vp, _vp, _sz, C.POINTER(PointCloud2Layout), C.POINTER(Filter1D),
                                                  C.POINTER(Filter1D), Interval, _i32, C.POINTER(_u32),
                                                  C.POINTER(_u32), C.POINTER(_u32)]),
    "rmclhip_rcc_compute_cross_statistics": (_i32, [_vp, _vp, _dbl, _vp]),
    "rmclhip_rcc_download": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rmclhip_rcc_device_views": (_i32, [_vp, _pp, _pp, _pp, _pp, _pp, C.POINTER(_u32)]),
    "rmclhip_rcc_set_outputs": (_i32, [_vp, _u32]),
    "rmclhip_rcc_get_outputs": (_i32, [_vp, C.POINTER(_u32)]),
    "rmclhip_rcc_simulate": (_i32, [_vp, _vp, _u32, _i32, C.POINTER(BundleViews)]),
    "rmclhip_rcc_simulate_async": (_i32, [_vp, _vp, _u32, _i32, C.POINTER(BundleViews)]),
    "rmclhip_statistics_p2l": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _f32, _vp]),
    "rmclhip_rcc_correct_once": (_i32, [_vp, _vp, _vp, _u32, _dbl, _i32, _vp, _vp]),
    "rmclhip_micp_correct_once": (_i32, [_vp, _u32, _vp, _vp, _vp, _u32, _dbl, _vp, _vp]),
    "rmclhip_rcc_correct_batch": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "rmclhip_rcc_sharded_create": (_i32, [_vp, _u32, _vp, _u32, _vp, _u32, _pp]),
    "rmclhip_rcc_sharded_destroy": (None, [_vp]),
    "rmclhip_rcc_sharded_size": (_u32, [_vp]),
    "rmclhip_rcc_sharded_replica": (_i32, [_vp, _u32, _pp]),
    "rmclhip_rcc_sharded_correct_batch": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "rmclhip_rcc_last_kernel_ms": (_i32, [_vp, C.POINTER(_f32), C.POINTER(_f32)]),
    "rmclhip_rcc_set_kernel_timing": (_i32, [_vp, _i32]),
    "rmclhip_rcc_time_find_sync": (_i32, [_vp, _vp, _u32, C.POINTER(_f32)]),
    "rmclhip_rcc_time_find": (_i32, [_vp, _vp, _u32, C.POINTER(_f32)]),
    "rmclhip_rcc_autotune": (_i32, [_vp, _vp, C.POINTER(_i32), C.POINTER(_f32)]),
    "rmclhip_rcc_autotune_batch": (_i32, [_vp, _vp, _u32, C.POINTER(_i32), C.POINTER(_f32)]),
    "rmclhip_rcc_time_reduce": (_i32, [_vp, _vp, _u32, C.POINTER(_f32)]),
    "rmclhip_rcc_time_correct_once": (_i32, [_vp, _vp, _vp, _u32, _dbl, _i32, _u32, C.POINTER(_f32)]),
    "rmclhip_rcc_time_caller_loop": (_i32, [_vp, _vp, _vp, _u32, _dbl, _u32, _vp, _vp, C.POINTER(_f32)]),
    "rmclhip_rcc_set_variant": (_i32, [_vp, _i32]),
    "rmclhip_rcc_find_variant": (_i32, [_vp, _u32, C.POINTER(_i32)]),
    "rmclhip_rcc_set_descent": (_i32, [_vp, _u32, _u32]),
    "rmclhip_rcc_set_batch_order": (_i32, [_vp, C.c_int]),
    "rmclhip_rcc_set_micp_fast": (_i32, [_vp, _i32]),
    "rmclhip_rcc_micp_fast_info": (_i32, [_vp, C.POINTER(MicpFastInfo)]),
    "rmclhip_rcc_ccs_info": (_i32, [_vp, C.POINTER(CcsInfo)]),
    "rmclhip_host_moment_statistics": (_i32, [_vp, _vp, _vp, _vp, _u32, _f32, _f32, _f32, _f32, _vp, _f32, _vp, C.POINTER(_u32),
                                               C.POINTER(_i32)]),
    "rmclhip_debug_wave_clock": (_i32, [_vp, _vp, _vp, _sz, C.POINTER(_u32)]),
    "rmclhip_debug_micp_moments": (_i32, [_vp, _vp, C.POINTER(_u32), C.POINTER(C.c_uint64)]),
    "rmclhip_debug_probe_find": (_i32, [_vp, _vp, _i32, _vp, _sz, C.POINTER(_u32)]),
    "rmclhip_rcc_find_batch": (_i32, [_vp, _vp, _u32]),
    "rmclhip_rcc_time_find_batch": (_i32, [_vp, _vp, _u32, _u32, C.POINTER(_f32)]),
    "rmclhip_umeyama_transform": (_i32, [_vp, _vp]),
    "rmclhip_cross_statistics_merge": (_i32, [_vp, _vp, _vp]),
    "rmclhip_cross_statistics_transform": (_i32, [_vp, _vp, _vp]),
    "rmclhip_transform_mult": (_i32, [_vp, _vp, _vp]),
    "rmclhip_transform_inv": (_i32, [_vp, _vp]),
    "rmclhip_pf_create": (_i32, [_vp, _vp, _pp]),
    "rmclhip_pf_destroy": (None, [_vp]),
    "rmclhip_pf_set_params": (_i32, [_vp, C.POINTER(PFParams)]),
    "rmclhip_pf_update": (_i32, [_vp, _vp, _vp, _u32, _vp, _u32, _vp]),
    "rmclhip_pf_update_async": (_i32, [_vp, _vp, _vp, _u32, _vp, _u32, _vp]),
    "rmclhip_pf_sync": (_i32, [_vp]),
    "rmclhip_pf_set_error_output": (_i32, [_vp, _vp]),
    "rmclhip_pf_motion_update": (_i32, [_vp, _vp, _vp, _u32, _vp, _dbl, _i32]),
    "rmclhip_pf_extract_weights": (_i32, [_vp, _vp, _u32, _vp]),
    "rmclhip_pf_time_update": (_i32, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _u32, C.POINTER(_f32)]),
    "rmclhip_pf_time_update_unfused": (_i32, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _i32, _u32, C.POINTER(_f32)]),
    "rmclhip_pf_set_variant": (_i32, [_vp, _i32]),
    "rmclhip_pf_set_schedule": (_i32, [_vp, _u32, _u32]),
    "rmclhip_pf_set_mapping": (_i32, [_vp, _i32, _u32, _vp, _u32]),
    "rmclhip_ctx_set_wait_mode": (_i32, [_vp, _i32]),
    "rmclhip_rcc_set_cpc_tracking": (_i32, [_vp, _i32]),
    "rmclhip_rcc_set_cpc_bounded": (_i32, [_vp, _i32]),
    "rmclhip_rcc_set_cpc_grid": (_i32, [_vp, _i32]),
    "rmclhip_pf_sample_beams_pointcloud2": (_i32, [_vp, _sz, C.POINTER(PointCloud2Layout), _u32, C.c_uint64, _vp, C.POINTER(_u32)]),
    "rmclhip_resampler_create": (_i32, [_vp, _pp]),
    "rmclhip_resampler_destroy": (None, [_vp]),
    "rmclhip_resampler_compute_stats": (_i32, [_vp, _vp, _u32, C.POINTER(LikelihoodStats)]),
    "rmclhip_resampler_compute_stats_weights": (_i32, [_vp, _vp, _u32, C.POINTER(LikelihoodStats)]),
    "rmclhip_resampler_gladiator": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _u32, C.POINTER(GladiatorConfig),
                                            C.c_uint64, _u32]),
    "rmclhip_resampler_residual": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _u32, _u32, C.POINTER(GladiatorConfig),
                                           C.c_uint64, _u32, C.POINTER(C.c_uint64)]),
    "rmclhip_comm_create": (_i32, [_vp, _u32, _pp]),
    "rmclhip_comm_create_loopback": (_i32, [_vp, _u32, _pp]),
    "rmclhip_comm_collective_ranks": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_i32)]),
    "rmclhip_comm_loopback_set_reduce_rotation": (_i32, [_vp, _u32]),
    "rmclhip_debug_tag_retries": (_i32, [C.POINTER(C.c_ulonglong)]),
    "rmclhip_debug_trace": (_i32, [_i32, _vp, _sz]),
    "rmclhip_comm_destroy": (None, [_vp]),
    "rmclhip_comm_size": (_u32, [_vp]),
    "rmclhip_shard_bounds": (None, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "rmclhip_pf_sharded_create": (_i32, [_vp, _vp, _u32, _vp, _u32, _pp]),
    "rmclhip_pf_sharded_destroy": (None, [_vp]),
    "rmclhip_pf_sharded_set_params": (_i32, [_vp, C.POINTER(PFParams)]),
    "rmclhip_pf_sharded_set_particles": (_i32, [_vp, _vp, _vp, _u32]),
    "rmclhip_pf_sharded_download": (_i32, [_vp, _vp, _vp]),
    "rmclhip_pf_update_sharded": (_i32, [_vp, _vp, _u32, _vp]),
    "rmclhip_pf_sharded_motion_update": (_i32, [_vp, _vp, _dbl, _i32]),
    "rmclhip_pf_sharded_step": (_i32, [_vp, _vp, _dbl, _i32, _vp, _u32, _vp, _i32, C.POINTER(GladiatorConfig), C.c_uint64, _u32,
                                       C.POINTER(LikelihoodStats)]),
    "rmclhip_pf_allgather_weights": (_i32, [_vp]),
    "rmclhip_pf_sharded_get_weights": (_i32, [_vp, _u32, _vp]),
    "rmclhip_pf_allreduce_stats": (_i32, [_vp, C.POINTER(LikelihoodStats)]),
    "rmclhip_pf_allreduce_pose_estimate": (_i32, [_vp, _u32, C.POINTER(PoseEstimate)]),
    "rmclhip_pf_sharded_resample": (_i32, [_vp, C.POINTER(GladiatorConfig), C.c_uint64, _u32]),
    "rmclhip_pf_sharded_resample_residual": (_i32, [_vp, C.POINTER(GladiatorConfig), C.c_uint64, _u32]),
    "rmclhip_malloc": (_i32, [_vp, _sz, _pp]),
    "rmclhip_free": (_i32, [_vp, _vp]),
    "rmclhip_memcpy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "rmclhip_memcpy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
}

_lib = None


def lib():
    """Load librmclhip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librmclhip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rmcl_amd/csrc`. rmcl_amd has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


LAB_PATH = os.path.join(os.path.dirname(LIB_PATH), "librmclhip_lab.so")
_lab = None


def load_lab():
    """Load librmclhip_lab.so (EXPERIMENTS: rejected traversal kinds, probes, older particle-filter kernels; see
    include/rmclhip_lab.h).  Its static initialiser registers the experiments' launchers with librmclhip.so.  Only tools/
    and the `lab` test group call this; the product never does."""
    global _lab
    if _lab is not None:
        return _lab
    lib()
    if not os.path.exists(LAB_PATH):
        raise ImportError("librmclhip_lab.so is missing (%s): `make -C rmcl_amd/csrc`" % LAB_PATH)
    _lab = C.CDLL(LAB_PATH, mode=C.RTLD_GLOBAL)
    _lab.rmclhip_lab_version.restype = C.c_char_p
    return _lab


def check(status):
    if status == OK:
        return
    msg = lib().rmclhip_last_error().decode("utf-8", "replace")
    if status == ERR_NO_DEVICE:
        raise NoDeviceError(status, msg)
    raise RmclHipError(status, msg)
