"""Host-side mirror of the reference's correspondence-operator interface, over the C ABI.

Reference classes mirrored (names, argument meaning and error behaviour kept):
  rmcl::Correspondences_<MemT>            rmcl/include/rmcl/registration/Correspondences.hpp:16-88
  rmcl::CorrespondencesCUDA               rmcl/include/rmcl/registration/CorrespondencesCUDA.hpp:10-17
  rmcl::RCCOptixSpherical / RCCEmbreeO1Dn rmcl/include/rmcl/registration/RCCOptix.hpp:18-93, RCCEmbree.hpp:18-83
  rm::import_*_map + rm::MapMap           rmcl_ros/src/rmcl/PCDSensorUpdaterEmbree.cpp:143-174

Everything here is plumbing: arithmetic happens in librmclhip.so (HIP kernels); a missing
library or device raises instead of falling back.
"""
import ctypes as C

import numpy as np

from . import _capi
from .types import CROSS_STATISTICS, TRANSFORM, _ptr


def _as_ptr(x):
    """device pointer from an int, a torch tensor or a DeviceArray"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ptr"):
        return C.c_void_p(x.ptr)
    raise TypeError("expected a device pointer, torch tensor or DeviceArray")


class Context:
    """One HIP device (rmclhip_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _capi.check(_capi.lib().rmclhip_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def device_name(self):
        buf = C.create_string_buffer(256)
        _capi.check(_capi.lib().rmclhip_ctx_device_name(self._h, buf, 256))
        return buf.value.decode()

    def set_wait_mode(self, mode):
        """how synchronous calls wait for results in host-mapped memory: "spin" (poll the completion tag, default) or "block"
        (hipStreamSynchronize); rmclhip_ctx_set_wait_mode"""
        _capi.check(_capi.lib().rmclhip_ctx_set_wait_mode(self._h, {"spin": 0, "block": 1}[mode]))

    def close(self):
        if self._h:
            _capi.lib().rmclhip_ctx_destroy(self._h)
            self._h = C.c_void_p()


class DeviceArray:
    """Minimal owning device buffer for hosts without torch (rmagine::Memory<T, VRAM_HIP> analogue)."""

    def __init__(self, ctx, dtype, count):
        self.ctx, self.dtype, self.count = ctx, np.dtype(dtype), int(count)
        p = C.c_void_p()
        _capi.check(_capi.lib().rmclhip_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @property
    def nbytes(self):
        return self.dtype.itemsize * self.count

    @classmethod
    def from_host(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        d = cls(ctx, arr.dtype, arr.size)
        d.upload(arr)
        return d

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.size == self.count
        _capi.check(_capi.lib().rmclhip_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), _ptr(arr), self.nbytes))

    def download(self):
        out = np.zeros(self.count, dtype=self.dtype)
        _capi.check(_capi.lib().rmclhip_memcpy_d2h(self.ctx.handle, _ptr(out), C.c_void_p(self.ptr), self.nbytes))
        return out

    def free(self):
        if self.ptr:
            _capi.lib().rmclhip_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipMap:
    """Immutable triangle mesh + BVH on the device (rm::EmbreeMap / rm::OptixMap analogue)."""

    def __init__(self, ctx, vertices, faces):
        self.ctx = ctx
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        self._h = C.c_void_p()
        _capi.check(_capi.lib().rmclhip_map_create(ctx.handle, _ptr(v), len(v), _ptr(f), len(f), C.byref(self._h)))

    @classmethod
    def from_scene(cls, ctx, meshes, instances=None):
        """A whole scene (what rm::import_embree_map makes of an assimp file, micp_localization.cpp:187-195):
        meshes = [(vertices, faces), ...]; instances = [(mesh index, 3x4 or 4x4 row-major affine), ...] or None for every
        mesh once, untransformed.  Face ids are global, see scene_instances()."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self._h = C.c_void_p()
        cm, ci, keep = _scene_arrays(meshes, instances)
        _capi.check(_capi.lib().rmclhip_map_create_scene(ctx.handle, cm, len(meshes), ci, 0 if instances is None else len(instances),
                                                         C.byref(self._h)))
        return self

    def scene_instances(self):
        """first global face id of every instance, + n_faces as the last entry"""
        n = C.c_uint32()
        _capi.check(_capi.lib().rmclhip_map_scene_instances(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value + 1, np.uint32)
        _capi.check(_capi.lib().rmclhip_map_scene_instances(self._h, _ptr(out), len(out), C.byref(n)))
        return out

    def scene_locate(self, face_id):
        """global face id -> (instance, face index within that instance's mesh)"""
        i, k = C.c_uint32(), C.c_uint32()
        _capi.check(_capi.lib().rmclhip_map_scene_locate(self._h, int(face_id), C.byref(i), C.byref(k)))
        return i.value, k.value

    @property
    def handle(self):
        return self._h

    def info(self):
        mi = _capi.MapInfo()
        _capi.check(_capi.lib().rmclhip_map_get_info(self._h, C.byref(mi)))
        return {k: (list(getattr(mi, k)) if k.startswith("bbox") else getattr(mi, k)) for k, _ in mi._fields_}

    def release(self):
        if self._h:
            _capi.lib().rmclhip_map_release(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _scene_arrays(meshes, instances):
    """ctypes arrays of rmclhip_mesh / rmclhip_instance (+ the numpy arrays they borrow, to be kept alive by the caller)"""
    keep = []
    cm = (_capi.Mesh * len(meshes))()
    for k, (v, f) in enumerate(meshes):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(f, dtype=np.uint32).reshape(-1, 3)
        keep += [v, f]
        cm[k].vertices_xyz, cm[k].faces_ijk, cm[k].n_vertices, cm[k].n_faces = v.ctypes.data, f.ctypes.data, len(v), len(f)
    ci = None
    if instances is not None:
        ci = (_capi.Instance * max(len(instances), 1))()
        for k, (mesh, A) in enumerate(instances):
            A = np.asarray(A, dtype=np.float32)
            if A.shape not in ((3, 4), (4, 4)):
                raise ValueError("instance transform must be 3x4 or 4x4 (row-major), got %r" % (A.shape,))
            ci[k].mesh = int(mesh)
            ci[k].transform[:] = A[:3].reshape(-1).tolist()
    return cm, ci, keep


def flatten_scene_host(meshes, instances=None):
    """the host-side flattening map_create_scene performs, without a device: (vertices, faces, first_face)"""
    cm, ci, keep = _scene_arrays(meshes, instances)
    ni = 0 if instances is None else len(instances)
    nv, nf = C.c_uint32(), C.c_uint32()
    L = _capi.lib()
    _capi.check(L.rmclhip_scene_flatten_host(cm, len(meshes), ci, ni, None, 0, None, 0, None, 0, C.byref(nv), C.byref(nf)))
    v = np.zeros((nv.value, 3), np.float32)
    f = np.zeros((nf.value, 3), np.uint32)
    ff = np.zeros((len(meshes) if instances is None else ni) + 1, np.uint32)
    _capi.check(L.rmclhip_scene_flatten_host(cm, len(meshes), ci, ni, _ptr(v), v.size, _ptr(f), f.size, _ptr(ff), ff.size,
                                             C.byref(nv), C.byref(nf)))
    return v, f, ff


def import_hip_scene(ctx, meshes, instances=None):
    """rm::import_embree_map analogue for a scene of several (instanced) meshes"""
    return HipMap.from_scene(ctx, meshes, instances)


def import_hip_map(ctx, vertices, faces):
    """rm::import_embree_map analogue for in-memory meshes (micp_localization.cpp:187-195)."""
    return HipMap(ctx, vertices, faces)


class MapMap(dict):
    """rm::MapMap: registry keyed '<name>.hip' (PCDSensorUpdaterEmbree.cpp:143-174)."""


def build_bvh_host(vertices, faces):
    """Host-only BVH build (no device): returns (info dict, nodes[n,32] u32, tris[nf,16] u32)."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
    mi = _capi.MapInfo()
    L = _capi.lib()
    _capi.check(L.rmclhip_bvh_build_host(_ptr(v), len(v), _ptr(f), len(f), C.byref(mi), None, 0, None, 0))
    nodes = np.zeros((mi.n_nodes, 32), dtype=np.uint32)
    tris = np.zeros((mi.n_tri_records, 16), dtype=np.uint32)
    _capi.check(L.rmclhip_bvh_build_host(_ptr(v), len(v), _ptr(f), len(f), C.byref(mi), _ptr(nodes), nodes.size,
                                         _ptr(tris), tris.size))
    info = {k: (list(getattr(mi, k)) if k.startswith("bbox") else getattr(mi, k)) for k, _ in mi._fields_}
    return info, nodes, tris


def build_bvh_host_pf(vertices, faces):
    """The particle filter's tree of the same map (leaves <= 2 triangles, same record array as build_bvh_host):
    returns (info dict, nodes[n,32] u32, qnodes[n,16] u32)."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
    mi = _capi.MapInfo()
    L = _capi.lib()
    _capi.check(L.rmclhip_bvh_build_host_pf(_ptr(v), len(v), _ptr(f), len(f), C.byref(mi), None, 0, None, 0))
    nodes = np.zeros((mi.n_nodes, 32), dtype=np.uint32)
    qnodes = np.zeros((mi.n_nodes, 16), dtype=np.uint32)
    _capi.check(L.rmclhip_bvh_build_host_pf(_ptr(v), len(v), _ptr(f), len(f), C.byref(mi), _ptr(nodes), nodes.size,
                                            _ptr(qnodes), qnodes.size))
    info = {k: (list(getattr(mi, k)) if k.startswith("bbox") else getattr(mi, k)) for k, _ in mi._fields_}
    return info, nodes, qnodes


def build_bvh_host_quantised(vertices, faces, n_nodes):
    """the 64-B quantised twins of the nodes of build_bvh_host (host only): qnodes[n_nodes, 16] u32."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
    q = np.zeros((int(n_nodes), 16), dtype=np.uint32)
    _capi.check(_capi.lib().rmclhip_bvh_build_host_quantised(_ptr(v), len(v), _ptr(f), len(f), _ptr(q), q.size))
    return q


# the five bundle attributes in the C ABI's argument order (rmclhip_rcc_download, rmclhip_bundle_views)
_ATTRS = ("hits", "ranges", "points", "normals", "face_ids")
_ATTR_BIT = dict(hits=_capi.OUT_HITS, ranges=_capi.OUT_RANGES, points=_capi.OUT_POINTS, normals=_capi.OUT_NORMALS, face_ids=_capi.OUT_FACE_IDS)
_ATTR_DTYPE = dict(hits=np.uint8, ranges=np.float32, points=np.float32, normals=np.float32, face_ids=np.uint32)
_ATTR_SHAPE = dict(hits=(), ranges=(), points=(3,), normals=(3,), face_ids=())


def statistics_p2l(ctx, Tpre, dataset_points, dataset_mask, model_points, model_normals, model_mask, n, max_dist):
    """rm::statistics_p2l(Tpre, dataset, model, params) as a free function on caller-owned DEVICE views
    (CorrespondencesCUDA.cpp:28): DeviceArrays / torch tensors / raw device pointers; either mask may be None.
    Returns CrossStatistics by value on the host."""
    T = np.ascontiguousarray(Tpre, dtype=TRANSFORM).reshape(1)
    out = np.zeros(1, dtype=CROSS_STATISTICS)
    _capi.check(_capi.lib().rmclhip_statistics_p2l(ctx.handle, _ptr(T), _as_ptr(dataset_points), _as_ptr(dataset_mask),
                                                   _as_ptr(model_points), _as_ptr(model_normals), _as_ptr(model_mask), int(n),
                                                   float(max_dist), _ptr(out)))
    return out[0].copy()


class UmeyamaReductionConstraints:
    """rmagine::UmeyamaReductionConstraints (only max_dist is used, micp_localization.cpp:525-526)."""

    def __init__(self, max_dist=1.0):
        self.max_dist = float(max_dist)


class CorrespondencesHIP:
    """rmcl::Correspondences_<VRAM_HIP> + CorrespondencesCUDA::computeCrossStatistics.

    Public attributes as in Correspondences.hpp:19-29: params, adaptive_max_dist_min, outdated;
    the dataset is handed over with set_dataset()/set_dataset_from_ranges().
    """

    def __init__(self, hip_map):
        if hip_map is None:
            raise RuntimeError("NO MAP")  # PCDSensorUpdaterOptix.cpp:179-182 error convention
        self.map = hip_map
        self.ctx = hip_map.ctx
        self.params = UmeyamaReductionConstraints(1.0)
        self.adaptive_max_dist_min = 1.0
        self.outdated = True
        self._h = C.c_void_p()
        _capi.check(_capi.lib().rmclhip_rcc_create(self.ctx.handle, hip_map.handle, C.byref(self._h)))
        self._model_shape = (0, 0)
        self._last_nposes = 1

    # -- Correspondences_ interface --------------------------------------------------------
    def setTsb(self, Tsb):
        T = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_rcc_set_tsb(self._h, _ptr(T)))

    def find(self, Tbm_est):
        """RCC*::find (RCCEmbree.cpp:26-36): fills the model buffers; returns None."""
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_rcc_find(self._h, _ptr(T)))
        self._last_nposes = 1

    def computeCrossStatistics(self, T_snew_sold, convergence_progress=0.0):
        """CorrespondencesCPU.cpp:10-39; returns CrossStatistics by value on the host."""
        self._push_params()
        T = np.ascontiguousarray(T_snew_sold, dtype=TRANSFORM).reshape(1)
        out = np.zeros(1, dtype=CROSS_STATISTICS)
        _capi.check(_capi.lib().rmclhip_rcc_compute_cross_statistics(self._h, _ptr(T), float(convergence_progress),
                                                                     _ptr(out)))
        return out[0].copy()

    def modelView(self, attributes=None):
        """host copy of {points, mask(hits), normals} (+ ranges, face_ids) of the last find; `attributes`: a subset of the five names
        to read back (default: the ones set_outputs() selected -- all five unless it was called)"""
        H, W = self._model_shape
        n = H * W * self._last_nposes
        names = _ATTRS if attributes is None else tuple(attributes)
        if attributes is None:
            sel = self.outputs()
            names = tuple(a for a in _ATTRS if sel & _ATTR_BIT[a])
        out = {a: np.zeros((n,) + _ATTR_SHAPE[a], _ATTR_DTYPE[a]) for a in names}
        if n:
            _capi.check(_capi.lib().rmclhip_rcc_download(self._h, *[_ptr(out[a]) if a in out else None for a in _ATTRS]))
        if "hits" in out:
            out["mask"] = out["hits"]
        return out

    # -- the rmagine-level Simulator interface (RCCEmbree.hpp:18-22: the RCC classes are simulators by protected inheritance) --------
    def set_outputs(self, mask):
        """bundle attribute selection of find / find_batch: an OR of _capi.OUT_* (rmclhip_rcc_set_outputs); a string list works too"""
        if not isinstance(mask, int):
            mask = sum(_ATTR_BIT[a] for a in mask)
        _capi.check(_capi.lib().rmclhip_rcc_set_outputs(self._h, int(mask)))

    def outputs(self):
        m = C.c_uint32(0)
        _capi.check(_capi.lib().rmclhip_rcc_get_outputs(self._h, C.byref(m)))
        return m.value

    def simulate(self, Tbm, attributes=("ranges",), into=None, poses_dev=None):
        """rm::Simulator::simulate(Tbm, Bundle&) / simulate<Bundle>(Tbm) / the batch form with an array of poses
        (scan_map_segmentation_embree.cpp:87, lidar_corrector_embree_benchmark.cpp:117): only the named attributes are written.
        into: {attribute: DeviceArray / torch tensor} of caller-owned device memory (the bundle); None: a fresh bundle of DeviceArrays
        is allocated.  poses_dev: device memory holding the poses instead of `Tbm` (lidar_corrector_optix_benchmark.cpp:119), `Tbm` is
        then the pose count.  Returns the bundle (dict of device buffers); `download_bundle` brings it to the host.  The operator's own
        model buffers and cached statistics are not touched."""
        H, W = self._model_shape
        if poses_dev is not None:
            nposes, tptr, on_dev = int(Tbm), _as_ptr(poses_dev), 1
        else:
            T = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
            nposes, tptr, on_dev = len(T), _ptr(T), 0
        n = H * W * nposes
        bundle = {} if into is None else dict(into)
        for a in attributes:
            if a not in _ATTRS:
                raise ValueError("unknown bundle attribute %r" % (a,))
            if a not in bundle:
                bundle[a] = DeviceArray(self.ctx, _ATTR_DTYPE[a], max(n, 1) * (3 if _ATTR_SHAPE[a] else 1))
        v = _capi.BundleViews()
        for a, field in zip(_ATTRS, ("hits_dev", "ranges_dev", "points_xyz_dev", "normals_xyz_dev", "face_ids_dev")):
            if a in bundle:
                setattr(v, field, _as_ptr(bundle[a]))
        _capi.check(_capi.lib().rmclhip_rcc_simulate(self._h, tptr, nposes, on_dev, C.byref(v)))
        return bundle

    @staticmethod
    def download_bundle(bundle):
        """host copies of a simulate() bundle of DeviceArrays: {attribute: ndarray} with points / normals as (n, 3)"""
        out = {}
        for a, d in bundle.items():
            h = d.download()
            out[a] = h.reshape(-1, 3) if _ATTR_SHAPE[a] else h
        return out

    # -- dataset ------------------------------------------------------------------------------
    def set_dataset(self, points, mask=None, device=False):
        """dataset {points, mask} (MICPSphericalSensorCPU.cpp:181-233 fills these)."""
        if device:
            n = int(points.numel() // 3) if hasattr(points, "numel") else points.count // 3
            _capi.check(_capi.lib().rmclhip_rcc_set_dataset(self._h, _as_ptr(points), _as_ptr(mask), n, 1))
        else:
            p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
            m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
            _capi.check(_capi.lib().rmclhip_rcc_set_dataset(self._h, _ptr(p), _ptr(m), len(p), 0))
        self.outdated = True

    def set_dataset_from_ranges(self, ranges):
        """unpackMessage mirror: points = dir*range (+orig), mask = range within model.range"""
        r = np.ascontiguousarray(ranges, dtype=np.float32).reshape(-1)
        nv = C.c_uint32(0)
        _capi.check(_capi.lib().rmclhip_rcc_set_dataset_from_ranges(self._h, _ptr(r), len(r), C.byref(nv)))
        self.outdated = True
        return nv.value

    # -- extras (fast paths) --------------------------------------------------------------------
    def correct_once(self, Tom, Tbo, n_iter, convergence_progress=0.0, refind_each_iteration=False):
        """MICP-L inner loop for this one sensor on the device (micp_localization.cpp:900-964)."""
        self._push_params()
        a = np.ascontiguousarray(Tom, dtype=TRANSFORM).reshape(1)
        b = np.ascontiguousarray(Tbo, dtype=TRANSFORM).reshape(1)
        T = np.zeros(1, dtype=TRANSFORM)
        s = np.zeros(1, dtype=CROSS_STATISTICS)
        _capi.check(_capi.lib().rmclhip_rcc_correct_once(self._h, _ptr(a), _ptr(b), int(n_iter),
                                                         float(convergence_progress), int(bool(refind_each_iteration)),
                                                         _ptr(T), _ptr(s)))
        self._last_nposes = 1
        return T[0].copy(), s[0].copy()

    def time_correct_once(self, Tom, Tbo, n_iter, convergence_progress=0.0, refind_each_iteration=False, iters=20):
        """mean host-clock ms of one synchronous correct_once at the C ABI (no Python in the timed loop)"""
        self._push_params()
        a = np.ascontiguousarray(Tom, dtype=TRANSFORM).reshape(1)
        b = np.ascontiguousarray(Tbo, dtype=TRANSFORM).reshape(1)
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_time_correct_once(self._h, _ptr(a), _ptr(b), int(n_iter), float(convergence_progress),
                                                              int(bool(refind_each_iteration)), int(iters), C.byref(ms)))
        self._last_nposes = 1
        return ms.value

    def time_caller_loop(self, Tom, Tbo, n_iter, convergence_progress=0.0, iters=20):
        """mean host-clock ms of the reference's UNCHANGED caller loop (find once, then n_iter x computeCrossStatistics + the host
        algebra + umeyama_transform: micp_localization.cpp:900-964) through the public C entry points; returns
        (ms, T_onew_oold, merged statistics of the last iteration)"""
        self._push_params()
        a = np.ascontiguousarray(Tom, dtype=TRANSFORM).reshape(1)
        b = np.ascontiguousarray(Tbo, dtype=TRANSFORM).reshape(1)
        Tout, st = np.zeros(1, dtype=TRANSFORM), np.zeros(1, dtype=CROSS_STATISTICS)
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_time_caller_loop(self._h, _ptr(a), _ptr(b), int(n_iter), float(convergence_progress),
                                                             int(iters), _ptr(Tout), _ptr(st), C.byref(ms)))
        self._last_nposes = 1
        return ms.value, Tout[0].copy(), st[0].copy()

    def correct_batch(self, Tbm):
        """v1 SphereCorrector::correct (lidar_corrector_embree_benchmark.cpp:127-135): Tdelta per pose."""
        self._push_params()
        T = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        out = np.zeros(len(T), dtype=TRANSFORM)
        st = np.zeros(len(T), dtype=CROSS_STATISTICS)
        _capi.check(_capi.lib().rmclhip_rcc_correct_batch(self._h, _ptr(T), len(T), _ptr(out), _ptr(st)))
        self._last_nposes = len(T)
        return out, st

    def find_batch(self, Tbm):
        """Simulator::simulate(Memory<Transform>, Bundle&): pose-major model buffers."""
        T = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        _capi.check(_capi.lib().rmclhip_rcc_find_batch(self._h, _ptr(T), len(T)))
        self._last_nposes = len(T)

    def time_find_batch(self, Tbm, iters=10):
        T = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_time_find_batch(self._h, _ptr(T), len(T), int(iters), C.byref(ms)))
        self._last_nposes = len(T)
        return ms.value

    def time_find(self, Tbm_est, iters=20):
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_time_find(self._h, _ptr(T), int(iters), C.byref(ms)))
        self._last_nposes = 1
        return ms.value

    def time_reduce(self, T_snew_sold, iters=20):
        self._push_params()
        T = np.ascontiguousarray(T_snew_sold, dtype=TRANSFORM).reshape(1)
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_time_reduce(self._h, _ptr(T), int(iters), C.byref(ms)))
        return ms.value

    def find_async(self, Tbm_est):
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_rcc_find_async(self._h, _ptr(T)))
        self._last_nposes = 1

    def find_async_fn(self, Tbm_est):
        """find_async(Tbm_est) as a zero-argument callable with the argument conversion done ONCE (a loop of identical finds then pays
        one ctypes call per step: bench.py's timed region).  The callable keeps this operator alive and reads its handle on every
        call: after close() it raises instead of handing a freed handle to the library."""
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1).copy()
        fn, ptr, check, owner = _capi.lib().rmclhip_rcc_find_async, _ptr(T), _capi.check, self

        def call(_keep=T):
            h = owner._h
            if not h:
                raise RuntimeError("find_async_fn: the operator was closed")
            rc = fn(h, ptr)
            if rc:
                check(rc)
            owner._last_nposes = 1
        return call

    def sync(self):
        _capi.check(_capi.lib().rmclhip_rcc_sync(self._h))

    def set_kernel_timing(self, on):
        """bracket synchronous find / computeCrossStatistics calls with HIP events so that last_kernel_ms() has values (opt-in)"""
        _capi.check(_capi.lib().rmclhip_rcc_set_kernel_timing(self._h, 1 if on else 0))

    def time_find_sync(self, Tbm_est, iters=50):
        """mean host-clock ms of one synchronous find at the C ABI"""
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_time_find_sync(self._h, _ptr(T), int(iters), C.byref(ms)))
        self._last_nposes = 1
        return ms.value

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _capi.check(_capi.lib().rmclhip_rcc_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_variant(self, variant):
        _capi.check(_capi.lib().rmclhip_rcc_set_variant(self._h, int(variant)))

    def autotune(self, Tbm_est):
        """time the product's single-scan traversals on this map / model / pose and keep the fastest as the automatic choice
        (rmclhip.h: rmclhip_rcc_autotune); returns (kind, kernel milliseconds)"""
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        kind, ms = C.c_int(), C.c_float()
        _capi.check(_capi.lib().rmclhip_rcc_autotune(self._h, _ptr(T), C.byref(kind), C.byref(ms)))
        return kind.value, ms.value

    def autotune_batch(self, Tbm_poses):
        """the same measurement for pose batches (find_batch / correct_batch): kinds 23 / 24 with and without the frontier start;
        returns (kind, kernel milliseconds) -- 19 / 22 stand for 23 / 24 without the frontier start"""
        P = np.ascontiguousarray(Tbm_poses, dtype=TRANSFORM).reshape(-1)
        kind, ms = C.c_int(), C.c_float()
        _capi.check(_capi.lib().rmclhip_rcc_autotune_batch(self._h, _ptr(P), len(P), C.byref(kind), C.byref(ms)))
        return kind.value, ms.value

    def set_traversal(self, kind):
        """traversal kind 0..32 alone (bits 4 and 5 of the kind travel in bits 13 and 14 of the variant word, see rmclhip.h)"""
        self.set_variant((int(kind) & 15) | (((int(kind) >> 4) & 1) << 13) | (((int(kind) >> 5) & 1) << 14))

    def set_micp_fast(self, mode):
        """moment form of the schedule-(R) loop of correctOnce: 0 = never, 1 = automatic, iterations on the host from the published
        moments (default), 2 = device loop through a hipGraph, 3 = device loop, moments in a pass of their own, 4 = device loop behind
        a find with the moment epilogue (rmclhip.h)"""
        _capi.check(_capi.lib().rmclhip_rcc_set_micp_fast(self._h, int(mode)))

    def ccs_info(self):
        """how computeCrossStatistics was served (rmclhip_ccs_info): calls, from_moments (host, no launch), passes, speculative_finds"""
        info = _capi.CcsInfo()
        _capi.check(_capi.lib().rmclhip_rcc_ccs_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def micp_fast_info(self):
        """outcomes of the moment-form attempts of this operator as a dict (rmclhip_micp_fast_info)"""
        info = _capi.MicpFastInfo()
        _capi.check(_capi.lib().rmclhip_rcc_micp_fast_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def find_variant(self, nposes=1):
        """the traversal the automatic rule (variant 15) launches for `nposes` scans of the current model"""
        v = C.c_int(0)
        _capi.check(_capi.lib().rmclhip_rcc_find_variant(self._h, int(nposes), C.byref(v)))
        return v.value

    def debug_wave_clock(self, Tbm_est):
        """diagnostics: uint32 [n_waves, 8] = {entry clock, exit clock, entry realtime (100 MHz), tile | xcc << 24, clock
        before / after the traversal, clock at stores issued, 0}"""
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        H, W = self._model_shape
        buf = np.zeros((H * W + 8 * 64 * 64) * 8, np.uint32)
        nw = C.c_uint32(0)
        _capi.check(_capi.lib().rmclhip_debug_wave_clock(self._h, _ptr(T), _ptr(buf), buf.size, C.byref(nw)))
        self._last_nposes = 1
        # kind 31 also leaves [n_waves, 16] phase stamps of its cooperative descent behind the table (traverse.hip.h RMCL_STAMP)
        self._last_descent_stamps = buf[nw.value * 8: nw.value * 24].reshape(nw.value, 16).copy()
        return buf[: nw.value * 8].reshape(nw.value, 8)

    def debug_micp_moments(self):
        """diagnostics: (96 moment totals, partial rows, undecided correspondences) of the last moment-form attempt"""
        tot = np.zeros(96, np.float64)
        rows, unc = C.c_uint32(0), C.c_uint64(0)
        _capi.check(_capi.lib().rmclhip_debug_micp_moments(self._h, _ptr(tot), C.byref(rows), C.byref(unc)))
        return tot, rows.value, unc.value

    def debug_probe_find(self, Tbm_est, mode=0):
        """diagnostics: per-wave step timeline of one spherical scan -> uint32 array [n_tiles, 256, 2]"""
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        H, W = self._model_shape
        cap = ((H + 7) // 8 + 1) * ((W + 7) // 8 + 1) * 64 * 512   # any tile shape
        cap = min(cap, (H * W + 64 * 64) * 8 * 2)
        buf = np.zeros(max(cap, 512), np.uint32)
        nt = C.c_uint32(0)
        _capi.check(_capi.lib().rmclhip_debug_probe_find(self._h, _ptr(T), int(mode), _ptr(buf), buf.size, C.byref(nt)))
        self._last_nposes = 1
        return buf[: nt.value * 512].reshape(nt.value, 256, 2)

    def _push_params(self):
        _capi.check(_capi.lib().rmclhip_rcc_set_params(self._h, float(self.params.max_dist),
                                                       float(self.adaptive_max_dist_min)))

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            _capi.lib().rmclhip_rcc_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def _borrow(cls, handle):
        """wrap an operator handle owned by someone else (a replica of ShardedCorrectorHip): never destroyed from here"""
        self = cls.__new__(cls)
        self.map = self.ctx = None
        self.params = UmeyamaReductionConstraints(1.0)
        self.adaptive_max_dist_min = 1.0
        self.outdated = True
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self._model_shape = (0, 0)
        self._last_nposes = 1
        return self


class ShardedCorrectorHip:
    """Pose batches of the v1 SphereCorrector::correct shape (lidar_corrector_optix_benchmark.cpp:86-133) over several devices of ONE
    process (rmclhip_rcc_sharded_*): one operator replica per device over one host BVH build, poses block-partitioned with
    distributed.shard_bounds, no exchange.  `replica_class` is the operator type of the replicas (RCCHipSpherical, RCCHipO1Dn, ...);
    configure them through `.replicas` (or `.for_each`), then `.correct_batch(Tbm)` == the unsharded RCCHip*.correct_batch."""

    def __init__(self, devices, vertices, faces, replica_class=None):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        _capi.check(_capi.lib().rmclhip_rcc_sharded_create(devs, len(devices), _ptr(v), len(v), _ptr(f), len(f), C.byref(self._h)))
        cls = replica_class or RCCHipSpherical
        self.replicas = []
        for r in range(len(devices)):
            h = C.c_void_p()
            _capi.check(_capi.lib().rmclhip_rcc_sharded_replica(self._h, r, C.byref(h)))
            rep = cls._borrow(h.value)
            rep._owner = self   # a replica in use keeps its owner (and so its handle) alive; close() below nulls the replicas first
            self.replicas.append(rep)

    @property
    def world(self):
        return len(self.replicas)

    def for_each(self, fn):
        for r in self.replicas:
            fn(r)

    def correct_batch(self, Tbm, want_stats=True):
        for r in self.replicas:
            r._push_params()
        T = np.ascontiguousarray(Tbm, dtype=TRANSFORM).reshape(-1)
        out = np.zeros(len(T), dtype=TRANSFORM)
        st = np.zeros(len(T), dtype=CROSS_STATISTICS)
        _capi.check(_capi.lib().rmclhip_rcc_sharded_correct_batch(self._h, _ptr(T), len(T), _ptr(out), _ptr(st) if want_stats else None))
        return out, st

    def close(self):
        if self._h:
            for r in self.replicas:
                r.close()          # borrowed: only forgets the handle -- a later call on the replica raises instead of touching freed memory
                r._owner = None
            _capi.lib().rmclhip_rcc_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RCCHipSpherical(CorrespondencesHIP):
    """rmcl::RCCOptixSpherical / RCCEmbreeSpherical on gfx950 (RCCEmbree.cpp:8-36)."""

    def setModel(self, sensor_model):
        _capi.check(_capi.lib().rmclhip_rcc_set_model_spherical(self._h, C.byref(sensor_model)))
        self._model_shape = (int(sensor_model.phi.size), int(sensor_model.theta.size))


class RCCHipO1Dn(CorrespondencesHIP):
    """rmcl::RCCEmbreeO1Dn / RCCOptixO1Dn on gfx950 (RCCEmbree.cpp:71-99)."""

    def setModel(self, width, height, range_min, range_max, orig, dirs):
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        if len(d) != width * height:
            raise ValueError("dirs must hold width*height directions")
        rng = _capi.Interval(range_min, range_max)
        o = _capi.Vec3(*[float(x) for x in orig])
        _capi.check(_capi.lib().rmclhip_rcc_set_model_o1dn(self._h, int(width), int(height), rng, o, _ptr(d)))
        self._model_shape = (int(height), int(width))

    def setInputPointCloud2(self, data, width, height, point_step, row_step, offset_x, offset_y, offset_z,
                            datatype=7, range_min=0.0, range_max=1e30, filter_height=None, filter_width=None,
                            device=False, nbytes=None):
        """A sensor_msgs/PointCloud2 as received (bytes / uint8 array, or a device pointer with device=True and
        nbytes) becomes the O1Dn model and the dataset in one device pass: estimateModelAndData
        (conversions.cpp:869-1002) + filter (scan_operations.cpp:41-116) + MICPO1DnSensorCPU::unpackMessage.
        filter_*: (skip_begin, skip_end, increment) or None.  Returns (width, height, n_valid) after filtering."""
        L = _capi.PointCloud2Layout(int(width), int(height), int(point_step), int(row_step), int(offset_x),
                                    int(offset_y), int(offset_z), int(datatype))
        fh = _capi.Filter1D(*filter_height) if filter_height is not None else None
        fw = _capi.Filter1D(*filter_width) if filter_width is not None else None
        if device:
            ptr, nb = _as_ptr(data), int(nbytes)
        else:
            buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8)) if isinstance(data, (bytes, bytearray, memoryview)) \
                else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
            ptr, nb = _ptr(buf), buf.size
        ow, oh, nv = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _capi.check(_capi.lib().rmclhip_rcc_set_input_pointcloud2(
            self._h, ptr, nb, C.byref(L), C.byref(fh) if fh is not None else None,
            C.byref(fw) if fw is not None else None, _capi.Interval(range_min, range_max), int(bool(device)),
            C.byref(ow), C.byref(oh), C.byref(nv)))
        self._model_shape = (oh.value, ow.value)
        self.outdated = True
        return ow.value, oh.value, nv.value


class RCCHipPinhole(CorrespondencesHIP):
    """rmcl::RCCEmbreePinhole / RCCOptixPinhole on gfx950 (RCCEmbree.cpp:39-68): depth camera."""

    def setModel(self, width, height, range_min, range_max, fx, fy, cx, cy):
        rng = _capi.Interval(range_min, range_max)
        _capi.check(_capi.lib().rmclhip_rcc_set_model_pinhole(self._h, int(width), int(height), rng, float(fx), float(fy),
                                                              float(cx), float(cy)))
        self._model_shape = (int(height), int(width))


class RCCHipOnDn(CorrespondencesHIP):
    """rmcl::RCCEmbreeOnDn / RCCOptixOnDn on gfx950 (RCCEmbree.cpp:102-130): N origins, N directions."""

    def setModel(self, width, height, range_min, range_max, origs, dirs):
        o = np.ascontiguousarray(origs, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        if len(o) != width * height or len(d) != width * height:
            raise ValueError("origs / dirs must hold width*height entries")
        rng = _capi.Interval(range_min, range_max)
        _capi.check(_capi.lib().rmclhip_rcc_set_model_ondn(self._h, int(width), int(height), rng, _ptr(o), _ptr(d)))
        self._model_shape = (int(height), int(width))


class CPCHip(CorrespondencesHIP):
    """rmcl::CPCEmbree on gfx950 (CPCEmbree.cpp:11-44): closest-point correspondences; find() pairs every
    dataset point with the nearest surface point of the map (no sensor model needed)."""

    def set_tracking(self, on=True):
        """start every query from the triangle the point was closest to in the previous find (same results; rmclhip.h)"""
        _capi.check(_capi.lib().rmclhip_rcc_set_cpc_tracking(self._h, 1 if on else 0))

    def set_bounded(self, on=True):
        """search only within params.max_dist: hits and every output of a point that hits are unchanged, points with no surface
        within max_dist get NaN outputs instead of their (gated-out) global closest point (rmclhip.h)"""
        _capi.check(_capi.lib().rmclhip_rcc_set_cpc_bounded(self._h, 1 if on else 0))

    def set_grid(self, on):
        """the map's near grid seeds points that have no tracking seed (default on; results do not depend on it): rmclhip_rcc_set_cpc_grid"""
        _capi.check(_capi.lib().rmclhip_rcc_set_cpc_grid(self._h, 1 if on else 0))

    def find(self, Tbm_est):
        self._push_params()   # hits = (distance <= params.max_dist)
        T = np.ascontiguousarray(Tbm_est, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_rcc_find_cpc(self._h, _ptr(T)))
        self._last_nposes = 1

    def set_dataset(self, points, mask=None, device=False):
        # element count exactly as the base class derives it (torch tensor / DeviceArray / host array)
        if device:
            n = int(points.numel() // 3) if hasattr(points, "numel") else int(points.count // 3)
        else:
            n = int(np.asarray(points).size // 3)
        super().set_dataset(points, mask, device)
        self._model_shape = (1, n)
