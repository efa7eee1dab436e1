"""Particle-filter sensor update, mirror of rmcl::SensorUpdater<MemT> over the C ABI.

Reference: SensorUpdaterBase{init,reset} + ParticleUpdater<MemT>::update(poses, attrs, cfg)
(rmcl_ros/include/rmcl_ros/rmcl/SensorUpdater.hpp:18-42, ParticleUpdater.hpp:24-44), implemented by
PCDSensorUpdaterEmbree / PCDSensorUpdaterOptix (rmcl_ros/src/rmcl/PCDSensorUpdater*.cpp).
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from .registration import _as_ptr
from .types import RANGE_MEASUREMENT, TRANSFORM, _ptr, gladiator_config, pf_params


def beams_from_points(points):
    """PCDSensorUpdaterEmbree.cpp:313-327: meas_s = {orig 0, dir = p/|p|, range = |p|, cov = 0.1 I}."""
    p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(len(p), dtype=RANGE_MEASUREMENT)
    # rmagine Vector3::l2norm / normalize in float32
    nrm = np.sqrt((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]).astype(np.float32)
    out["dir"]["x"], out["dir"]["y"], out["dir"]["z"] = p[:, 0] / nrm, p[:, 1] / nrm, p[:, 2] / nrm
    out["range"] = nrm
    out["cov"][:, 0] = out["cov"][:, 4] = out["cov"][:, 8] = np.float32(0.1)
    return out


def sample_beams_pointcloud2(data, width, height, point_step, row_step, offset_x, offset_y, offset_z, samples, seed,
                             datatype=7):
    """PCDSensorUpdaterEmbree.cpp:276-327 on the raw sensor_msgs/PointCloud2 bytes, through the C ABI
    (rmclhip_pf_sample_beams_pointcloud2, a host function): `samples` uniformly random points -- std::mt19937(seed),
    index = draw % (width * height), up to 100 retries for a point without NaN -- as RangeMeasurements."""
    buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8)) if isinstance(data, (bytes, bytearray, memoryview)) \
        else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    L = _capi.PointCloud2Layout(int(width), int(height), int(point_step), int(row_step), int(offset_x), int(offset_y),
                                int(offset_z), int(datatype))
    out = np.zeros(int(samples), dtype=RANGE_MEASUREMENT)
    n = C.c_uint32(0)
    _capi.check(_capi.lib().rmclhip_pf_sample_beams_pointcloud2(_ptr(buf), buf.size, C.byref(L), int(samples), int(seed),
                                                               _ptr(out), C.byref(n)))
    return out[: n.value].copy()


def sample_beams(cloud_xyz, samples, seed):
    """the same sampling on an [n, 3] float32 array (an unorganised cloud: width n, height 1, point_step 12)"""
    pts = np.ascontiguousarray(cloud_xyz, dtype=np.float32).reshape(-1, 3)
    return sample_beams_pointcloud2(pts.view(np.uint8).reshape(-1), len(pts), 1, 12, 12 * len(pts), 0, 4, 8, samples, seed)


def combined_forget_rate(forget_rate_per_meter, forget_rate_per_second, dist_travelled, dt):
    """TFMotionUpdaterCPU.cpp:172-174: per-meter and per-second rates turned absolute (exponential) and multiplied."""
    space = 1.0 - math.pow(1.0 - forget_rate_per_meter, dist_travelled)
    tim = 1.0 - math.pow(1.0 - forget_rate_per_second, dt)
    return space * tim


class TFMotionUpdaterHip:
    """rmcl::TFMotionUpdaterGPU on gfx950 + the wall-collision test of TFMotionUpdaterCPU (MotionUpdater<MemT>)."""

    def __init__(self, hip_map, check_collision=True):
        if hip_map is None:
            raise RuntimeError("NO MAP")
        self.map, self.ctx, self.check_collision = hip_map, hip_map.ctx, check_collision
        self._h = C.c_void_p()

    def init(self):
        if not self._h:
            _capi.check(_capi.lib().rmclhip_pf_create(self.ctx.handle, self.map.handle, C.byref(self._h)))

    def reset(self):
        """TFMotionUpdaterCPU::reset forgets the previous odometry pose (host-side state of the caller)."""

    def update(self, particle_poses, particle_attrs, n_particles, T_bnew_bold, forget_rate):
        self.init()
        T = np.ascontiguousarray(T_bnew_bold, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_pf_motion_update(self._h, _as_ptr(particle_poses), _as_ptr(particle_attrs),
                                                         int(n_particles), _ptr(T), float(forget_rate),
                                                         int(bool(self.check_collision))))
        return {}

    def close(self):
        if self._h:
            _capi.lib().rmclhip_pf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PCDSensorUpdaterHip:
    """rmcl::PCDSensorUpdaterOptix on gfx950: update() mutates attrs in place on the device."""

    def __init__(self, hip_map):
        if hip_map is None:
            raise RuntimeError("NO MAP")  # PCDSensorUpdaterOptix.cpp:179-182
        self.map = hip_map
        self.ctx = hip_map.ctx
        self.config = pf_params()
        self._h = C.c_void_p()
        self._beams = None
        self._Tsb = None

    def init(self):
        """SensorUpdaterBase::init (PCDSensorUpdaterEmbree.cpp:107-114)."""
        if not self._h:
            _capi.check(_capi.lib().rmclhip_pf_create(self.ctx.handle, self.map.handle, C.byref(self._h)))
        self._push_params()

    def reset(self):
        """SensorUpdaterBase::reset (PCDSensorUpdaterEmbree.cpp:116-119): nothing to reset."""

    def setInput(self, beams, Tsb):
        """Input<>::setInput analogue: the sampled measurements (sensor frame) and the sensor->base TF."""
        self._beams = np.ascontiguousarray(beams, dtype=RANGE_MEASUREMENT).reshape(-1)
        self._Tsb = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)

    def update(self, particle_poses, particle_attrs, n_particles=None, sync=True):
        """ParticleUpdater::update(poses, attrs): device views of n Transform / ParticleAttributes."""
        if not self._h:
            self.init()
        if self._beams is None:
            raise RuntimeError("setInput() first")
        if n_particles is None:
            n_particles = particle_poses.count if hasattr(particle_poses, "count") else particle_poses.shape[0]
        self._push_params()
        fn = _capi.lib().rmclhip_pf_update if sync else _capi.lib().rmclhip_pf_update_async
        _capi.check(fn(self._h, _as_ptr(particle_poses), _as_ptr(particle_attrs), int(n_particles),
                       _ptr(self._beams), len(self._beams), _ptr(self._Tsb)))
        return {}

    def sync(self):
        _capi.check(_capi.lib().rmclhip_pf_sync(self._h))

    def set_error_output(self, errors_dev):
        if not self._h:
            self.init()
        _capi.check(_capi.lib().rmclhip_pf_set_error_output(self._h, _as_ptr(errors_dev)))

    def extract_weights(self, particle_attrs, n_particles, weights_dev):
        _capi.check(_capi.lib().rmclhip_pf_extract_weights(self._h, _as_ptr(particle_attrs), int(n_particles),
                                                           _as_ptr(weights_dev)))

    def time_update(self, particle_poses, particle_attrs, n_particles, iters=5):
        if not self._h:
            self.init()
        self._push_params()
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_pf_time_update(self._h, _as_ptr(particle_poses), _as_ptr(particle_attrs),
                                                       int(n_particles), _ptr(self._beams), len(self._beams),
                                                       _ptr(self._Tsb), int(iters), C.byref(ms)))
        return ms.value

    def time_update_unfused(self, particle_poses, particle_attrs, n_particles, sync_each_beam=True, iters=2):
        """the reference's GPU schedule as a comparator (PCDSensorUpdaterOptix.cpp:319-338): one single-beam update per beam, synchronised
        after each (or only at the end); host-clock ms per sequence of all beams (rmclhip_pf_time_update_unfused)"""
        if not self._h:
            self.init()
        self._push_params()
        ms = C.c_float(0)
        _capi.check(_capi.lib().rmclhip_pf_time_update_unfused(self._h, _as_ptr(particle_poses), _as_ptr(particle_attrs), int(n_particles),
                                                               _ptr(self._beams), len(self._beams), _ptr(self._Tsb),
                                                               1 if sync_each_beam else 0, int(iters), C.byref(ms)))
        return ms.value

    def set_variant(self, v):
        if not self._h:
            self.init()
        _capi.check(_capi.lib().rmclhip_pf_set_variant(self._h, int(v)))

    def set_schedule(self, refill_idle_lanes=0, tail_lanes=8):
        if not self._h:
            self.init()
        _capi.check(_capi.lib().rmclhip_pf_set_schedule(self._h, int(refill_idle_lanes), int(tail_lanes)))

    def set_mapping(self, mapping, particles_per_block=0, order=None):
        """ray dealing of the update (rmclhip_pf_set_mapping): 0 beam-minor (default), 1 particle-minor (converged clouds); order: a
        DeviceArray of uint32 slot -> particle indices (kept alive by this object) or None"""
        if not self._h:
            self.init()
        self._order = order
        _capi.check(_capi.lib().rmclhip_pf_set_mapping(self._h, int(mapping), int(particles_per_block),
                                                       order.ptr if order is not None else None, order.count if order is not None else 0))

    def _push_params(self):
        _capi.check(_capi.lib().rmclhip_pf_set_params(self._h, C.byref(self.config)))

    def close(self):
        if self._h:
            _capi.lib().rmclhip_pf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GladiatorResamplerHip:
    """rmcl::GladiatorResamplerGPU on gfx950 (Resampler<MemT>: init / reset / update(poses, attrs, poses_new,
    attrs_new) -> {n_particles}; GladiatorResamplerGPU.cpp:46-81).  The tournament is out of place (the node's
    particle double buffer, rmcl_localization.cpp:600-640); `seed` and the running step counter select the
    Philox stream, so a run is reproducible."""

    def __init__(self, ctx, seed=1234):
        self.ctx = ctx
        self.config = gladiator_config()
        self.seed = int(seed)
        self.step = 0
        self._h = C.c_void_p()

    def init(self):
        if not self._h:
            _capi.check(_capi.lib().rmclhip_resampler_create(self.ctx.handle, C.byref(self._h)))

    def reset(self):
        self.step = 0

    def compute_stats(self, particle_attrs, n_particles):
        """compute_stats (resampling.cu:83-92): {sum, max} of the particle likelihoods."""
        self.init()
        out = _capi.LikelihoodStats()
        _capi.check(_capi.lib().rmclhip_resampler_compute_stats(self._h, _as_ptr(particle_attrs), int(n_particles),
                                                                C.byref(out)))
        return {"sum": out.sum, "max": out.max}

    def compute_stats_weights(self, weights, n):
        """{sum, max} of a dense float32 weight vector on the device (the all-gathered likelihood.mean of a sharded cloud): the bits
        compute_stats gives for attributes holding the same values (rmclhip_resampler_compute_stats_weights)"""
        self.init()
        out = _capi.LikelihoodStats()
        _capi.check(_capi.lib().rmclhip_resampler_compute_stats_weights(self._h, _as_ptr(weights), int(n), C.byref(out)))
        return {"sum": out.sum, "max": out.max}

    def update(self, particle_poses, particle_attrs, particle_poses_new, particle_attrs_new, n_particles,
               first=0, count=None):
        """champions first..first+count-1 (default: all) vs random enemies out of all n_particles; results in
        particle_*_new[0..count)."""
        self.init()
        if count is None:
            count = int(n_particles) - int(first)
        _capi.check(_capi.lib().rmclhip_resampler_gladiator(
            self._h, _as_ptr(particle_poses), _as_ptr(particle_attrs), int(n_particles), _as_ptr(particle_poses_new),
            _as_ptr(particle_attrs_new), int(first), int(count), C.byref(self.config), self.seed, self.step))
        self.step += 1
        return {"n_particles": int(count)}

    def close(self):
        if self._h:
            _capi.lib().rmclhip_resampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidualResamplerHip(GladiatorResamplerHip):
    """rmcl::ResidualResamplerCPU on gfx950 (ResidualResamplerCPU.cpp:55-203), the PF node's second resampler plugin: random
    particles are inserted floor(L / sum(L) * N_new) times each, perturbed, until the new cloud is full.  Same interface and
    parameters as the gladiator (`config.trans_dist_metric` is ignored: this resampler measures |dt|^2); `last_draws` = the
    iterations the reference's sequential loop would have run."""

    def __init__(self, ctx, seed=1234):
        super().__init__(ctx, seed)
        self.last_draws = 0

    def update(self, particle_poses, particle_attrs, particle_poses_new, particle_attrs_new, n_particles, n_particles_new=None,
               first=0, count=None):
        """slots first..first+count-1 (default: all) of a new cloud of n_particles_new (default n_particles) particles; results in
        particle_*_new[0..count)."""
        self.init()
        n_new = int(n_particles if n_particles_new is None else n_particles_new)
        if count is None:
            count = n_new - int(first)
        nd = C.c_uint64(0)
        _capi.check(_capi.lib().rmclhip_resampler_residual(
            self._h, _as_ptr(particle_poses), _as_ptr(particle_attrs), int(n_particles), _as_ptr(particle_poses_new),
            _as_ptr(particle_attrs_new), n_new, int(first), int(count), C.byref(self.config), self.seed, self.step, C.byref(nd)))
        self.step += 1
        self.last_draws = int(nd.value)
        return {"n_particles": int(count)}


class ShardedParticleFilterHip:
    """A particle cloud block-partitioned over the devices of ONE process (rmclhip_comm / rmclhip_pf_sharded: RCCL
    ncclCommInitAll + all-gather / all-reduce over xGMI) -- the multi-GPU form of PCDSensorUpdater + GladiatorResampler +
    RmclNode::estimateStats for the single-process node (rmcl_localization.cpp:482-552, 642-731)."""

    def __init__(self, vertices, faces, devices=(0,), loopback=False):
        """loopback=True: the in-process stand-in for RCCL (rmclhip_comm_create_loopback) -- `devices` may then repeat a device, which
        lets a one-GPU box run and check the ndev > 1 code paths; never a production configuration."""
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._comm = C.c_void_p()
        create = _capi.lib().rmclhip_comm_create_loopback if loopback else _capi.lib().rmclhip_comm_create
        _capi.check(create(devs, len(devices), C.byref(self._comm)))
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        self._h = C.c_void_p()
        try:
            _capi.check(_capi.lib().rmclhip_pf_sharded_create(self._comm, _ptr(v), len(v), _ptr(f), len(f), C.byref(self._h)))
        except Exception:
            _capi.lib().rmclhip_comm_destroy(self._comm)
            self._comm = C.c_void_p()
            raise
        self.world = len(devices)
        self.n_total = 0
        self.config_ = pf_params()

    def set_particles(self, poses, attrs):
        p = np.ascontiguousarray(poses, dtype=TRANSFORM).reshape(-1)
        a = np.ascontiguousarray(attrs).reshape(-1)
        assert a.dtype.itemsize == 36 and len(a) == len(p)
        _capi.check(_capi.lib().rmclhip_pf_sharded_set_particles(self._h, _ptr(p), _ptr(a), len(p)))
        self.n_total = len(p)

    def download(self):
        from .types import PARTICLE_ATTRIBUTES
        p, a = np.zeros(self.n_total, TRANSFORM), np.zeros(self.n_total, PARTICLE_ATTRIBUTES)
        _capi.check(_capi.lib().rmclhip_pf_sharded_download(self._h, _ptr(p), _ptr(a)))
        return p, a

    def update(self, beams, Tsb):
        """sensor update on every device's block + the weight all-gather; returns the dense weights as rank 0 holds them"""
        b = np.ascontiguousarray(beams, dtype=RANGE_MEASUREMENT).reshape(-1)
        T = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_pf_sharded_set_params(self._h, C.byref(self.config_)))
        _capi.check(_capi.lib().rmclhip_pf_update_sharded(self._h, _ptr(b), len(b), _ptr(T)))
        return self.weights(0)

    def motion_update(self, T_bnew_bold, forget_rate, check_collision=True):
        """TFMotionUpdater*::update on every device's block, in place (rmclhip_pf_sharded_motion_update): pose <- pose * T_bnew_bold,
        n_meas -= forget_rate * n_meas, and with check_collision the wall-collision ray of the CPU updater"""
        T = np.ascontiguousarray(T_bnew_bold, dtype=TRANSFORM).reshape(1)
        _capi.check(_capi.lib().rmclhip_pf_sharded_set_params(self._h, C.byref(self.config_)))   # (max_n_meas of a collided particle)
        _capi.check(_capi.lib().rmclhip_pf_sharded_motion_update(self._h, _ptr(T), float(forget_rate), int(bool(check_collision))))

    def step(self, beams, Tsb, T_bnew_bold=None, forget_rate=0.0, check_collision=True, resample=None, cfg=None, seed=42, step=0):
        """one cycle of the filter node (rmcl_localization.cpp:84, 432-552) behind ONE C call: motion (skipped when T_bnew_bold is None)
        -> sensor update -> weight all-gather -> {sum, max} -> resampling (None, "gladiator" or "residual"); returns {sum, max} of the
        likelihoods after the sensor update"""
        b = np.ascontiguousarray(beams, dtype=RANGE_MEASUREMENT).reshape(-1)
        T = np.ascontiguousarray(Tsb, dtype=TRANSFORM).reshape(1)
        Tm = None if T_bnew_bold is None else np.ascontiguousarray(T_bnew_bold, dtype=TRANSFORM).reshape(1)
        cfg = cfg if cfg is not None else gladiator_config()
        st = _capi.LikelihoodStats()
        _capi.check(_capi.lib().rmclhip_pf_sharded_set_params(self._h, C.byref(self.config_)))
        _capi.check(_capi.lib().rmclhip_pf_sharded_step(self._h, None if Tm is None else _ptr(Tm), float(forget_rate), int(bool(check_collision)),
                                                       _ptr(b), len(b), _ptr(T), {None: 0, "gladiator": 1, "residual": 2}[resample],
                                                       C.byref(cfg), int(seed), int(step), C.byref(st)))
        return {"sum": st.sum, "max": st.max}

    def weights(self, rank=0):
        w = np.zeros(self.n_total, np.float32)
        _capi.check(_capi.lib().rmclhip_pf_sharded_get_weights(self._h, int(rank), _ptr(w)))
        return w

    def stats(self):
        """{sum, max} of the likelihoods: every device reduces ITS copy of the gathered weights in the single-device kernel's order
        (no collective since round 6: rmclhip_pf_allreduce_stats)"""
        st = _capi.LikelihoodStats()
        _capi.check(_capi.lib().rmclhip_pf_allreduce_stats(self._h, C.byref(st)))
        return {"sum": st.sum, "max": st.max}

    def collective_ranks(self):
        """(ranks the collective library itself reports for this communicator -- ncclCommCount --, "rccl" | "loopback")"""
        n, is_rccl = C.c_uint32(0), C.c_int(0)
        _capi.check(_capi.lib().rmclhip_comm_collective_ranks(self._comm, C.byref(n), C.byref(is_rccl)))
        return n.value, ("rccl" if is_rccl.value else "loopback")

    def set_loopback_reduce_rotation(self, first_rank):
        """TEST knob (loopback communicators): the all-reduce adds the ranks starting at first_rank (include/rmclhip_lab.h)"""
        _capi.check(_capi.lib().rmclhip_comm_loopback_set_reduce_rotation(self._comm, int(first_rank)))

    def pose_estimate(self, max_induction_particles=0xFFFFFFFF):
        e = _capi.PoseEstimate()
        _capi.check(_capi.lib().rmclhip_pf_allreduce_pose_estimate(self._h, int(min(max_induction_particles, 0xFFFFFFFF)), C.byref(e)))
        pose = np.frombuffer(bytes(bytearray(memoryview(e.pose))), dtype=TRANSFORM)[0].copy()
        return {"pose": pose, "covariance": np.array(e.covariance, dtype=np.float64).reshape(6, 6),
                "likelihood": {"mean": e.likelihood_mean, "sigma": e.likelihood_sigma, "min": e.likelihood_min, "max": e.likelihood_max},
                "trans_bb_min": np.array(e.trans_bb_min), "trans_bb_max": np.array(e.trans_bb_max), "nparticles": e.n_particles}

    def resample(self, cfg=None, seed=42, step=0, residual=False):
        """all-gather the cloud, then every device resamples its shard: the gladiator tournament (default) or the residual resampler"""
        cfg = cfg if cfg is not None else gladiator_config()
        fn = _capi.lib().rmclhip_pf_sharded_resample_residual if residual else _capi.lib().rmclhip_pf_sharded_resample
        _capi.check(fn(self._h, C.byref(cfg), int(seed), int(step)))

    def close(self):
        if self._h:
            _capi.lib().rmclhip_pf_sharded_destroy(self._h)
            self._h = C.c_void_p()
        if self._comm:
            _capi.lib().rmclhip_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
