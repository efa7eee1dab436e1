"""POD types of the hot path as numpy dtypes, plus the rmagine-style transform algebra
(evaluated by librmclhip's host entry points, so Python never re-implements arithmetic).

Layouts: rmagine::Transform = {Quaternion{x,y,z,w}, Vector{x,y,z}, uint32 stamp} (32 B,
rmcl_ros/src/nodes/rmcl_localization.cpp:245-249); rmcl::ParticleAttributes (36 B,
ParticleAttributes.hpp:18-34); rmcl::RangeMeasurement (64 B, RangeMeasurement.hpp:10-21).
"""
import ctypes as C
import math

import numpy as np

from . import _capi

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

MAX_N_MEAS = 10000  # ParticleAttributes.hpp:34


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def transform(q=(0.0, 0.0, 0.0, 1.0), t=(0.0, 0.0, 0.0)):
    """Transform from quaternion (x,y,z,w) and translation."""
    T = np.zeros((), dtype=TRANSFORM)
    T["R"]["x"], T["R"]["y"], T["R"]["z"], T["R"]["w"] = q
    T["t"]["x"], T["t"]["y"], T["t"]["z"] = t
    return T


def identity():
    return transform()


def euler_to_quat(roll, pitch, yaw):
    """rmagine EulerAngles -> Quaternion (ZYX), evaluated in double then rounded to f32."""
    cr, sr = math.cos(roll / 2), math.sin(roll / 2)
    cp, sp = math.cos(pitch / 2), math.sin(pitch / 2)
    cy, sy = math.cos(yaw / 2), math.sin(yaw / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy)


def transform_from_rpy(t, rpy):
    return transform(euler_to_quat(*rpy), t)


def _one(T, dtype):
    return np.ascontiguousarray(T, dtype=dtype).reshape(1)


def mult(a, b):
    """Transform::operator* (e.g. micp_localization.cpp:963)."""
    a, b, out = _one(a, TRANSFORM), _one(b, TRANSFORM), np.zeros(1, dtype=TRANSFORM)
    _capi.check(_capi.lib().rmclhip_transform_mult(_ptr(a), _ptr(b), _ptr(out)))
    return out[0].copy()


def inv(a):
    """Transform::operator~ (micp_localization.cpp:926)."""
    a, out = _one(a, TRANSFORM), np.zeros(1, dtype=TRANSFORM)
    _capi.check(_capi.lib().rmclhip_transform_inv(_ptr(a), _ptr(out)))
    return out[0].copy()


def cross_statistics_identity():
    return np.zeros((), dtype=CROSS_STATISTICS)


def cross_statistics_merge(a, b):
    """CrossStatistics::operator+= (micp_localization.cpp:936-937)."""
    a, b, out = _one(a, CROSS_STATISTICS), _one(b, CROSS_STATISTICS), np.zeros(1, dtype=CROSS_STATISTICS)
    _capi.check(_capi.lib().rmclhip_cross_statistics_merge(_ptr(a), _ptr(b), _ptr(out)))
    return out[0].copy()


def cross_statistics_transform(T, s):
    """Transform * CrossStatistics (MICPSensor.hpp:182)."""
    T, s, out = _one(T, TRANSFORM), _one(s, CROSS_STATISTICS), np.zeros(1, dtype=CROSS_STATISTICS)
    _capi.check(_capi.lib().rmclhip_cross_statistics_transform(_ptr(T), _ptr(s), _ptr(out)))
    return out[0].copy()


def umeyama_transform(s):
    """rm::umeyama_transform (micp_localization.cpp:952-953)."""
    s, out = _one(s, CROSS_STATISTICS), np.zeros(1, dtype=TRANSFORM)
    _capi.check(_capi.lib().rmclhip_umeyama_transform(_ptr(s), _ptr(out)))
    return out[0].copy()


def spherical_model(phi_min, phi_inc, phi_n, theta_min, theta_inc, theta_n, range_min, range_max):
    """rmagine::SphericalModel (fields: rmcl_ros/src/util/conversions.cpp:22-34)."""
    m = _capi.SphericalModel()
    m.phi.min, m.phi.inc, m.phi.size = phi_min, phi_inc, phi_n
    m.theta.min, m.theta.inc, m.theta.size = theta_min, theta_inc, theta_n
    m.range.min, m.range.max = range_min, range_max
    return m


def pf_params(dist_sigma=2.0, real_hit_sim_miss_error=100.0, real_miss_sim_hit_error=100.0,
              real_miss_sim_miss_error=0.0, range_min=0.05, range_max=80.0, max_n_meas=MAX_N_MEAS,
              correspondence_type=0):
    """sensor_update.* defaults of PCDSensorUpdaterEmbree.cpp:122-134."""
    p = _capi.PFParams()
    p.dist_sigma = dist_sigma
    p.real_hit_sim_miss_error = real_hit_sim_miss_error
    p.real_miss_sim_hit_error = real_miss_sim_hit_error
    p.real_miss_sim_miss_error = real_miss_sim_miss_error
    p.sensor_range.min, p.sensor_range.max = range_min, range_max
    p.max_n_meas = max_n_meas
    p.correspondence_type = correspondence_type
    return p


def gladiator_config(min_noise_tx=0.03, min_noise_ty=0.03, min_noise_tz=0.0, min_noise_roll=0.0, min_noise_pitch=0.0,
                     min_noise_yaw=0.01, likelihood_forget_per_meter=0.3, likelihood_forget_per_radian=0.2,
                     trans_dist_metric=0):
    """resampling.* parameters with the defaults of GladiatorResamplerGPU::updateParams (GladiatorResamplerGPU.cpp:34-44);
    trans_dist_metric 0 = |t| like the reference's GPU kernel, 1 = |t|^2 like its CPU implementation."""
    return _capi.GladiatorConfig(min_noise_tx, min_noise_ty, min_noise_tz, min_noise_roll, min_noise_pitch,
                                 min_noise_yaw, likelihood_forget_per_meter, likelihood_forget_per_radian,
                                 trans_dist_metric)
