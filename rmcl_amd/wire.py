"""Wire-format adapters: the ROS message layouts the reference's sensors and updaters
consume, restated as plain field mappings so a host without ROS bindings (or a bag reader) can drive the operators.

Reference: rmcl_ros/src/util/conversions.cpp:22-120 (ScanInfo / CameraInfo / DepthInfo / O1DnInfo / OnDnInfo ->
rmagine models), :869-1002 (PointCloud2 -> O1Dn), PCDSensorUpdaterEmbree.cpp:290-327 (beam sampling from a
PointCloud2).  The device-side PointCloud2 path is RCCHipO1Dn.setInputPointCloud2.
"""
import numpy as np

from . import _capi
from .pf import beams_from_points

FLOAT32, FLOAT64 = 7, 8   # sensor_msgs/PointField datatypes


def spherical_from_scan_info(info):
    """rmcl_msgs/ScanInfo -> rm::SphericalModel (conversions.cpp:22-34). `info`: mapping or object with
    phi_min, phi_inc, phi_n, theta_min, theta_inc, theta_n, range_min, range_max."""
    g = (lambda k: info[k]) if isinstance(info, dict) else (lambda k: getattr(info, k))
    m = _capi.SphericalModel()
    m.phi.min, m.phi.inc, m.phi.size = g("phi_min"), g("phi_inc"), g("phi_n")
    m.theta.min, m.theta.inc, m.theta.size = g("theta_min"), g("theta_inc"), g("theta_n")
    m.range.min, m.range.max = g("range_min"), g("range_max")
    return m


def pinhole_from_camera_info(width, height, k, range_min=0.0, range_max=1e30):
    """sensor_msgs/CameraInfo -> rm::PinholeModel (conversions.cpp:36-46): f = (K[0], K[4]), c = (K[2], K[5]);
    the message carries no range, so the caller supplies it.  Returns RCCHipPinhole.setModel keyword arguments."""
    k = [float(x) for x in np.asarray(k).reshape(-1)]
    if len(k) != 9:
        raise ValueError("CameraInfo.k must have 9 entries")
    return dict(width=int(width), height=int(height), range_min=float(range_min), range_max=float(range_max),
                fx=k[0], fy=k[4], cx=k[2], cy=k[5])


def pinhole_from_depth_info(info):
    """rmcl_msgs/DepthInfo -> rm::PinholeModel (conversions.cpp:48-60)."""
    g = (lambda k: info[k]) if isinstance(info, dict) else (lambda k: getattr(info, k))
    return dict(width=int(g("width")), height=int(g("height")), range_min=float(g("range_min")),
                range_max=float(g("range_max")), fx=float(g("fx")), fy=float(g("fy")), cx=float(g("cx")), cy=float(g("cy")))


def xyz_from_pointcloud2(data, n_points, point_step, offset_x, offset_y, offset_z, datatype=FLOAT32):
    """the x / y / z fields of a PointCloud2 byte buffer as an [n, 3] float32 array (host side; used for beam
    sampling, where only `samples` random points are touched)."""
    buf = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) \
        else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    if datatype not in (FLOAT32, FLOAT64):
        raise ValueError("Field X has unknown DataType. Check Topic of PC")   # conversions.cpp:994
    ft = "<f4" if datatype == FLOAT32 else "<f8"
    size = 4 if datatype == FLOAT32 else 8
    if n_points and (n_points - 1) * point_step + max(offset_x, offset_y, offset_z) + size > buf.size:
        raise ValueError("cloud data shorter than its layout")
    out = np.empty((n_points, 3), np.float32)
    for c, off in enumerate((offset_x, offset_y, offset_z)):
        col = np.lib.stride_tricks.as_strided(buf[off:], shape=(n_points, size), strides=(point_step, 1))
        out[:, c] = np.ascontiguousarray(col).view(ft).reshape(-1).astype(np.float32)
    return out


def sample_beams_pointcloud2(data, n_points, point_step, offset_x, offset_y, offset_z, samples, seed, datatype=FLOAT32):
    """PCDSensorUpdaterEmbree.cpp:290-327 on the raw message bytes (cloud treated as n_points x 1 like the reference's
    `random_point_id * point_step` addressing), through the C ABI sampler (rmclhip_pf_sample_beams_pointcloud2)."""
    from .pf import sample_beams_pointcloud2 as _sample
    return _sample(data, n_points, 1, point_step, point_step * n_points, offset_x, offset_y, offset_z, samples, seed, datatype)
