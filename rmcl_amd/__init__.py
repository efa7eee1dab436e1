"""rmcl_amd -- MI355X (gfx950) implementation of RMCL / MICP-L's ray-casting-correspondence +
pose-correction hot path.  Host-side mirror of the reference's operator interfaces over the C ABI
of librmclhip.so (include/rmclhip.h).  No CPU fallback: importing works anywhere the shared
library is built, computing needs a HIP device.
"""
from . import _capi, types, synthetic, wire  # noqa: F401
from ._capi import NoDeviceError, RmclHipError  # noqa: F401
from .micp import MICPLocalization, MICPSensor  # noqa: F401
from . import pf  # noqa: F401
from .pf import (GladiatorResamplerHip, PCDSensorUpdaterHip, ResidualResamplerHip, ShardedParticleFilterHip, TFMotionUpdaterHip,  # noqa: F401
                 beams_from_points, combined_forget_rate, sample_beams)
from .registration import (CPCHip, Context, CorrespondencesHIP, DeviceArray, HipMap, MapMap, RCCHipO1Dn,  # noqa: F401
                           RCCHipOnDn, RCCHipPinhole, RCCHipSpherical, ShardedCorrectorHip, build_bvh_host, build_bvh_host_pf, build_bvh_host_quantised,
                           flatten_scene_host, import_hip_map, import_hip_scene, statistics_p2l)

from ._capi import load_lab  # noqa: F401  (experiments library; tools/ and the `lab` tests only)

__version__ = "0.1.0"
