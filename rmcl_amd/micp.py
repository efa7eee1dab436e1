"""MICP-L correction loop over the HIP correspondence operators (host side, ROS-free).

Restates the callers of the hot path so that parity tests read like the reference:
  MICPSensor_<MemT>::findCorrespondences / computeCrossStatistics
      rmcl_ros/include/rmcl_ros/micpl/MICPSensor.hpp:146-184
  MICPLocalizationNode::correctOnce (inner ICP loop + convergence heuristic)
      rmcl_ros/src/nodes/micp_localization.cpp:900-1007
Arithmetic is delegated to librmclhip (types.mult / inv / cross_statistics_* / umeyama_transform).
"""
import math

import numpy as np

from . import types as T


class MICPSensor:
    """MICPSensor_<VRAM_HIP>: one sensor = one correspondence operator + its frames."""

    def __init__(self, name, correspondences, Tsb=None, Tbo=None, merge_weight_multiplier=1.0):
        self.name = name
        self.correspondences_ = correspondences
        self.Tsb = T.identity() if Tsb is None else Tsb
        self.Tbo = T.identity() if Tbo is None else Tbo
        self.Tom = T.identity()
        self.merge_weight_multiplier = merge_weight_multiplier
        self.total_dataset_measurements = 0
        self.valid_dataset_measurements = 0
        self.correspondences_.setTsb(self.Tsb)

    def setTom(self, Tom):
        self.Tom = Tom

    def findCorrespondences(self):
        # MICPSensor.hpp:146-151
        Tbm = T.mult(self.Tom, self.Tbo)
        self.correspondences_.find(Tbm)
        self.correspondences_.outdated = False

    def computeCrossStatistics(self, T_bnew_bold, convergence_progress=0.0):
        # MICPSensor.hpp:159-184
        T_snew_sold = T.mult(T.mult(T.inv(self.Tsb), T_bnew_bold), self.Tsb)
        stats_s = self.correspondences_.computeCrossStatistics(T_snew_sold, convergence_progress)
        return T.cross_statistics_transform(self.Tsb, stats_s)


class MICPLocalization:
    """The state + correctOnce() of MICPLocalizationNode, minus ROS."""

    def __init__(self, sensors, optimization_iterations=5, adaptive_max_dist=True, disable_correction=False):
        self.sensors_vec_ = list(sensors)
        self.optimization_iterations_ = optimization_iterations  # default 5, micp_localization.cpp:129
        self.adaptive_max_dist_ = adaptive_max_dist              # :139
        self.disable_correction_ = disable_correction
        self.Tom_ = T.identity()
        self.convergence_progress_ = 0.0
        self.correction_stats_latest_ = {}

    def _device_loop(self, Tom):
        """the inner loop of correctOnce for all sensors resident on the device (rmclhip_micp_correct_once): finds, per-sensor
        reductions, merges, solve and compositions without a host round trip per sensor and iteration"""
        import ctypes as C
        from . import _capi
        from .types import CROSS_STATISTICS, TRANSFORM, _ptr
        n = len(self.sensors_vec_)
        handles = (C.c_void_p * n)(*[s.correspondences_._h for s in self.sensors_vec_])
        Tbo = np.array([s.Tbo for s in self.sensors_vec_], dtype=TRANSFORM)
        w = np.array([float(s.merge_weight_multiplier) for s in self.sensors_vec_], dtype=np.float64)
        for s in self.sensors_vec_:
            s.setTom(Tom)
            s.correspondences_.setTsb(s.Tsb)
            s.correspondences_._push_params()
            s.correspondences_.outdated = False
            s.correspondences_._last_nposes = 1
        Tin = np.ascontiguousarray(Tom, dtype=TRANSFORM).reshape(1)
        Tout, merged = np.zeros(1, TRANSFORM), np.zeros(1, CROSS_STATISTICS)
        _capi.check(_capi.lib().rmclhip_micp_correct_once(handles, n, _ptr(Tin), _ptr(Tbo), _ptr(w), int(self.optimization_iterations_),
                                                          float(self.convergence_progress_), _ptr(Tout), _ptr(merged)))
        return Tout[0].copy(), merged[0].copy()

    def correctOnce(self, record=None, device_loop=False):
        """micp_localization.cpp:856-1016. `record`, if a list, receives T_onew_oold after each iteration.
        device_loop=True runs the inner loop (finds + iterations) on the device for all sensors at once; it sets every
        sensor's Tom and pushes its parameters, but does NOT call sensor.findCorrespondences() (sensor-side effects of that
        method, e.g. visualisation hooks of a subclass, do not run) and cannot fill `record`."""
        Tom = self.Tom_
        valid_measurements = sum(s.valid_dataset_measurements for s in self.sensors_vec_)
        if device_loop and not self.disable_correction_ and self.optimization_iterations_ > 0:
            # the device loop returns only the final transform and does not call sensor.findCorrespondences(): per-iteration
            # records and more sensors than the device block holds need the host loop -- say so instead of returning less
            if record is not None:
                raise ValueError("correctOnce(record=..., device_loop=True): the device-resident loop does not return per-iteration "
                                 "transforms; use the host loop (device_loop=False) to record them")
            if len(self.sensors_vec_) > 8:
                raise ValueError("correctOnce(device_loop=True): at most 8 sensors (rmclhip_micp_correct_once); use the host loop")
            T_onew_oold, Cmerged_o = self._device_loop(Tom)
            return self._finish(Tom, T_onew_oold, Cmerged_o, valid_measurements)
        for s in self.sensors_vec_:
            s.setTom(Tom)
            s.findCorrespondences()
        T_onew_oold = T.identity()
        Cmerged_o = T.cross_statistics_identity()
        for _ in range(self.optimization_iterations_):
            Cmerged_o = T.cross_statistics_identity()
            Cmerged_weighted_o = T.cross_statistics_identity()
            for s in self.sensors_vec_:
                T_bnew_bold = T.mult(T.mult(T.inv(s.Tbo), T_onew_oold), s.Tbo)           # :926
                Cs_b = s.computeCrossStatistics(T_bnew_bold, self.convergence_progress_)  # :928
                Cs_o = T.cross_statistics_transform(s.Tbo, Cs_b)                          # :931
                Cs_weighted_o = Cs_o.copy()
                # :934 -- `n_meas *= double` on an unsigned count truncates
                Cs_weighted_o["n_meas"] = np.uint32(int(float(Cs_weighted_o["n_meas"]) * s.merge_weight_multiplier))
                Cmerged_o = T.cross_statistics_merge(Cmerged_o, Cs_o)                     # :936
                Cmerged_weighted_o = T.cross_statistics_merge(Cmerged_weighted_o, Cs_weighted_o)
            if self.disable_correction_:
                break
            T_inner = T.umeyama_transform(Cmerged_weighted_o)                             # :952
            T_onew_oold = T.mult(T_onew_oold, T_inner)                                    # :963
            if record is not None:
                record.append(T_onew_oold.copy())
        return self._finish(Tom, T_onew_oold, Cmerged_o, valid_measurements)

    def _finish(self, Tom, T_onew_oold, Cmerged_o, valid_measurements):
        T_onew_map = T.mult(Tom, T_onew_oold)                                             # :972
        n_meas = int(Cmerged_o["n_meas"])
        if not self.disable_correction_ and n_meas > 0:
            q = T_onew_map["R"]
            nrm = np.float32(math.sqrt(float(q["x"]) ** 2 + float(q["y"]) ** 2 + float(q["z"]) ** 2 + float(q["w"]) ** 2))
            for k in "xyzw":
                T_onew_map["R"][k] = np.float32(q[k] / nrm)                               # :983 normalizeInplace
            self.Tom_ = T_onew_map
        # convergence heuristic, :986-1007
        if n_meas == 0 or not self.adaptive_max_dist_ or valid_measurements == 0:
            self.convergence_progress_ = 0.0
        else:
            t = T_onew_map["t"]
            trans_force = math.sqrt(float(t["x"]) ** 2 + float(t["y"]) ** 2 + float(t["z"]) ** 2)
            trans_progress = 1.0 / math.exp(10.0 * trans_force)
            qscalar = float(T_onew_map["R"]["w"])  # R.dot(Quaternion::Identity())
            rot_progress = qscalar * qscalar
            match_ratio = float(n_meas) / float(valid_measurements)
            self.convergence_progress_ = float(np.float32(trans_progress * rot_progress * match_ratio))
        cov = Cmerged_o["covariance"]
        self.correction_stats_latest_ = dict(valid_matches=n_meas, cov_trace=float(cov[0] + cov[4] + cov[8]),
                                             valid_measurements=valid_measurements)
        return T_onew_oold
