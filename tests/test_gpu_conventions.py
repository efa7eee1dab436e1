"""GPU twins of tests/test_conventions.py: the HIP reduction reads a model entry only where model.mask > 0 and is even in
the normal's sign, so neither the miss payload (NaN here, anything in rmagine) nor the normal-flip convention can change
CrossStatistics / the pose delta (MICPSensorCPU.cpp:70-84)."""
import numpy as np
import pytest

import oracle_micp as om
from test_gpu_reduce import _stats_close, _transform_close

pytestmark = pytest.mark.gpu


def test_gpu_statistics_ignore_miss_payload_and_normal_sign(ra, orc, ctx, meshes):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    truth = T.transform_from_rpy((1.0, 2.0, 1.5), (0.0, 0.1, -0.3))
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, T.identity(), truth, bvh=True)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.params.max_dist = 1.0
    rcc.adaptive_max_dist_min = 1.0
    rcc.find(est)
    s_gpu = rcc.computeCrossStatistics(T.identity(), 0.0)
    sim = m.simulate_spherical(model, T.identity(), est, bvh=True)
    miss = sim["hits"] == 0
    assert miss.any()
    pts, nrm = sim["points"].copy(), sim["normals"].copy()
    pts[miss] = (123.0, -45.0, 6.0)                       # garbage instead of NaN at the misses
    rng = np.random.RandomState(0)
    nrm *= np.where(rng.rand(len(nrm)) < 0.5, -1.0, 1.0).astype(np.float32)[:, None]   # arbitrary normal orientation
    nrm[miss] = (0.0, 0.0, 1.0)
    ref = orc.statistics_p2l_f64(T.identity(), ds, mask, pts, nrm, sim["hits"], 1.0)
    _stats_close(s_gpu, ref)
    _transform_close(T.umeyama_transform(s_gpu), orc.umeyama(orc.statistics_p2l_exact(T.identity(), ds, mask, pts, nrm, sim["hits"], 1.0)), 1e-5)
    rcc.close()
