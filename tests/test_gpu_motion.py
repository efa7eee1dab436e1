"""GPU parity of the particle-filter motion update: particle_move_and_forget_kernel
(rmcl_ros/src/rmcl/particle_motion.cu:11-34) + collision_in_between (TFMotionUpdaterCPU.cpp:17-50,207-221)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("collision", [False, True])
def test_motion_update_matches_oracle(ra, orc, ctx, meshes, collision):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    n = 5003
    poses, attrs = syn.uniform_particles(n, seed=17, bb_min=(-9.9, -9.9, 0.2, 0, 0, -math.pi), bb_max=(9.9, 9.9, 3.0, 0, 0, math.pi))
    rng = np.random.RandomState(1)
    attrs["likelihood"]["n_meas"] = rng.randint(0, 10001, n)
    attrs["likelihood"]["mean"] = rng.uniform(0, 1, n)
    attrs["likelihood"]["sigma"] = rng.uniform(0, 0.1, n)
    T_delta = T.transform_from_rpy((0.6, -0.1, 0.0), (0.0, 0.0, 0.15))          # a 60 cm odometry step
    rate = ra.combined_forget_rate(0.01, 0.001, 0.61, 0.1)
    assert 0.0 < rate < 1.0
    upd = ra.TFMotionUpdaterHip(hm, check_collision=collision)
    d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    upd.update(d_poses, d_attrs, n, T_delta, rate)
    p_ref, a_ref = poses.copy(), attrs.copy()
    m.pf_motion_update(p_ref, a_ref, T_delta, rate, collision=collision, bvh=True)
    p_gpu, a_gpu = d_poses.download(), d_attrs.download()
    assert p_gpu.tobytes() == p_ref.tobytes()                                   # same operation order: bit-exact poses
    assert a_gpu.tobytes() == a_ref.tobytes()
    killed = (a_ref["likelihood"]["n_meas"] == 10000) & (a_ref["likelihood"]["mean"] == 0)
    assert killed.any() == collision                                            # some steps cross walls / boxes
    # a zero step never collides; a big forget rate empties n_meas
    d_poses2, d_attrs2 = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    upd.update(d_poses2, d_attrs2, n, T.identity(), 1.0)
    a2 = d_attrs2.download()
    assert np.all(a2["likelihood"]["n_meas"] == 0) and np.array_equal(a2["likelihood"]["mean"], attrs["likelihood"]["mean"])
    assert np.allclose(d_poses2.download()["t"]["x"], poses["t"]["x"], atol=1e-6)   # identity step keeps the pose
