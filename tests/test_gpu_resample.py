"""GPU parity of the gladiator resampler: gladiator_resample_kernel + simple_stats_kernel
(rmcl_ros/src/rmcl/resampling.cu:41-219) vs the oracle on the pinned Philox stream.  Who wins, n_meas and copied
records bit-exact; perturbed poses within 1e-6."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(n, seed):
    from rmcl_amd import synthetic as syn
    poses, attrs = syn.uniform_particles(n, seed=seed, bb_min=(-9, -9, 0.2, -0.2, -0.2, -math.pi), bb_max=(9, 9, 3.0, 0.2, 0.2, math.pi))
    rng = np.random.RandomState(seed)
    attrs["likelihood"]["mean"] = rng.uniform(0, 1, n)
    attrs["likelihood"]["sigma"] = rng.uniform(0, 0.1, n)
    attrs["likelihood"]["n_meas"] = rng.randint(0, 10001, n)
    poses["stamp"] = rng.randint(0, 1 << 30, n)
    return poses, attrs


@pytest.mark.parametrize("n,metric", [(100003, 0), (4097, 1), (1, 0)])
def test_gladiator_matches_oracle(ra, orc, ctx, n, metric):
    from rmcl_amd import types as T
    poses, attrs = _cloud(n, 5)
    kw = dict(min_noise_tz=0.01, min_noise_roll=0.005, min_noise_pitch=0.005, trans_dist_metric=metric)
    rs = ra.GladiatorResamplerHip(ctx, seed=0xDEADBEEF12345)
    rs.config = T.gladiator_config(**kw)
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n)
    for step in range(2):                                  # the step counter advances the stream
        res = rs.update(d_p, d_a, d_pn, d_an, n)
        assert res == {"n_particles": n}
        pn_ref, an_ref = orc.gladiator_resample(poses, attrs, orc.gladiator_config(**kw), seed=0xDEADBEEF12345, step=step)
        pn, an = d_pn.download(), d_an.download()
        assert an.tobytes() == an_ref.tobytes()            # winners, likelihoods, n_meas: bit-exact
        assert np.array_equal(pn["stamp"], pn_ref["stamp"])
        for k in "xyz":
            assert np.allclose(pn["t"][k], pn_ref["t"][k], rtol=0, atol=1e-6)
        for k in "xyzw":
            assert np.allclose(pn["R"][k], pn_ref["R"][k], rtol=0, atol=1e-6)
        same = (pn.view(np.uint8).reshape(n, 32) == pn_ref.view(np.uint8).reshape(n, 32)).all(1)
        assert same.mean() > 0.999                         # double-evaluated transcendentals: near always bit-equal
    # a shard of the champions against the whole cloud == slice of the full tournament
    if n > 1000:
        rs.step = 0
        d_ps, d_as = ra.DeviceArray(ctx, T.TRANSFORM, 500), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, 500)
        rs.update(d_p, d_a, d_ps, d_as, n, first=1234, count=500)
        pn0, an0 = orc.gladiator_resample(poses, attrs, orc.gladiator_config(**kw), seed=0xDEADBEEF12345, step=0, first=1234, count=500)
        assert d_as.download().tobytes() == an0.tobytes()
    with pytest.raises(ra.RmclHipError):
        rs.update(d_p, d_a, d_p, d_a, n)                   # in place is refused
    with pytest.raises(ra.RmclHipError):
        rs.update(d_p, d_a, d_pn, d_an, n, first=n, count=1)


def test_likelihood_stats_match_oracle(ra, orc, ctx):
    for n in (1, 1000, 300007):
        _, attrs = _cloud(n, 9)
        rs = ra.GladiatorResamplerHip(ctx)
        s = rs.compute_stats(ra.DeviceArray.from_host(ctx, attrs), n)
        r = orc.likelihood_stats(attrs)
        assert s["max"] == r["max"]
        assert abs(s["sum"] - r["sum"]) <= 1e-6 * abs(r["sum"])


@pytest.mark.parametrize("n,n_new,power", [(100003, 100003, 3.0), (4097, 12000, 1.0), (50000, 777, None), (3, 5, 1.0)])
def test_residual_matches_oracle(ra, orc, ctx, n, n_new, power):
    """rmclhip_resampler_residual (ResidualResamplerCPU.cpp:55-203) -- three data-parallel passes -- against the oracle's statement-by-
    statement restatement of the sequential loop on the same pinned stream: source particle of every slot, likelihoods, n_meas and
    the number of loop iterations bit-exact, perturbed poses within 1e-6; a shard of the slots == a slice of the whole."""
    from rmcl_amd import types as T
    poses, attrs = _cloud(n, 11)
    rng = np.random.RandomState(12)
    if power is None:       # a converged filter shrinking its cloud: 25 particles hold nearly all the weight
        L = np.full(n, 1e-5, np.float32)
        L[rng.choice(n, 25, replace=False)] = rng.uniform(0.5, 1.0, 25)
    else:
        L = (rng.uniform(0.05, 1, n) ** power).astype(np.float32)
    attrs["likelihood"]["mean"] = L
    kw = dict(min_noise_tz=0.01, min_noise_roll=0.005, min_noise_pitch=0.005)
    rs = ra.ResidualResamplerHip(ctx, seed=0xC0FFEE1234567)
    rs.config = T.gladiator_config(**kw)
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n_new), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n_new)
    for step in range(2):
        res = rs.update(d_p, d_a, d_pn, d_an, n, n_new)
        assert res == {"n_particles": n_new}
        pn_ref, an_ref, filled, draws = orc.residual_resample(poses, attrs, orc.gladiator_config(**kw), seed=0xC0FFEE1234567, step=step,
                                                             n_new=n_new)
        assert filled == n_new and rs.last_draws == draws
        pn, an = d_pn.download(), d_an.download()
        assert an.tobytes() == an_ref.tobytes()
        assert np.array_equal(pn["stamp"], pn_ref["stamp"])
        for k in "xyz":
            assert np.allclose(pn["t"][k], pn_ref["t"][k], rtol=0, atol=1e-6)
        for k in "xyzw":
            assert np.allclose(pn["R"][k], pn_ref["R"][k], rtol=0, atol=1e-6)
        same = (pn.view(np.uint8).reshape(n_new, 32) == pn_ref.view(np.uint8).reshape(n_new, 32)).all(1)
        assert same.mean() > 0.99
    if n_new > 1000:
        rs.step = 0
        d_ps, d_as = ra.DeviceArray(ctx, T.TRANSFORM, 500), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, 500)
        rs.update(d_p, d_a, d_ps, d_as, n, n_new, first=1234, count=500)
        _, an0, _, _ = orc.residual_resample(poses, attrs, orc.gladiator_config(**kw), seed=0xC0FFEE1234567, step=0, n_new=n_new)
        assert d_as.download().tobytes() == an0[1234:1734].tobytes()
    with pytest.raises(ra.RmclHipError):
        rs.update(d_p, d_a, d_p, d_a, n)                       # in place is refused
    with pytest.raises(ra.RmclHipError):
        rs.update(d_p, d_a, d_pn, d_an, n, n_new, first=n_new, count=1)
    rs.close()


def test_residual_refuses_inputs_the_reference_would_hang_on(ra, orc, ctx):
    from rmcl_amd import types as T
    poses, attrs = _cloud(1000, 13)
    rs = ra.ResidualResamplerHip(ctx)
    d_p = ra.DeviceArray.from_host(ctx, poses)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, 10), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, 10)
    attrs["likelihood"]["mean"] = 0.0
    with pytest.raises(ra.RmclHipError, match="sum to zero"):
        rs.update(d_p, ra.DeviceArray.from_host(ctx, attrs), d_pn, d_an, 1000, 10)
    attrs["likelihood"]["mean"] = np.random.RandomState(1).uniform(0.5, 1, 1000)   # 10 slots for 1000 similar particles: every share < 1
    with pytest.raises(ra.RmclHipError, match="truncates to 0"):
        rs.update(d_p, ra.DeviceArray.from_host(ctx, attrs), d_pn, d_an, 1000, 10)
    rs.close()
