"""CPU tests of the oracle's gladiator resampling (resampling.cu:41-219 restated) and of its pinned random
stream: Philox4x32-10 against the published Random123 known-answer vectors."""
import numpy as np


def test_philox_known_answer_vectors(orc):
    """Random123 kat_vectors, philox4x32-10 (Salmon et al., SC'11)."""
    kat = [(([0] * 4, [0] * 2), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           (([0xffffffff] * 4, [0xffffffff] * 2), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           (([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]),
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for (ctr, key), expect in kat:
        assert list(orc.philox4x32_10(ctr, key)) == expect


def test_euler_round_trip(orc):
    rng = np.random.RandomState(0)
    for _ in range(200):
        r, p, y = rng.uniform(-3.1, 3.1), rng.uniform(-1.5, 1.5), rng.uniform(-3.1, 3.1)
        rr, pp, yy = orc.quat_to_euler(orc.euler_to_quat(r, p, y))
        assert abs(rr - r) < 2e-5 and abs(pp - p) < 2e-5 and abs(yy - y) < 2e-5
    assert orc.quat_to_euler((0.0, 0.75, 0.0, 0.75))[1] == np.float32(np.pi / 2)   # gimbal clamp


def test_gladiator_semantics(orc):
    from rmcl_amd import synthetic as syn
    n = 20000
    poses, attrs = syn.uniform_particles(n, seed=3)
    rng = np.random.RandomState(0)
    attrs["likelihood"]["mean"] = rng.uniform(0, 1, n)
    attrs["likelihood"]["n_meas"] = rng.randint(0, 10001, n)
    cfg = orc.gladiator_config()
    pn, an = orc.gladiator_resample(poses, attrs, cfg, seed=1234, step=0)
    lost = an["likelihood"]["mean"] != attrs["likelihood"]["mean"]
    assert 0.45 < lost.mean() < 0.55                                  # a random enemy is better half of the time
    assert np.all(an["likelihood"]["mean"] >= attrs["likelihood"]["mean"])
    keep = ~lost
    assert pn[keep].tobytes() == poses[keep].tobytes() and an[keep].tobytes() == attrs[keep].tobytes()
    # winners: default noise only in x, y, yaw; z / roll / pitch stay (0 noise), n_meas shrinks by >= 20 %
    i = np.nonzero(lost)[0]
    src = np.array([np.nonzero(attrs["likelihood"]["mean"] == an["likelihood"]["mean"][k])[0][0] for k in i[:200]])
    assert np.array_equal(pn["t"]["z"][i[:200]], poses["t"]["z"][src])
    assert np.all(np.abs(pn["t"]["x"][i[:200]] - poses["t"]["x"][src]) < 0.2)
    assert np.std(pn["t"]["x"][i[:200]] - poses["t"]["x"][src]) > 0.015
    assert np.all(an["likelihood"]["n_meas"][i[:200]] <= np.floor(attrs["likelihood"]["n_meas"][src] * 0.8000001))
    q = np.stack([pn["R"][k] for k in "xyzw"], 1)
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    # reproducible; other step / seed -> other draws; shard == slice of the whole
    pn2, an2 = orc.gladiator_resample(poses, attrs, cfg, seed=1234, step=0)
    assert pn2.tobytes() == pn.tobytes() and an2.tobytes() == an.tobytes()
    pn3, _ = orc.gladiator_resample(poses, attrs, cfg, seed=1234, step=1)
    assert pn3.tobytes() != pn.tobytes()
    pn4, an4 = orc.gladiator_resample(poses, attrs, cfg, seed=1234, step=0, first=777, count=1000)
    assert pn4.tobytes() == pn[777:1777].tobytes() and an4.tobytes() == an[777:1777].tobytes()
    # |t|^2 metric of the CPU reference forgets less for sub-metre noise
    _, an5 = orc.gladiator_resample(poses, attrs, orc.gladiator_config(trans_dist_metric=1, likelihood_forget_per_radian=0.0,
                                                                    min_noise_tx=0.3, min_noise_ty=0.3), 1234, 0)
    _, an6 = orc.gladiator_resample(poses, attrs, orc.gladiator_config(trans_dist_metric=0, likelihood_forget_per_radian=0.0,
                                                                    min_noise_tx=0.3, min_noise_ty=0.3), 1234, 0)
    assert an5["likelihood"]["n_meas"].sum() > an6["likelihood"]["n_meas"].sum()


def test_likelihood_stats(orc):
    from rmcl_amd import synthetic as syn
    _, attrs = syn.uniform_particles(5000, seed=1)
    attrs["likelihood"]["mean"] = np.random.RandomState(2).uniform(0, 3, 5000)
    s = orc.likelihood_stats(attrs)
    assert s["max"] == attrs["likelihood"]["mean"].max()
    assert abs(s["sum"] - attrs["likelihood"]["mean"].astype(np.float64).sum()) < 1e-3
    assert orc.likelihood_stats(attrs[:0]) == {"sum": 0.0, "max": 0.0}


def test_residual_semantics(orc):
    """ResidualResamplerCPU::update (ResidualResamplerCPU.cpp:55-203) as restated: every inserted particle descends from a particle
    whose share L / sum * N_new is >= 1, copies come in runs of floor(share) (the last run clamped), noise is inversely
    proportional to L / max, n_meas shrinks, the run is reproducible and bounded."""
    from rmcl_amd import synthetic as syn
    n = 20000
    poses, attrs = syn.uniform_particles(n, seed=3)
    rng = np.random.RandomState(0)
    L = (rng.uniform(0, 1, n) ** 3).astype(np.float32)
    attrs["likelihood"]["mean"] = L
    attrs["likelihood"]["n_meas"] = rng.randint(100, 10001, n)
    cfg = orc.gladiator_config()
    pn, an, filled, draws = orc.residual_resample(poses, attrs, cfg, seed=99, step=0)
    assert filled == n and draws >= n // 4
    share = np.floor(L.astype(np.float64) / L.astype(np.float64).sum() * n)
    # every new particle carries the likelihood of an old particle with share >= 1
    src_of = {float(v): i for i, v in enumerate(L)}
    src = np.array([src_of[float(v)] for v in an["likelihood"]["mean"]])
    assert np.all(share[src] >= 1)
    # runs: consecutive slots with the same source; every run but the last has exactly floor(share) members
    edges = np.nonzero(np.diff(src))[0] + 1
    runs = np.split(np.arange(n), edges)
    for r in runs[:-1]:
        assert len(r) % int(share[src[r[0]]]) == 0            # (the same particle may be drawn twice in a row)
    assert len(runs[-1]) <= share[src[-1]] or len(runs[-1]) % int(share[src[-1]]) <= share[src[-1]]
    # noise width ~ min_noise / (L / max): the best particles move least; z / roll / pitch have no noise by default
    dx = pn["t"]["x"] - poses["t"]["x"][src]
    rel = L[src] / L.max()
    assert np.array_equal(pn["t"]["z"], poses["t"]["z"][src])
    good, poor = rel > 0.7, rel < 0.35
    assert good.sum() > 500 and poor.sum() > 500 and np.std(dx[poor]) > 1.8 * np.std(dx[good])
    assert abs(np.std(dx * rel) - 0.03) < 0.003                # = min_noise_tx
    # n_meas: reduced by forget_per_radian^l2norm (~0.2: rmagine's l2norm of a unit quaternion is ~1) times forget_per_meter^|dt|^2
    assert np.all(an["likelihood"]["n_meas"] <= np.floor(attrs["likelihood"]["n_meas"][src] * 0.2000001))
    q = np.stack([pn["R"][k] for k in "xyzw"], 1)
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    # reproducible; another step -> other draws; a different size of the new cloud; the draw budget is honoured
    pn2, an2, f2, d2 = orc.residual_resample(poses, attrs, cfg, seed=99, step=0)
    assert pn2.tobytes() == pn.tobytes() and an2.tobytes() == an.tobytes() and (f2, d2) == (filled, draws)
    pn3, _, _, _ = orc.residual_resample(poses, attrs, cfg, seed=99, step=1)
    assert pn3.tobytes() != pn.tobytes()
    _, an4, f4, _ = orc.residual_resample(poses, attrs, cfg, seed=99, step=0, n_new=3 * n + 7)
    assert f4 == 3 * n + 7
    _, _, f5, d5 = orc.residual_resample(poses, attrs, cfg, seed=99, step=0, max_draws=100)
    assert d5 == 100 and f5 < n
    # uniform weights: every share is 1.0 or 0.99999...: the loop either copies one particle per draw or never terminates in
    # the reference; bounded here
    attrs["likelihood"]["mean"] = 0.25
    _, _, f6, d6 = orc.residual_resample(poses, attrs, cfg, seed=99, step=0, max_draws=5 * n)
    assert (f6 == n and d6 == n) or (f6 == 0 and d6 == 5 * n)
