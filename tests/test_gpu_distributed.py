"""The RCCL path on the GPU that is there: torch.distributed backend "nccl" (= RCCL) at world size 1 drives the REAL
PCDSensorUpdaterHip through rmcl_amd.distributed.ShardedSensorUpdate / ShardedResample and must reproduce the
unsharded update bit for bit (the sharded classes had never executed on a GPU).
World sizes 2 and 3 (ragged) run on CPU with gloo in tests/test_distributed_cpu.py; 8 GPUs are the driver's to launch."""
import math
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_close_rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_world1_sharded_update_equals_unsharded():
    """runs in a fresh process: torch (which bundles its own HIP runtime) has to initialise the device BEFORE
    librmclhip.so does -- the order bench.py uses -- and the pytest process has already created HIP contexts."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def _main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.cuda.init()
    import rmcl_amd as ra
    from rmcl_amd import distributed as D, synthetic as syn, types as T
    ctx = ra.Context(0)
    meshes = {"room30k": syn.noisy_room(30000)}.__getitem__
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        v, f = meshes("room30k")
        hm = ra.import_hip_map(ctx, v, f)
        n = 3001
        poses, attrs = syn.uniform_particles(n, seed=21, bb_min=(-8, -8, 0.2, 0, 0, -math.pi), bb_max=(8, 8, 3, 0, 0, math.pi))
        beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::4] * np.float32(5.0))
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.setInput(beams, syn.tsb_offset())
        # unsharded
        d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
        upd.update(d_p, d_a)
        ref = d_a.download()
        # sharded (one shard = everything) + the RCCL all-gather of the weights
        d_a2 = ra.DeviceArray.from_host(ctx, attrs)
        sh = D.ShardedSensorUpdate(upd, n, 0, 1)
        w_local = torch.empty(n, dtype=torch.float32, device="cuda")
        for _ in range(2):
            d_a2.upload(attrs)
            gathered = sh.update(d_p, d_a2, w_local)
        torch.cuda.synchronize()
        assert gathered.is_cuda and gathered.numel() == n
        assert np.array_equal(gathered.cpu().numpy(), ref["likelihood"]["mean"])
        assert d_a2.download().tobytes() == ref.tobytes()
        ssum, smax = D.allreduce_sum_max(w_local)
        assert abs(ssum - float(ref["likelihood"]["mean"].astype(np.float64).sum())) < 1e-6 * max(1.0, abs(ssum))
        assert smax == float(ref["likelihood"]["mean"].max())
        # round 6: {sum, max} from the gathered vector on this rank, by the single-GPU kernel: the single-GPU statistics bit for bit
        rsm = ra.GladiatorResamplerHip(ctx)
        gsum, gmax = D.gathered_sum_max(gathered[:n], rsm.compute_stats_weights)
        st1 = rsm.compute_stats(d_a, n)
        assert gsum == st1["sum"] and gmax == st1["max"]
        rsm.close()
        # record all-gather (the distributed tournament's exchange) through RCCL
        rec = torch.from_numpy(poses.view(np.uint8).reshape(n, 32).copy()).cuda()
        allp = D.allgather_records(rec, n)
        assert allp.shape == (n, 32) and allp.cpu().numpy().tobytes() == poses.tobytes()
        upd.close()
        print("RCCL_WORLD1_OK", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    _main()


def _trace(ra, on):
    import ctypes as C
    buf = C.create_string_buffer(1 << 16)
    ra._capi.check(ra._capi.lib().rmclhip_debug_trace(int(on), buf, len(buf)))
    return buf.value.decode()


def _phases(trace):
    """{label: [tokens]} of a rmclhip_debug_trace recording"""
    out, cur = {}, None
    for tok in trace.split():
        if tok.endswith(":"):
            cur = tok[:-1]
            out.setdefault(cur, [])
        elif cur is not None:
            out[cur].append(tok)
    return out


def _enqueues_precede_waits(tokens, world):
    """every maximal run of E tokens covers all ranks before the first W of the run that follows it: E0 E1 .. W0 W1 .., never E0 W0 E1 W1"""
    i, ok, seen_block = 0, True, False
    while i < len(tokens):
        es = []
        while i < len(tokens) and tokens[i][0] == "E":
            es.append(tokens[i]); i += 1
        ws = []
        while i < len(tokens) and tokens[i][0] == "W":
            ws.append(tokens[i]); i += 1
        if es:
            seen_block = True
            ok &= len(ws) == world            # one wait per rank, after the enqueues
            ok &= len(set(es)) == len(es)     # a rank is enqueued once per phase
    return ok and seen_block


@pytest.mark.parametrize("devices,loopback", [((0,), False), ((0,), True), ((0, 0), True), ((0, 0, 0, 0), True), ((0,) * 8, True)],
                         ids=["rccl-1", "loopback-1", "loopback-2", "loopback-4", "loopback-8"])
def test_c_abi_sharded_particle_filter_on_one_device(ra, orc, ctx, meshes, devices, loopback):
    """(round 4: also with the in-process LOOPBACK communicator at 2 and 4 ranks on device 0 -- rmclhip_comm_create_loopback -- so that
    the ndev > 1 branches of every sharded entry point execute on this box and are held to the same results; the recorded call
    sequence must enqueue every rank's part of a phase before the host waits for any.)
    multi-GPU behind the C ABI (rmclhip_comm_create = RCCL ncclCommInitAll in ONE process, the shape of the reference's
    single-process node, rmcl_localization.cpp:482-552) at ndev = 1 on the GPU that is there: sharded sensor update + weight
    all-gather == unsharded update; all-reduced {sum, max}; the pose estimate (Markley mean + 6x6 covariance,
    rmcl_localization.cpp:642-731) vs the oracle's double-precision restatement; distributed tournament == single-GPU one."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    hm = ra.import_hip_map(ctx, v, f)
    n = 4096
    poses, attrs = syn.uniform_particles(n, seed=5, bb_min=(-8, -8, 0.2, 0, 0, -math.pi), bb_max=(8, 8, 3, 0, 0, math.pi))
    # a cloud with some structure: half of the particles clustered around one pose (a converging filter)
    rng = np.random.RandomState(2)
    c = T.transform_from_rpy((1.0, -2.0, 1.2), (0.0, 0.0, 0.7))
    for i in range(0, n, 2):
        poses[i] = T.mult(c, T.transform_from_rpy(tuple(rng.normal(0, 0.15, 3)), (rng.normal(0, 0.02), rng.normal(0, 0.02), rng.normal(0, 0.1))))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::8] * np.float32(4.0))
    Tsb = syn.tsb_offset()
    # unsharded reference on the product
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, Tsb)
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    upd.update(d_p, d_a)
    ref_attrs = d_a.download()
    upd.close()

    sh = ra.ShardedParticleFilterHip(v, f, devices=devices, loopback=loopback)
    world = len(devices)
    assert sh.world == world
    sh.set_particles(poses, attrs)
    _trace(ra, 1)
    w = sh.update(beams, Tsb)
    tr = _phases(_trace(ra, 0))
    # the update is enqueued on every rank and NOT waited for on the host (the gather waits for it on the device, an event per rank)
    assert tr["update"] == ["E%d" % r for r in range(world)], tr
    assert _enqueues_precede_waits(tr["gather"], world), tr
    for rk in range(world):
        assert np.array_equal(sh.weights(rk), w)      # every rank holds the same dense vector
    p2, a2 = sh.download()
    assert a2.tobytes() == ref_attrs.tobytes() and p2.tobytes() == poses.tobytes()
    assert np.array_equal(w, ref_attrs["likelihood"]["mean"])
    st = sh.stats()
    Lm = ref_attrs["likelihood"]["mean"].astype(np.float64)
    assert abs(st["sum"] - Lm.sum()) <= 1e-6 * Lm.sum() and st["max"] == np.float32(Lm.max())
    # pose estimate over all particles and over the first 1000 (max_induction_particles)
    for n_ind in (n, 1000):
        est = sh.pose_estimate(n_ind)
        ref = orc.estimate_stats(poses, ref_attrs, n_ind)
        assert est["nparticles"] == ref["nparticles"] == n_ind
        for k in ("mean", "sigma", "min", "max"):
            assert abs(est["likelihood"][k] - ref["likelihood"][k]) <= 1e-9 + 1e-9 * abs(ref["likelihood"][k]), k
        assert np.array_equal(est["trans_bb_min"], ref["trans_bb_min"].astype(np.float32))
        assert np.array_equal(est["trans_bb_max"], ref["trans_bb_max"].astype(np.float32))
        qa = np.array([est["pose"]["R"][k] for k in "xyzw"], np.float64)
        qb = np.array([ref["pose"]["R"][k] for k in "xyzw"], np.float64)
        assert min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) < 1e-6
        assert np.allclose([est["pose"]["t"][k] for k in "xyz"], [ref["pose"]["t"][k] for k in "xyz"], rtol=1e-6, atol=1e-6)
        assert np.allclose(est["covariance"], ref["covariance"], rtol=1e-4, atol=1e-6 * np.abs(ref["covariance"]).max())
    # distributed gladiator tournament == the single-GPU tournament on the same cloud
    rs = ra.GladiatorResamplerHip(ctx)
    d_p2, d_a2 = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, ref_attrs)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n)
    rs.seed = 42
    rs.update(d_p2, d_a2, d_pn, d_an, n)
    _trace(ra, 1)
    sh.resample(seed=42, step=0)
    assert _enqueues_precede_waits(_phases(_trace(ra, 0))["resample"], world)
    p3, a3 = sh.download()
    assert p3.tobytes() == d_pn.download().tobytes() and a3.tobytes() == d_an.download().tobytes()
    rs.close()
    # ... and the distributed residual resampling == the single-GPU one (on the cloud the tournament left behind)
    rr = ra.ResidualResamplerHip(ctx, seed=43)
    d_p4, d_a4 = ra.DeviceArray.from_host(ctx, p3), ra.DeviceArray.from_host(ctx, a3)
    rr.update(d_p4, d_a4, d_pn, d_an, n)
    _trace(ra, 1)
    sh.resample(seed=43, step=0, residual=True)
    assert _enqueues_precede_waits(_phases(_trace(ra, 0))["resample"], world)
    p5, a5 = sh.download()
    assert p5.tobytes() == d_pn.download().tobytes() and a5.tobytes() == d_an.download().tobytes()
    assert a5.tobytes() != a3.tobytes()
    rr.close()
    sh.close()


@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_c_abi_sharded_pose_batch_equals_unsharded(ra, orc, ctx, meshes, devices):
    """rmclhip_rcc_sharded_correct_batch: one operator replica per entry of `devices` over ONE host BVH build, the poses of the batch
    block-partitioned, every replica's chain enqueued before any is waited for.  A one-GPU box runs the ndev > 1 branches with
    several replicas on device 0 (allowed: no collective is involved): bit-identical to rmclhip_rcc_correct_batch of one operator,
    ragged partitions and a batch smaller than the number of replicas included; against the oracle to 1e-5."""
    import oracle_micp as om
    from rmcl_amd import synthetic as syn, types as T
    from test_gpu_reduce import _transform_close
    v, f = meshes("sphere20k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_vlp16_900(0.0)
    Tsb = syn.tsb_offset()
    ident = T.identity()
    meas = m.simulate_spherical(model, Tsb, ident, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rng = np.random.RandomState(21)
    poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-0.3, 0.3, 3)), (0.0, 0.0, rng.uniform(-0.05, 0.05)))
                      for _ in range(23)], dtype=T.TRANSFORM)

    def configure(r):
        r.setTsb(Tsb)
        r.setModel(model)
        r.set_dataset(ds, mask)
        r.params.max_dist = r.adaptive_max_dist_min = 1.0

    one = ra.RCCHipSpherical(hm)
    configure(one)
    sh = ra.ShardedCorrectorHip(devices, v, f)
    assert sh.world == len(devices)
    sh.for_each(configure)
    for n in (23, 2, 1):
        Td1, st1 = one.correct_batch(poses[:n])
        for _ in range(2):    # twice: the staging buffers are reused
            Td, st = sh.correct_batch(poses[:n])
            assert Td.tobytes() == Td1.tobytes() and st.tobytes() == st1.tobytes()
    Tr, sr = om.correct_batch(m, model, Tsb, poses, ds, mask, 1.0, nthreads=8)
    Td, st = sh.correct_batch(poses)
    for i in range(len(poses)):
        assert int(st[i]["n_meas"]) == int(sr[i]["n_meas"])
        _transform_close(Td[i], Tr[i], 1e-5)
    sh.close()
    one.close()


def test_loopback_ragged_three_ranks(ra, orc, ctx, meshes):
    """1001 particles over three loopback ranks on device 0 (334 + 334 + 333, padded shards in the gather): update + all-gather ==
    unsharded update, {sum, max} and the pose estimate from the all-reduced moments == the one-rank results; gladiator and residual
    resampling of the ragged partition == the one-rank cloud, particle for particle."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    n = 1001
    poses, attrs = syn.uniform_particles(n, seed=9, bb_min=(-8, -8, 0.2, 0, 0, -math.pi), bb_max=(8, 8, 3, 0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::8] * np.float32(4.0))
    Tsb = syn.tsb_offset()
    res = {}
    clouds = {}
    for devices in ((0,), (0, 0, 0)):
        sh = ra.ShardedParticleFilterHip(v, f, devices=devices, loopback=True)
        sh.set_particles(poses, attrs)
        w = sh.update(beams, Tsb)
        res[len(devices)] = (w.copy(), sh.download()[1], sh.stats(), sh.pose_estimate(n))
        # both resamplers on the ragged partition (round 4: the padded gather is squeezed dense before the tournament / the fill),
        # twice each (the second call runs on the first one's output and on reused buffers), then another update on the new cloud
        out = []
        for residual in (False, True):
            for step in (0, 1):
                sh.resample(seed=7, step=step, residual=residual)
                out.append(tuple(x.copy() for x in sh.download()))
        w2 = sh.update(beams, Tsb)
        out.append((w2.copy(),))
        clouds[len(devices)] = out
        sh.close()
    for a, b in zip(clouds[1], clouds[3]):
        for x, y in zip(a, b):
            assert x.tobytes() == y.tobytes()
    (w1, a1, s1, e1), (w3, a3, s3, e3) = res[1], res[3]
    assert np.array_equal(w1, w3) and a1.tobytes() == a3.tobytes()
    assert abs(s1["sum"] - s3["sum"]) <= 1e-6 * abs(s1["sum"]) and s1["max"] == s3["max"]
    assert np.allclose([e1["pose"]["t"][k] for k in "xyz"], [e3["pose"]["t"][k] for k in "xyz"], rtol=1e-6, atol=1e-7)
    assert np.allclose(e1["covariance"], e3["covariance"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("world,resample", [(1, "gladiator"), (3, "residual"), (8, "gladiator")])
def test_sharded_cycle_motion_update_resample_equals_the_single_device_cycle(ra, orc, ctx, meshes, world, resample):
    """the whole cycle of the filter node (rmcl_localization.cpp:84, 432-552) behind the one-process ABI --
    rmclhip_pf_sharded_motion_update (k_pf_motion per rank, collision ray included) and rmclhip_pf_sharded_step (motion -> sensor update
    -> weight all-gather -> {sum, max} -> resampling) -- on loopback ranks of device 0: particle for particle, bit for bit, the cloud the
    single-device sequence TFMotionUpdaterHip / PCDSensorUpdaterHip / resampler leaves; three cycles, a ragged particle count, particles
    that cross a wall (collision -> likelihood {0, 0, MAX_N_MEAS}); and the recorded call sequence enqueues every rank's launch of a
    phase before the host waits for any."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    hm = ra.import_hip_map(ctx, v, f)
    n = 5003
    poses, attrs = syn.uniform_particles(n, seed=15, bb_min=(-9.5, -9.5, 0.2, 0, 0, -math.pi), bb_max=(9.5, 9.5, 3, 0, 0, math.pi))
    attrs["likelihood"]["n_meas"] = np.random.RandomState(1).randint(0, 300, n)
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::4] * np.float32(4.0))
    Tsb = syn.tsb_offset()
    steps = [T.transform_from_rpy((0.9, 0.1, 0.0), (0.0, 0.0, 0.05)), T.transform_from_rpy((0.4, -0.3, 0.02), (0.0, 0.0, -0.2)),
             T.transform_from_rpy((1.5, 0.0, 0.0), (0.0, 0.0, 0.0))]
    # ---- the single-device cycle
    mot = ra.TFMotionUpdaterHip(hm, check_collision=True)
    mot.init()
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, Tsb)
    rs = ra.ResidualResamplerHip(ctx, seed=77) if resample == "residual" else ra.GladiatorResamplerHip(ctx, seed=77)
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n)
    ref = []
    for k, Tm in enumerate(steps):
        mot.update(d_p, d_a, n, Tm, 0.1)
        after_motion = d_a.download()
        upd.update(d_p, d_a)
        st = rs.compute_stats(d_a, n)
        rs.step = k
        rs.update(d_p, d_a, d_pn, d_an, n)
        d_p, d_pn, d_a, d_an = d_pn, d_p, d_an, d_a
        ref.append((d_p.download(), d_a.download(), st, after_motion))
    assert (ref[0][3]["likelihood"]["n_meas"] == 10000).sum() > 20      # some particles did cross a wall
    # ---- the sharded cycle, phase by phase (cycle 0) and through the one-call step (cycles 1, 2)
    sh = ra.ShardedParticleFilterHip(v, f, devices=(0,) * world, loopback=True)
    sh.set_particles(poses, attrs)
    _trace(ra, 1)
    sh.motion_update(steps[0], 0.1, check_collision=True)
    tr = _phases(_trace(ra, 0))
    assert _enqueues_precede_waits(tr["motion"], world), tr
    assert sh.download()[1].tobytes() == ref[0][3].tobytes()
    sh.update(beams, Tsb)
    st0 = sh.stats()
    sh.resample(seed=77, step=0, residual=(resample == "residual"))
    p, a = sh.download()
    assert p.tobytes() == ref[0][0].tobytes() and a.tobytes() == ref[0][1].tobytes()
    # round 6: {sum, max} are reduced on every rank from the gathered weights in the single-device kernel's order: the same BITS
    assert st0["max"] == ref[0][2]["max"] and st0["sum"] == ref[0][2]["sum"]
    for k in (1, 2):
        _trace(ra, 1)
        st = sh.step(beams, Tsb, T_bnew_bold=steps[k], forget_rate=0.1, check_collision=True, resample=resample, seed=77, step=k)
        tr = _phases(_trace(ra, 0))
        # motion and update are enqueued on every rank without a host wait in between; the first wait belongs to the gather
        assert tr["motion"] == ["E%d" % r for r in range(world)] and tr["update"] == ["E%d" % r for r in range(world)], tr
        assert _enqueues_precede_waits(tr["gather"], world) and _enqueues_precede_waits(tr["resample"], world), tr
        p, a = sh.download()
        assert p.tobytes() == ref[k][0].tobytes() and a.tobytes() == ref[k][1].tobytes(), "cycle %d" % k
        assert st["max"] == ref[k][2]["max"] and st["sum"] == ref[k][2]["sum"]
        assert "stats" not in tr or tr["stats"] == ["E%d" % r for r in range(world)] + ["W0"], tr   # no collective, ONE host wait
    sh.close()
    for o in (mot, upd, rs):
        o.close()


@pytest.mark.parametrize("resample", ["gladiator", "residual"])
def test_sharded_cycle_does_not_depend_on_the_collectives_order_of_summation(ra, ctx, meshes, resample):
    """on real RCCL the order in which an all-reduce adds the ranks' partials is the library's.  The loopback
    communicator is told to start its sums at rank 1, 2, ... (rmclhip_comm_loopback_set_reduce_rotation): the cycle's {sum, max} -- which feed
    size_t(L / sum * N) in the residual resampler -- and the resampled cloud must stay the single-device bits, because since round 6 they
    come from every rank's own copy of the gathered weights, not from an all-reduce.  (The pose estimate still uses all-reduces of moment
    sums: it may move in its last bits and is compared to 1e-12.)"""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    hm = ra.import_hip_map(ctx, v, f)
    n, world = 4001, 5
    poses, attrs = syn.uniform_particles(n, seed=21, bb_min=(-9, -9, 0.3, 0, 0, -math.pi), bb_max=(9, 9, 3, 0, 0, math.pi))
    attrs["likelihood"]["n_meas"] = np.random.RandomState(3).randint(0, 300, n)
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::4] * np.float32(4.0))
    Tsb = syn.tsb_offset()
    Tm = T.transform_from_rpy((0.4, -0.1, 0.0), (0.0, 0.0, 0.1))
    # the single-device cycle
    mot = ra.TFMotionUpdaterHip(hm, check_collision=True)
    mot.init()
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, Tsb)
    rs = ra.ResidualResamplerHip(ctx, seed=5) if resample == "residual" else ra.GladiatorResamplerHip(ctx, seed=5)
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n)
    mot.update(d_p, d_a, n, Tm, 0.1)
    upd.update(d_p, d_a)
    st_ref = rs.compute_stats(d_a, n)
    rs.step = 0
    rs.update(d_p, d_a, d_pn, d_an, n)
    p_ref, a_ref = d_pn.download(), d_an.download()
    est = []
    for rot in range(world):
        sh = ra.ShardedParticleFilterHip(v, f, devices=(0,) * world, loopback=True)
        assert sh.collective_ranks() == (world, "loopback")
        sh.set_loopback_reduce_rotation(rot)
        sh.set_particles(poses, attrs)
        st = sh.step(beams, Tsb, T_bnew_bold=Tm, forget_rate=0.1, check_collision=True, resample=resample, seed=5, step=0)
        assert st["sum"] == st_ref["sum"] and st["max"] == st_ref["max"], "rotation %d" % rot
        p, a = sh.download()
        assert p.tobytes() == p_ref.tobytes() and a.tobytes() == a_ref.tobytes(), "rotation %d" % rot
        # a statistics call on a cloud nobody gathered since it changed (the resampling replaced it) gathers first: the single-device
        # statistics of the resampled cloud, bit for bit
        st2, st2_ref = sh.stats(), rs.compute_stats(d_an, n)
        assert st2["sum"] == st2_ref["sum"] and st2["max"] == st2_ref["max"]
        est.append(sh.pose_estimate())
        sh.close()
    for e in est[1:]:
        assert np.allclose(e["covariance"], est[0]["covariance"], rtol=1e-9, atol=1e-12)
        assert np.allclose([e["pose"]["t"][k] for k in "xyz"], [est[0]["pose"]["t"][k] for k in "xyz"], rtol=0, atol=1e-6)
    for o in (mot, upd, rs):
        o.close()


def test_c5_full_size_one_run_on_eight_loopback_ranks(ra, orc, ctx, meshes):
    """BASELINE config C5 at FULL size as one run: 1 000 000 particles x 256 beams on the 1 M-triangle sphere through
    rmclhip_pf_update_sharded + weight all-gather + {sum, max} + pose estimate + the distributed tournament, on eight LOOPBACK ranks of
    this one GPU (RCCL itself needs eight GPUs; the loopback communicator runs the same ndev = 8 code on one).  Checked against the eight
    single-shard updates (each rank's block == PCDSensorUpdaterHip on that block alone), the gathered weights, an oracle sample from
    every shard, and ONE unsharded tournament over the whole cloud."""
    from rmcl_amd import distributed as D, synthetic as syn, types as T
    v, f = meshes("sphere1m")
    world, n = 8, 1000000
    poses, attrs = syn.uniform_particles(n, seed=5, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(10.0))
    assert len(beams) == 256
    sh = ra.ShardedParticleFilterHip(v, f, devices=(0,) * world, loopback=True)
    sh.set_particles(poses, attrs)
    st = sh.step(beams, T.identity())                       # update + gather + {sum, max}, no motion, no resampling
    w = sh.weights(3)
    p1, a1 = sh.download()
    assert p1.tobytes() == poses.tobytes()
    assert np.array_equal(w, a1["likelihood"]["mean"]) and np.all(a1["likelihood"]["n_meas"] == 256)
    # the eight single-shard updates on one plain updater
    hm = ra.import_hip_map(ctx, v, f)
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, T.identity())
    m = orc.Mesh(v, f)
    for r in range(world):
        lo, hi = D.shard_bounds(n, r, world)
        d_p, d_a = ra.DeviceArray.from_host(ctx, poses[lo:hi]), ra.DeviceArray.from_host(ctx, attrs[lo:hi])
        upd.update(d_p, d_a)
        assert d_a.download().tobytes() == a1[lo:hi].tobytes(), "shard %d" % r
        sub = slice(lo + 17 * r, lo + 17 * r + 2000)        # 2 000 particles of every shard against the oracle
        ref = attrs[sub].copy()
        m.pf_update(poses[sub], ref, beams, T.identity(), orc.pf_params(), bvh=2, nthreads=16)
        assert_close_rel(a1["likelihood"]["mean"][sub], ref["likelihood"]["mean"], 1e-5, 1e-12, "C5 shard %d mean" % r)
        d_p.free(); d_a.free()
    Lm = a1["likelihood"]["mean"].astype(np.float64)
    assert abs(st["sum"] - Lm.sum()) <= 1e-6 * Lm.sum() and st["max"] == np.float32(Lm.max())
    est = sh.pose_estimate(10000)
    ref_e = orc.estimate_stats(poses, a1, 10000)
    assert np.allclose([est["pose"]["t"][k] for k in "xyz"], [ref_e["pose"]["t"][k] for k in "xyz"], rtol=1e-6, atol=1e-6)
    # ONE unsharded tournament over the million particles == the distributed one
    rs = ra.GladiatorResamplerHip(ctx, seed=9)
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, a1)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n)
    rs.update(d_p, d_a, d_pn, d_an, n)
    sh.resample(seed=9, step=0)
    p2, a2 = sh.download()
    assert p2.tobytes() == d_pn.download().tobytes() and a2.tobytes() == d_an.download().tobytes()
    rs.close(); upd.close(); sh.close()


def test_allgather_alone_map_refcount_and_kernel_clock(ra, orc, ctx, meshes):
    """three entry points nothing else calls directly: rmclhip_pf_allgather_weights on its own (after a motion update changed nothing the
    weights depend on, a second gather returns the same dense vector on every rank), rmclhip_map_retain / _release (a map that a
    registry retained outlives its creator's release: an operator built afterwards still traces), rmclhip_rcc_last_kernel_ms (HIP-event
    time of the last synchronous find with kernel timing on: positive, below the call's host time)."""
    import ctypes as C
    import time
    from rmcl_amd import _capi, synthetic as syn, types as T
    v, f = meshes("room30k")
    n = 777
    poses, attrs = syn.uniform_particles(n, seed=4, bb_min=(-8, -8, 0.2, 0, 0, -math.pi), bb_max=(8, 8, 3, 0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::8] * np.float32(4.0))
    sh = ra.ShardedParticleFilterHip(v, f, devices=(0, 0, 0), loopback=True)
    sh.set_particles(poses, attrs)
    w = sh.update(beams, syn.tsb_offset()).copy()
    _capi.check(_capi.lib().rmclhip_pf_allgather_weights(sh._h))
    for rank in range(3):
        again = np.zeros(n, np.float32)
        _capi.check(_capi.lib().rmclhip_pf_sharded_get_weights(sh._h, rank, again.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(again, w), rank
    sh.close()

    hm = ra.import_hip_map(ctx, v, f)
    handle = C.c_void_p(hm.handle.value)
    _capi.check(_capi.lib().rmclhip_map_retain(handle))     # the registry's reference ("<name>.hip" in an rm::MapMap)
    hm.release()                                            # the creator lets go
    borrowed = ra.HipMap.__new__(ra.HipMap)
    borrowed.ctx, borrowed._h = ctx, handle
    model = syn.model_c1()
    rcc = ra.RCCHipSpherical(borrowed)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    Tbm = T.transform_from_rpy((1.0, -1.5, 1.2), (0.01, -0.02, 0.5))
    rcc.find(Tbm)
    ref = orc.Mesh(v, f).simulate_spherical(model, T.identity(), Tbm, bvh=True, nthreads=8)
    assert np.array_equal(rcc.modelView()["face_ids"], ref["face_ids"])
    rcc.set_kernel_timing(True)
    t0 = time.perf_counter()
    rcc.find(Tbm)
    host_ms = (time.perf_counter() - t0) * 1e3
    find_ms, _ = rcc.last_kernel_ms()
    assert 0.0 < find_ms < host_ms, (find_ms, host_ms)
    rcc.close()
    borrowed.release()                                      # the last reference
