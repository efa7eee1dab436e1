"""N > 1 path on CPU: world_size-2 (and 3, ragged) gloo processes shard the particle cloud with
rmcl_amd.distributed, run the rank-local sensor update (a CPU stand-in backed by the oracle -- the HIP
updater needs a GPU), all-gather the weights and must reproduce the unsharded result exactly.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle as orc
    from rmcl_amd import distributed as D, synthetic as syn
    from rmcl_amd.pf import beams_from_points

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v, f = syn.cube_room()
        m = orc.Mesh(v, f)
        poses, attrs = syn.uniform_particles(n_total, seed=13, bb_min=(-4, -4, -2, 0, 0, -3.1), bb_max=(4, 4, 2, 0, 0, 3.1))
        beams = beams_from_points(syn.model_directions(syn.model_pf16())[::16] * np.float32(3.5))
        Tsb = syn.tsb_offset()
        lo, hi = D.shard_bounds(n_total, rank, world)
        local_attrs = attrs[lo:hi].copy()

        class OracleUpdater:
            """TEST STUB with the PCDSensorUpdaterHip call surface ShardedSensorUpdate drives (update /
            extract_weights); the rank-local beam evaluation is the CPU oracle's."""

            def update(self, poses_dev, attrs_dev, n_particles=None):
                if n_particles:
                    m.pf_update(poses_dev[:n_particles], attrs_dev[:n_particles], beams, Tsb, orc.pf_params(), bvh=True)

            def extract_weights(self, attrs_dev, n, weights_ptr):
                import ctypes
                w = np.ascontiguousarray(attrs_dev["likelihood"]["mean"][:n])
                ctypes.memmove(weights_ptr, w.ctypes.data, 4 * n)

        sharded = D.ShardedSensorUpdate(OracleUpdater(), n_total, rank, world)
        assert (sharded.lo, sharded.hi, sharded.n_local) == (lo, hi, hi - lo)
        w_local = torch.zeros(hi - lo, dtype=torch.float32)
        for _ in range(2):   # twice: the preallocated gather buffers are reused
            local_attrs = attrs[lo:hi].copy()
            gathered = sharded.update(poses[lo:hi], local_attrs, w_local)
        ssum, smax = D.allreduce_sum_max(w_local)
        # round 6: the statistics every rank computes ITSELF from the gathered vector (no second collective): identical on every rank by
        # construction, equal to the unsharded cloud's
        gsum, gmax = D.gathered_sum_max(gathered[:n_total])
        # unsharded reference on every rank
        ref = attrs.copy()
        m.pf_update(poses, ref, beams, Tsb, orc.pf_params(), bvh=True)
        ok = np.array_equal(gathered.numpy(), ref["likelihood"]["mean"])
        ok &= abs(ssum - float(ref["likelihood"]["mean"].astype(np.float64).sum())) < 1e-6
        ok &= smax == float(ref["likelihood"]["mean"].max())
        ok &= gsum == float(np.float32(ref["likelihood"]["mean"].astype(np.float64).sum())) and gmax == float(max(np.float32(0), ref["likelihood"]["mean"].max()))
        ok &= gathered.numel() == n_total
        # distributed gladiator tournament == the unsharded one (Philox counter = global champion index)
        cfg = orc.gladiator_config(min_noise_roll=0.01)

        def resample(poses_all, attrs_all, n, first, count):
            pa = poses_all.numpy().reshape(-1).view(orc.TRANSFORM)
            aa = attrs_all.numpy().reshape(-1).view(orc.PARTICLE_ATTRIBUTES)
            pn, an = orc.gladiator_resample(pa, aa, cfg, seed=77, step=3, first=first, count=count)
            return pn, an

        full = ref.copy()
        lp = torch.from_numpy(poses[lo:hi].copy().view(np.uint8).reshape(hi - lo, 32))
        la = torch.from_numpy(full[lo:hi].copy().view(np.uint8).reshape(hi - lo, 36))
        pn, an = D.ShardedResample(resample, n_total, rank, world).update(lp, la)
        pn_ref, an_ref = orc.gladiator_resample(poses, full, cfg, seed=77, step=3)
        ok &= pn.tobytes() == pn_ref[lo:hi].tobytes() and an.tobytes() == an_ref[lo:hi].tobytes()
        # the same exchange for the residual resampler: every rank fills its slots [lo, hi) of the new cloud from the gathered one
        if float(full["likelihood"]["mean"].astype(np.float64).sum()) > 0 and n_total >= 100:
            def resample_residual(poses_all, attrs_all, n, first, count):
                pa = poses_all.numpy().reshape(-1).view(orc.TRANSFORM)
                aa = attrs_all.numpy().reshape(-1).view(orc.PARTICLE_ATTRIBUTES)
                pr, ar, filled, _ = orc.residual_resample(pa, aa, cfg, seed=78, step=1, max_draws=200 * n)
                assert filled == n
                return pr[first:first + count], ar[first:first + count]

            pr, ar = D.ShardedResample(resample_residual, n_total, rank, world).update(lp, la)
            pr_ref, ar_ref, filled, _ = orc.residual_resample(poses, full, cfg, seed=78, step=1, max_draws=200 * n_total)
            ok &= filled == n_total and pr.tobytes() == pr_ref[lo:hi].tobytes() and ar.tobytes() == ar_ref[lo:hi].tobytes()
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as fh:
            fh.write("OK" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1000), (3, 101), (2, 1)])
def test_sharded_pf_update_allgather(tmp_path, world, n_total):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("rank%d.txt" % r)).read_text() == "OK"


def test_shard_bounds_partition():
    from rmcl_amd.distributed import shard_bounds, shard_capacity
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) <= shard_capacity(n, world)


def test_c_abi_shard_bounds_and_comm_without_device(ra):
    """the C ABI's block partition (rmclhip_shard_bounds, used by rmclhip_pf_sharded_*) is the Python one; without a HIP device
    rmclhip_comm_create fails loudly (no CPU fallback, and librccl is not even loaded)."""
    import ctypes as C
    from rmcl_amd.distributed import shard_bounds
    L = ra._capi.lib()
    for n in (0, 1, 7, 8, 9, 100000, 1000003):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                lo, hi = C.c_uint32(), C.c_uint32()
                L.rmclhip_shard_bounds(n, rank, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_bounds(n, rank, world)
    import torch
    if not torch.cuda.is_available():
        comm = C.c_void_p()
        assert L.rmclhip_comm_create(None, 1, C.byref(comm)) == ra._capi.ERR_NO_DEVICE and not comm
        assert b"no HIP device" in L.rmclhip_last_error()
    assert L.rmclhip_comm_create(None, 0, C.byref(C.c_void_p())) == ra._capi.ERR_INVALID
    assert L.rmclhip_comm_size(None) == 0


# ---- pose batches sharded over ranks (north_star: pose-corrections/s at 1/2/4/8 GPUs; SURVEY.md 8(e): no exchange) ----------------
def _pose_worker(rank, world, port, nposes, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle as orc
    import oracle_micp as om
    from rmcl_amd import distributed as D, synthetic as syn, types as T

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v, f = syn.cube_room()
        m = orc.Mesh(v, f)
        model = syn.model_c1()
        Tsb = syn.tsb_offset()
        truth = T.transform_from_rpy((0.5, -0.3, 0.2), (0.0, 0.0, 0.3))
        meas = m.simulate_spherical(model, Tsb, truth, bvh=True)
        ds, mask = om.dataset_from_ranges(model, meas["ranges"])
        rng = np.random.RandomState(5)
        poses = np.array([T.mult(truth, T.transform_from_rpy(tuple(rng.uniform(-0.2, 0.2, 3)), (0.0, 0.0, rng.uniform(-0.05, 0.05))))
                          for _ in range(nposes)], dtype=T.TRANSFORM)

        def correct_fn(block):   # TEST STUB with RCCHip*.correct_batch's call surface, the CPU oracle inside
            return om.correct_batch(m, model, Tsb, block, ds, mask, 1.0)

        sh = D.ShardedBatchCorrector(correct_fn, rank, world)
        lo, hi, Td_local, st_local = sh.correct(poses)
        _, _, Td_all, _ = sh.correct(poses, gather=True)
        Tr, sr = om.correct_batch(m, model, Tsb, poses, ds, mask, 1.0)      # unpartitioned, on every rank
        ok = (lo, hi) == D.shard_bounds(nposes, rank, world)
        ok &= np.asarray(Td_local).tobytes() == np.asarray(Tr[lo:hi]).tobytes()
        ok &= np.asarray(Td_all).tobytes() == np.asarray(Tr).tobytes()
        if hi > lo:
            ok &= all(int(a["n_meas"]) == int(b["n_meas"]) for a, b in zip(st_local, sr[lo:hi]))
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as fh:
            fh.write("OK" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nposes", [(2, 10), (3, 7), (3, 2)])
def test_sharded_pose_batch_equals_unpartitioned(tmp_path, world, nposes):
    """the partitioned batch corrector (each rank its block of the pose list, no data-path collective; an optional all-gather of the
    32-B deltas) returns exactly what the unpartitioned call returns -- ragged blocks and ranks without a pose included."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pose_worker, args=(world, port, nposes, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("rank%d.txt" % r)).read_text() == "OK"
