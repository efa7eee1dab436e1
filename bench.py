#!/usr/bin/env python
"""bench.py -- headline benchmark of the RMCL / MICP-L hot path on MI355X.

Metric (BASELINE.json): ray-mesh intersections/s (+ derived pose-corrections/s), 100k-triangle mesh,
128x1024 spherical scan.

Default workload (every N): config C2 = one pose x 131 072 rays against the sphere-100k mesh per step, i.e.
one RCC find() launch (rmcl/src/rmcl/registration/RCCEmbree.cpp:26-36) with all five output attributes
written.  A single MICP pose does not shard: --gpus N runs N independent replicas, one
process per GPU, a different pose per rank, no data-path collective ("replicas only", weak scaling).

--workload pf: config C4 per GPU (100k particles x 256 beams, sphere-100k), particles block-partitioned over
the ranks (weak scaling: 100k per GPU) and ONE RCCL all-gather of the 4 B x N weights per step.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N ...          (launched plainly: spawns its own N ranks, one per GPU, and waits for them)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Every run (any N) also reports `extras.pf_sharded`: config C5's per-GPU term -- 125 000 particles x 256 beams per GPU on the
1M-triangle sphere, particles block-partitioned over the ranks, sensor update + ONE all-gather of the weights (RCCL when
N > 1) + the {sum, max} all-reduce; at N = 8 that block IS config C5.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the stream the kernel runs on
(rmclhip_rcc_time_find / rmclhip_pf_time_update); `cpu_baseline` times the CPU oracle (a port -- the
reference's Embree path cannot be built here) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import math
import os
import sys
import time

# several operators = several streams: ask the HIP runtime for eight hardware queues BEFORE anything initialises it (torch does, below);
# with the default four, a fifth stream shares a queue with another and the two no longer overlap (rmclhip_ctx_create)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_raycast(n_rays, n_tri, n_poses):
    """SURVEY.md 8(d) B_rc: outputs 33 B/ray (hits 1 + ranges 4 + points 12 + normals 12 + face ids 4)
    + 36 B/triangle + 32 B per BVH2 node of the reference tree + 32 B/pose."""
    return n_rays * 33 + n_tri * 36 + (2 * n_tri - 1) * 32 + n_poses * 32


def algorithmic_bytes_pf(n_particles, n_beams):
    """SURVEY.md 8(d) B_pf: pose 32 B read + attrs 36 B read + 36 B written per particle, 64 B per beam."""
    return n_particles * (32 + 36 + 36) + n_beams * 64


def measured_traffic(kernel_key):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, produced by
    tools/pmc_traffic.sh + tools/traffic_from_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes, KiB -> bytes;
    gfx950's 2x FETCH_SIZE under-count applies to wide 16 B/lane streams only and is noted there).  Keyed by the
    traversal kind that was profiled ("k_find_kind17", ...): a kind without a profile reports null, never another kind's."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        d = json.load(fh)
    return d.get(kernel_key, {}).get("hbm_bytes_per_launch")


def product_tree_bytes(key):
    """cache-side bytes the PRODUCT's tree moves for the timed scan (tools/product_tree_bytes.py -> profiles/traversal_product_tree.json)"""
    path = os.path.join(ROOT, "profiles", "traversal_product_tree.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh).get(key)


def median_kernel_ms(fn, batches=9):
    """median over `batches` of the mean launch-to-launch time of a batch of back-to-back launches (one noisy
    launch moves a mean of 20 by 5 %; it does not move the median of batch means)"""
    ts = sorted(fn() for _ in range(batches))
    return ts[len(ts) // 2]


def main():
    # Libraries below us write to stdout on their own (RCCL prints a version banner when its first communicator is created):
    # everything but the ONE result line goes to stderr -- fd 1 is pointed at fd 2 for the whole run and the JSON line is
    # written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=("c2", "pf"), default="c2")
    ap.add_argument("--variant", type=int, default=15,
                    help="find traversal: 15 automatic (default), 1 one lane per ray, 2 four lanes per ray, 0 wave packet")
    ap.add_argument("--no-autotune", action="store_true",
                    help="time the headline scan on the automatic RULE's traversal instead of the one rmclhip_rcc_autotune measures fastest "
                         "on this map / model / pose (INTEGRATION.md: the sensor set-up calls it once; results are bit-identical either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    ap.add_argument("--all-on-device0", action="store_true", help="testing only: every rank uses GPU 0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rmcl_amd as ra
    from rmcl_amd import synthetic as syn, types as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        os.dup2(real_stdout, 1)
        sys.exit(_self_launch(args.gpus, sys.argv[1:]))
    dist = None
    if args.all_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    ctx = ra.Context(local_rank)
    v, f = syn.uv_sphere(100000)
    hm = ra.import_hip_map(ctx, v, f)
    cores = os.cpu_count() or 1
    extras = {}
    cpu = None
    pose_batches = None

    if args.workload == "c2":
        model = syn.model_c2()
        n_rays = model.phi.size * model.theta.size
        rcc = ra.RCCHipSpherical(hm)
        rcc.set_variant(args.variant)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        # replicas: rank r localises a scan from a different pose inside the same map
        Tbm = T.mult(syn.pose_c2_truth(), T.transform_from_rpy((0.05 * rank, -0.03 * rank, 0.0), (0.0, 0.0, 0.11 * rank)))
        # dominant kernel, measured live FIRST: HIP events on the rcc's own stream around back-to-back launches; median of 9
        # batches of 40 launches.  (It runs before the W + K steps on purpose: a run with few steps -- the driver's K = 20 is
        # 0.35 ms of GPU time -- would otherwise be timed on a device that has not left its idle clocks: 19.0 vs 18.2 us per
        # step measured; the metric is steady-state rays/s.  DESIGN.md section 5.)
        rule_kind = rcc.find_variant(1)
        rule_kernel_ms = median_kernel_ms(lambda: rcc.time_find(Tbm, iters=40))
        tuned = None
        if args.variant == 15 and not args.no_autotune:
            # what INTEGRATION.md's sensor set-up does once per (map, model): the operator measures its own single-scan traversals here
            # (kinds 2 / 23 / 24 with and without the frontier start, kind 32 = the cooperative descent with a short and a long list, then
            # the tile shape) and keeps the fastest.  The step below is the same rmclhip_rcc_find_async either way.
            k_t, ms_t = rcc.autotune(Tbm)
            tuned = {"rule_kind": rule_kind, "rule_kernel_ms": round(rule_kernel_ms, 5), "chosen_kind": k_t, "autotune_kernel_ms": round(ms_t, 5)}
        kernel_ms = median_kernel_ms(lambda: rcc.time_find(Tbm, iters=40)) if tuned else rule_kernel_ms
        step = rcc.find_async_fn(Tbm)   # rmclhip_rcc_find_async(handle, pose) with the pose converted once: one C call per step
        for _ in range(max(args.warmup, 1)):
            step()
        rcc.sync()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        rcc.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()   # this rank's K steps are complete; the closing barrier itself is not part of the K steps
        barrier()
        torch.cuda.synchronize()
        elapsed = max_over_ranks(t1 - t0)
        units_per_step = n_rays

        b_alg = algorithmic_bytes_raycast(n_rays, len(f), 1)
        kind = rcc.find_variant(1)   # the traversal that was timed: the autotuned one, the rule's (--no-autotune) or --variant's
        kname = "k_find<spherical, kind %d>" % kind
        traffic = measured_traffic("k_find_kind%d" % kind)
        if tuned is not None:
            extras["headline_traversal"] = dict(tuned, timed_kind=kind, rule_rays_per_s=round(n_rays / (rule_kernel_ms * 1e-3), 1),
                                                note="rmclhip_rcc_autotune ran once before the timed steps (a product call, INTEGRATION.md); "
                                                     "--no-autotune times the automatic rule's kind; results are bit-identical")

        if rank == 0 and not args.no_extras:
            # the boundary's Simulator surface (round 6): the MICP bundle {points, normals, hits} = 25 B/ray that Correspondences_::model_buffers_
            # carries (Correspondences.hpp:81-85) beside the five-output headline (33 B/ray); simulate() into a caller-owned
            # Bundle<Ranges, Normals> (scan_map_segmentation_embree.cpp:80-87); the free statistics_p2l on caller-owned views
            from rmcl_amd import _capi as _cb
            rcc.set_outputs(_cb.OUT_MICP)
            mb_ms = median_kernel_ms(lambda: rcc.time_find(Tbm, iters=40), 5)
            extras["find_micp_bundle_ms"] = round(mb_ms, 5)
            extras["find_micp_bundle_rays_per_s"] = round(n_rays / (mb_ms * 1e-3), 1)
            extras["find_micp_bundle_algorithmic_GBps"] = round((algorithmic_bytes_raycast(n_rays, len(f), 1) - 8 * n_rays) / (mb_ms * 1e-3) / 1e9, 1)
            rcc.set_outputs(_cb.OUT_ALL)
            bundle = rcc.simulate(Tbm, attributes=("ranges", "normals"))
            extras["simulate_ranges_normals_sync_call_ms"] = round(_median_call_ms(lambda: rcc.simulate(Tbm, attributes=("ranges", "normals"), into=bundle), reps=50), 5)
            rcc.find(Tbm)
            mvd = rcc.simulate(Tbm, attributes=("points", "normals", "hits"))
            extras["statistics_p2l_free_function_sync_call_ms"] = round(_median_call_ms(
                lambda: ra.statistics_p2l(ctx, T.identity(), mvd["points"], mvd["hits"], mvd["points"], mvd["normals"], mvd["hits"], n_rays, 1.0), reps=50), 5)
            # C3: MICP-L inner loop: (R) 1 find + 10 x (reduce + solve), (B) 10 x (find + reduce + solve).
            # The measured scan is the product's own simulation at the ground-truth pose (no oracle involved).
            rcc.find(syn.pose_c2_truth())
            rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
            rcc.params.max_dist = 1.0
            rcc.adaptive_max_dist_min = 0.15
            est = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation())
            for name, refind in (("R", False), ("B", True)):
                # complete synchronous corrections timed at the C ABI (host clock inside the library: what a C / C++
                # caller sees); the same call through the Python harness costs ~15 us more
                if not refind:
                    rcc.correct_once(est, T.identity(), 10, 0.0, False)   # the moment form learns its bounds on the first call
                dt = rcc.time_correct_once(est, T.identity(), 10, 0.0, refind, iters=50) * 1e-3
                extras["c3_schedule_%s_ms" % name] = round(dt * 1e3, 4)
                extras["c3_schedule_%s_pose_corrections_per_s" % name] = round(1.0 / dt, 1)
                extras["c3_schedule_%s_icp_iterations_per_s" % name] = round(10.0 / dt, 1)
            # the reference's UNCHANGED caller loop (micp_localization.cpp:900-964: find once, then per iteration computeCrossStatistics
            # + the CrossStatistics algebra + umeyama_transform ON THE HOST) through the public C entry points, timed in C: round 4
            # serves the per-iteration calls from the moments the find published (no launch); with the moment form off every call is a
            # streaming reduction + a completion wait (round 3's behaviour)
            rcc.find(est)      # (a find not followed by computeCrossStatistics: the next ones are plain finds)
            extras["find_sync_call_cabi_ms"] = round(rcc.time_find_sync(est, iters=100), 5)
            rcc.set_kernel_timing(True)
            extras["find_sync_call_with_event_timing_cabi_ms"] = round(rcc.time_find_sync(est, iters=100), 5)
            rcc.set_kernel_timing(False)
            ms, Tcl, scl = rcc.time_caller_loop(est, T.identity(), 10, 0.0, iters=50)
            extras["c3_schedule_R_unchanged_caller_cabi_ms"] = round(ms, 4)
            extras["c3_schedule_R_unchanged_caller_pose_corrections_per_s"] = round(1e3 / ms, 1)
            extras["c3_schedule_R_unchanged_caller_served"] = rcc.ccs_info()
            rcc.set_micp_fast(0)
            extras["c3_schedule_R_unchanged_caller_streaming_reduce_cabi_ms"] = round(rcc.time_caller_loop(est, T.identity(), 10, 0.0, iters=30)[0], 4)
            rcc.set_micp_fast(4)
            rcc.correct_once(est, T.identity(), 10, 0.0, False)
            extras["c3_schedule_R_device_loop_ms"] = round(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=50), 4)
            # (R) with the moments in a pass of their own (round 3's first form, three launches: A/B of the find's moment epilogue)
            rcc.set_micp_fast(3)
            rcc.correct_once(est, T.identity(), 10, 0.0, False)
            extras["c3_schedule_R_separate_moments_pass_ms"] = round(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=50), 4)
            rcc.set_micp_fast(1)
            # (R) again with the moment form off: one streaming launch per iteration (the fallback of the default form)
            info = rcc.micp_fast_info()
            extras["c3_schedule_R_moment_form"] = {k: info[k] for k in ("attempts", "done", "cap_exits", "overflows", "last_uncertain")}
            rcc.set_micp_fast(0)
            extras["c3_schedule_R_per_iteration_form_ms"] = round(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=50), 4)
            extras["c3_zero_iterations_ms"] = round(rcc.time_correct_once(est, T.identity(), 0, 0.0, False, iters=50), 4)
            rcc.set_micp_fast(1)
            red_ms = rcc.time_reduce(T.identity(), iters=100)
            extras["reduce_ms"] = round(red_ms, 5)
            extras["reduce_GBps"] = round((n_rays * 38 + 64) / (red_ms * 1e-3) / 1e9, 1)
            # pose batches (v1 corrector shape): 64 poses x 128x1024 rays in one launch
            rng = np.random.RandomState(0)
            poses = np.array([T.mult(syn.pose_c2_truth(), T.transform_from_rpy(tuple(rng.uniform(-0.5, 0.5, 3)), (0, 0, rng.uniform(-3, 3))))
                              for _ in range(64)], dtype=T.TRANSFORM)
            bms = rcc.time_find_batch(poses, iters=5)
            extras["find_batch64_ms"] = round(bms, 4)
            extras["find_batch64_rays_per_s"] = round(64 * n_rays / (bms * 1e-3), 1)
            rcc.correct_batch(poses)
            t1 = time.perf_counter()
            for _ in range(3):
                rcc.correct_batch(poses)
            dt = (time.perf_counter() - t1) / 3
            k_b, ms_b = rcc.autotune_batch(poses)      # kinds 23 / 24 with and without the frontier start, measured on these poses
            extras["find_batch64_autotuned"] = {"rule_kind": 24, "chosen_kind": k_b, "ms": round(ms_b, 4)}
            rcc.setModel(model)                        # forget the measurement: everything below runs on the rule
            extras["correct_batch64_ms"] = round(dt * 1e3, 4)
            extras["correct_batch64_pose_corrections_per_s"] = round(64 / dt, 1)
            # C4: particle filter, 100k particles x 256 beams
            pms, prays = _pf_c4(ra, syn, T, np, ctx, hm, 100000, 256, iters=3)
            extras["c4_pf_update_ms"] = round(pms, 4)
            extras["c4_pf_update_note"] = ("median of 5 timings of 3 launches; particles uniform in [-5,5]^2 x [-1,1] inside the "
                                           "radius-10 sphere (SURVEY C4 says [-8,8]^2, whose corners lie outside the sphere: c4_pf_update_survey_box_ms)")
            extras["c4_particle_beam_evals_per_s"] = round(prays / (pms * 1e-3), 1)
            # SURVEY section 8(d)'s box as written: its corners lie outside the radius-10 sphere, ~14 % of the particles see the map from outside
            sms, _ = _pf_c4(ra, syn, T, np, ctx, hm, 100000, 256, iters=3, bb=((-8, -8, -1), (8, 8, 1)))
            extras["c4_pf_update_survey_box_ms"] = round(sms, 4)
            extras["c4_particle_updates_per_s"] = round(100000 / (pms * 1e-3), 1)
            # the REFERENCE's GPU schedule as a comparator (PCDSensorUpdaterOptix.cpp:319-338): one launch per beam, attributes re-read and
            # re-written per beam (B_pf unfused = 256 x 100 000 x 104 B = 2.66 GB), the stream synchronised after every beam (:337)
            ums, ums_async = _pf_c4_unfused(ra, syn, T, np, ctx, hm, 100000, 256)
            extras["c4_unfused_reference_schedule_ms"] = round(ums, 3)
            extras["c4_unfused_reference_schedule_no_sync_between_beams_ms"] = round(ums_async, 3)
            b_unf = 256 * 100000 * 104
            extras["c4_unfused_reference_schedule_roofline"] = {
                "algorithmic_bytes": b_unf, "achieved_GBps": round(b_unf / (ums_async * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": round(b_unf / (ums_async * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "fused_algorithmic_bytes": algorithmic_bytes_pf(100000, 256), "fused_ms": round(pms, 4),
                "fused_over_unfused_speedup": round(ums / pms, 1)}
            extras["c4_pf_algorithmic_GBps"] = round(algorithmic_bytes_pf(100000, 256) / (pms * 1e-3) / 1e9, 2)
            # the filter's steady state: a CONVERGED cloud (100k particles ~ N(pose, 0.25 m, 5 deg yaw)), round 3's dealing and the
            # particle-coherent one (a wave's lanes = the same beam of 16 Morton-neighbouring particles x 4 beams); measured neutral:
            # the kernel is VALU-issue bound, coherence removes cache lines, not instructions (profiles/r04_pf_converged_mapping.txt)
            cms, _ = _pf_c4(ra, syn, T, np, ctx, hm, 100000, 256, iters=3, converged_at=T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4)))
            cms1, _ = _pf_c4(ra, syn, T, np, ctx, hm, 100000, 256, iters=3, converged_at=T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4)),
                             particle_minor=True)
            extras["c4_converged_pf_update_ms"] = round(cms, 4)
            extras["c4_converged_particle_beam_evals_per_s"] = round(prays / (cms * 1e-3), 1)
            extras["c4_converged_particle_minor_morton_ms"] = round(cms1, 4)
            extras.update(_pf_cycle(ra, syn, T, np, ctx, hm, 100000))
            # the same scan on a REALISTIC map (room-100k: occluders, vertex noise, open ceiling) and with the O1Dn model (the
            # documented deployment model: directions are data, +12 B/ray read)
            vr, fr = syn.noisy_room(100000)
            hmr = ra.import_hip_map(ctx, vr, fr)
            Troom = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
            rr = ra.RCCHipSpherical(hmr)
            rr.setTsb(T.identity())
            rr.setModel(model)
            rms = median_kernel_ms(lambda: rr.time_find(Troom, iters=40), 5)
            extras["find_room100k_ms"] = round(rms, 5)
            extras["find_room100k_rays_per_s"] = round(n_rays / (rms * 1e-3), 1)
            k_t, ms_t = rr.autotune(Troom)
            extras["find_room100k_autotuned"] = {"rule_kind": 23, "chosen_kind": k_t, "ms": round(ms_t, 5)}
            rr.setModel(model)
            # C3 on the room, over the size of the initial error (the sphere has no correspondence near the gate; a room has): the
            # undecided correspondences go to the host up to 1024 of them, beyond that the device loop takes the call
            # (profiles/r04_c3_regimes.txt)
            rr.find(Troom)
            rr.set_dataset_from_ranges(rr.modelView()["ranges"])
            rr.params.max_dist, rr.adaptive_max_dist_min = 1.0, 0.15
            for tag, sc in (("2cm_0.2deg", 0.1), ("10cm_1deg", 0.5), ("20cm_2deg", 1.0)):
                est_r = T.mult(Troom, T.transform_from_rpy((0.2 * sc, 0.0, 0.0), (0.0, 0.0, 0.0349 * sc)))
                for _ in range(3):
                    rr.correct_once(est_r, T.identity(), 10, 0.0, False)
                extras["c3_schedule_R_room100k_%s_ms" % tag] = round(rr.time_correct_once(est_r, T.identity(), 10, 0.0, False, iters=50), 4)
                extras["c3_schedule_R_room100k_%s_undecided" % tag] = rr.micp_fast_info()["last_uncertain"]
                # the reference's unchanged caller loop in the same regime (find + 10 x computeCrossStatistics through the C entry points)
                served0 = rr.ccs_info()
                rr.time_caller_loop(est_r, T.identity(), 10, 0.0, iters=3)
                extras["c3_unchanged_caller_room100k_%s_cabi_ms" % tag] = round(rr.time_caller_loop(est_r, T.identity(), 10, 0.0, iters=50)[0], 4)
                served1 = rr.ccs_info()
                extras["c3_unchanged_caller_room100k_%s_served" % tag] = {k: served1[k] - served0[k] for k in served1}
            rr.close()
            rpms, _ = _pf_c4(ra, syn, T, np, ctx, hmr, 100000, 256, iters=3, bb=((-9, -9, 0.3), (9, 9, 3)))
            extras["c4_room100k_pf_update_ms"] = round(rpms, 4)
            extras["c4_room100k_particle_beam_evals_per_s"] = round(prays / (rpms * 1e-3), 1)
            rcms, _ = _pf_c4(ra, syn, T, np, ctx, hmr, 100000, 256, iters=3, converged_at=T.transform_from_rpy((1.5, -2.0, 1.6), (0, 0, 0.4)))
            extras["c4_converged_room100k_pf_update_ms"] = round(rcms, 4)
            extras["c4_converged_room100k_particle_beam_evals_per_s"] = round(prays / (rcms * 1e-3), 1)
            dirs_c2 = syn.model_directions(model)
            for nm, hmx, Tx in (("sphere100k", hm, Tbm), ("room100k", hmr, Troom)):
                ro = ra.RCCHipO1Dn(hmx)
                ro.setTsb(T.identity())
                ro.setModel(model.theta.size, model.phi.size, float(model.range.min), float(model.range.max), (0.0, 0.0, 0.0), dirs_c2)
                oms = median_kernel_ms(lambda: ro.time_find(Tx, iters=40), 5)
                extras["find_o1dn_%s_ms" % nm] = round(oms, 5)
                extras["find_o1dn_%s_GBps" % nm] = round((algorithmic_bytes_raycast(n_rays, len(f), 1) + 12 * n_rays) / (oms * 1e-3) / 1e9, 1)
                ro.close()
            hmr.release()
            # closest-point correspondences on the C2 dataset, and a 16x900 scan (rays in flight below the chip's
            # width: the four-lanes-per-ray traversal is selected automatically)
            cpc = ra.CPCHip(hm)
            cpc.setTsb(T.identity())
            cpc.params.max_dist = 1.0
            rcc.find(syn.pose_c2_truth())
            mv = rcc.modelView()
            cpc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
            for tracking, key in ((False, "cpc_find_cold_ms"), (True, "cpc_find_ms")):
                # tracking (default): the previous call's triangle bounds the search -- the registration loop's steady state
                cpc.set_tracking(tracking)
                cpc.find(est)
                t1 = time.perf_counter()
                for _ in range(20):
                    cpc.find(est)
                dt = (time.perf_counter() - t1) / 20
                extras[key] = round(dt * 1e3, 4)
            extras["cpc_closest_points_per_s"] = round(n_rays / dt, 1)
            # the cold query WITHOUT the map's near grid (round 3's cold; rmclhip_rcc_set_cpc_grid 0): no seed at all, as rtcPointQuery
            cpc.set_tracking(False)
            cpc.set_grid(False)
            cpc.find(est)
            t1 = time.perf_counter()
            for _ in range(20):
                cpc.find(est)
            extras["cpc_find_cold_no_grid_ms"] = round((time.perf_counter() - t1) / 20 * 1e3, 4)
            cpc.set_grid(True)
            # opt-in: search only within max_dist (hits and hit outputs unchanged; rmclhip_rcc_set_cpc_bounded), cold
            cpc.set_tracking(False)
            cpc.set_bounded(True)
            cpc.find(est)
            t1 = time.perf_counter()
            for _ in range(20):
                cpc.find(est)
            extras["cpc_find_cold_bounded_ms"] = round((time.perf_counter() - t1) / 20 * 1e3, 4)
            cpc.close()
            # TWO operators (two sensors of one node, or two scans of a stream in flight): each has its own stream and buffers, their
            # finds alternate without waiting for each other.  A single 128x1024 scan leaves the chip partly idle -- its launch ends
            # with its slowest wave while the average wave takes two thirds of that, plus ~2.3 us between dependent launches --; this
            # is the throughput when the next scan may start under the previous one's tail.  NOT the headline (one scan at a time).
            pair = []
            for k in range(2):
                o = ra.RCCHipSpherical(hm)
                o.setTsb(T.identity())
                o.setModel(model)
                o.find(Tbm)
                pair.append(o)
            reps = 100
            ts = []
            for _ in range(5):
                for o in pair:
                    o.sync()
                t1 = time.perf_counter()
                for _ in range(reps):
                    for o in pair:
                        o.find_async(Tbm)
                for o in pair:
                    o.sync()
                ts.append((time.perf_counter() - t1) / (2 * reps))
            two_ms = sorted(ts)[2] * 1e3
            extras["find_two_operators_in_flight_ms_per_scan"] = round(two_ms, 5)
            extras["find_two_operators_in_flight_rays_per_s"] = round(n_rays / (two_ms * 1e-3), 1)
            for o in pair:
                o.close()
            small = ra.RCCHipSpherical(hm)
            small.setTsb(T.identity())
            small.setModel(syn.model_vlp16_900())
            sms = small.time_find(Tbm, iters=100)
            extras["find_16x900_ms"] = round(sms, 5)
            extras["find_16x900_rays_per_s"] = round(16 * 900 / (sms * 1e-3), 1)
            # measurement-driven choice of the traversal on the operator's own map / model (rmclhip_rcc_autotune) instead of the rule
            k_t, ms_t = small.autotune(Tbm)
            extras["find_16x900_autotuned"] = {"rule_kind": 2, "chosen_kind": k_t, "ms": round(ms_t, 5)}
            # the reference's own (stale) benchmark shape: 1000 poses x vlp16_900 per correct() call, sphere with
            # 100k faces (lidar_corrector_{embree,optix}_benchmark.cpp; BASELINE.md: OptiX 73.7 k, Embree 5.5 k
            # corrections/s on the authors' hardware)
            rng = np.random.RandomState(1)
            v1poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng.uniform(-3, 3)))
                                for _ in range(1000)], dtype=T.TRANSFORM)
            small.find(T.identity())
            small.set_dataset_from_ranges(small.modelView()["ranges"])
            small.params.max_dist = 1.0
            small.correct_batch(v1poses)
            t1 = time.perf_counter()
            for _ in range(3):
                small.correct_batch(v1poses)
            dt = (time.perf_counter() - t1) / 3
            extras["v1_bench_1000x16x900_ms"] = round(dt * 1e3, 4)
            extras["v1_bench_rays_per_s"] = round(1000 * 16 * 900 / dt, 1)
            extras["v1_bench_pose_corrections_per_s"] = round(1000 / dt, 1)
            extras["v1_bench_poses"] = "1000 DIFFERENT poses (+-1 m, any yaw); the reference's benchmark passes 1000 copies of ONE pose (z + 0.2 m): v1_bench_identical_poses_*"
            same = np.array([T.transform_from_rpy((0.0, 0.0, 0.2), (0, 0, 0))] * 1000, dtype=T.TRANSFORM)
            small.correct_batch(same)
            t1 = time.perf_counter()
            for _ in range(3):
                small.correct_batch(same)
            dts = (time.perf_counter() - t1) / 3
            extras["v1_bench_identical_poses_ms"] = round(dts * 1e3, 4)
            extras["v1_bench_identical_poses_pose_corrections_per_s"] = round(1000 / dts, 1)
            # the only figures the reference records for this shape (source comments, the authors' own machines, NOT this hardware):
            extras["v1_bench_reference_source_comments"] = {"optix_gpu_100k_faces_pose_corrections_per_s": 73700, "embree_cpu_100k_faces_pose_corrections_per_s": 5464,
                                                            "where": "rmcl_ros/src/benchmarks/lidar_corrector_{optix,embree}_benchmark.cpp:161 / :144 (BASELINE.md)"}
            # stage split in the reference benchmark's terms (lidar_corrector_embree_benchmark.cpp:185-190 records Sim 96.8 %,
            # Red 3.2 %, SVD 0.015 % on its CPU): sim = the batched find (HIP events), the rest = batched reduction + per-pose
            # solve + result download (one launch each, host clock)
            sim_ms = small.time_find_batch(v1poses, iters=5)
            extras["v1_bench_stage_split_ms"] = {"sim": round(sim_ms, 4), "red_svd_download": round(dt * 1e3 - sim_ms, 4)}
            small.close()
            # two sensors through the device-resident N-sensor loop (rmclhip_micp_correct_once): 128x1024 + 16x900, 10 iterations
            sA, sB = ra.RCCHipSpherical(hm), ra.RCCHipSpherical(hm)
            sens = []
            for rc_, mdl in ((sA, model), (sB, syn.model_vlp16_900())):
                rc_.setTsb(T.identity())
                rc_.setModel(mdl)
                rc_.find(syn.pose_c2_truth())
                rc_.set_dataset_from_ranges(rc_.modelView()["ranges"])
                rc_.params.max_dist, rc_.adaptive_max_dist_min = 1.0, 0.15
                sens.append(ra.MICPSensor("s%d" % len(sens), rc_, Tsb=T.identity(), Tbo=T.identity()))
            loc = ra.MICPLocalization(sens, optimization_iterations=10)
            loc.Tom_ = est
            extras["micp_two_sensors_device_loop_ms"] = round(_median_call_ms(lambda: (setattr(loc, "Tom_", est), loc.correctOnce(device_loop=True)), reps=15), 4)
            extras["micp_two_sensors_host_loop_ms"] = round(_median_call_ms(lambda: (setattr(loc, "Tom_", est), loc.correctOnce()), reps=15), 4)
            # the same call at the C ABI, without the Python host class around it (arguments built once)
            import ctypes as C
            from rmcl_amd import _capi as _c
            hnd = (C.c_void_p * 2)(sA._h, sB._h)
            Tbo2, w2 = np.array([T.identity(), T.identity()], dtype=T.TRANSFORM), np.ones(2, np.float64)
            Tin, Tout, mrg = np.ascontiguousarray(est, dtype=T.TRANSFORM).reshape(1), np.zeros(1, T.TRANSFORM), np.zeros(1, T.CROSS_STATISTICS)
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            extras["micp_two_sensors_device_loop_cabi_ms"] = round(_median_call_ms(
                lambda: _c.check(_c.lib().rmclhip_micp_correct_once(hnd, 2, vp(Tin), vp(Tbo2), vp(w2), 10, 0.0, vp(Tout), vp(mrg))), reps=25), 4)
            # (the key's "device_loop" is round 3's name for rmclhip_micp_correct_once; since round 4 the iterations of that call run on the
            # host from the published moments -- the same figure under a name that says so)
            extras["micp_two_sensors_cabi_ms"] = extras["micp_two_sensors_device_loop_cabi_ms"]
            sA.close()
            sB.close()
            # the particle filter through the multi-GPU C ABI on this one GPU (RCCL ncclCommInitAll + all-gather + all-reduces)
            # (under a multi-rank launch torch's own RCCL process group is live in this process: the in-library communicator is then the
            # in-process loopback one, so that two RCCL instances never meet; at N = 1 it is RCCL's ncclCommInitAll)
            shp = ra.ShardedParticleFilterHip(v, f, devices=(local_rank,), loopback=(world > 1))
            extras["pf_sharded_cabi_communicator"] = "loopback (in-process)" if world > 1 else "rccl ncclCommInitAll"
            extras["pf_sharded_cabi_collective_ranks"] = list(shp.collective_ranks())   # ncclCommCount of the C-ABI communicator
            pposes, pattrs = syn.uniform_particles(100000, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
            shp.set_particles(pposes, pattrs)
            pdirs = syn.model_directions(syn.model_pf16())
            pbeams = ra.beams_from_points(pdirs * np.float32(6.0))
            extras["pf_sharded_cabi_update_allgather_ms"] = round(_median_call_ms(lambda: shp.update(pbeams, T.identity()), reps=5, warm=1), 4)
            extras["pf_sharded_cabi_pose_estimate_ms"] = round(_median_call_ms(lambda: shp.pose_estimate(), reps=9, warm=1), 4)
            # (round 6: no all-reduce behind this call any more -- every device reduces its copy of the gathered weights; the key keeps its name)
            extras["pf_sharded_cabi_allreduce_stats_ms"] = round(_median_call_ms(lambda: shp.stats(), reps=9, warm=1), 4)
            shp.close()

        if rank == 0 and not args.no_extras:
            # the reference's own 1 M- and 10 M-face rows (lidar_corrector_{optix,embree}_benchmark.cpp:144-152, :161-169): maps that leave
            # the L2s (1 M: 128 MB) and the MALL (10 M: 1.29 GB)
            extras["large_maps"] = _large_maps(ra, syn, T, np, ctx)
            # config C5 at FULL size as one run on this ONE GPU: eight loopback ranks (the ndev = 8 code of the C ABI, in-process copies
            # instead of RCCL), 1 M particles x 256 beams, 1 M-triangle sphere: update + gather + {sum, max}; NOT a multi-GPU figure
            extras["c5_full_loopback8"] = _c5_full_loopback(ra, syn, T, np)
            # meshes whose SAH tree is deeper than the kernels' 64-entry stack (rounds 1-4 refused them with RMCLHIP_ERR_UNSUPPORTED): the
            # builder bounds the stack by construction since round 5 (bvh_build.cpp: height budget + tallest-first collapse)
            extras["stack_bounded_maps"] = _stack_bounded_maps(ra, syn, T, np, ctx)

        # config C5's per-GPU term, on every rank and for every N (the only BASELINE config that shards)
        if not args.no_extras:
            blk = _pf_sharded_block(ra, syn, T, np, torch, dist, ctx, rank, world)
            if rank == 0:
                extras["pf_sharded"] = blk
            # pose batches over the ranks (no exchange): the N-GPU pose-corrections/s
            pose_batches = _pose_batch_block(ra, syn, T, np, torch, dist, hm, rank, world)
            if rank == 0:
                extras["pose_batches"] = pose_batches

        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc  # cpu_baseline leg only: the oracle is the thing timed, never the product path
            m = orc.Mesh(v, f)
            # the oracle keeps ONE pool of worker threads alive across calls and deals the scan out in chunks of 512 rays,
            # chunk c to worker c mod N (no shared counter): it runs on ALL host threads; 8 scans per call amortise the
            # wake-up of the pool; output arrays are reused (fresh ones cost first-touch page faults in every worker)
            per_call = 8
            Tb = np.array([Tbm] * per_call, dtype=T.TRANSFORM)
            # threads: what the cgroup lets this process use (the GPU boxes show 256 hardware threads under a 16-CPU quota)
            usable, visible, quota = _cpu_quota()
            # round 4: the timed path is the oracle's BVH4 walk with SSE slab tests and per-row / per-column trig tables (bvh=2: what a CPU
            # ray caster of Embree's class does per ray; results bit-identical to the scalar BVH2 walk, tests/test_oracle.py), the
            # scalar BVH2 walk of rounds 1-3 is reported beside it
            CPU_WALK = 2
            buf = m.simulate_spherical(model, T.identity(), Tb, bvh=CPU_WALK, nthreads=usable)
            reps, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 7.0:
                m.simulate_spherical(model, T.identity(), Tb, bvh=CPU_WALK, nthreads=usable, out=buf)
                reps += per_call
            dt = time.perf_counter() - t1
            rs_, ts = 0, time.perf_counter()
            while time.perf_counter() - ts < 2.5:
                m.simulate_spherical(model, T.identity(), Tb, bvh=True, nthreads=usable, out=buf)
                rs_ += per_call
            dts = time.perf_counter() - ts
            # all visible threads, for the record (round 2 reported this row: oversubscribed under the quota)
            ra_, ta = 0, time.perf_counter()
            while visible != usable and time.perf_counter() - ta < 2.0:
                m.simulate_spherical(model, T.identity(), Tb, bvh=CPU_WALK, nthreads=visible, out=buf)
                ra_ += per_call
            dta = time.perf_counter() - ta
            # 1-thread row: the same scan, one core, ~3 s
            one = m.simulate_spherical(model, T.identity(), Tb[:1], bvh=CPU_WALK, nthreads=1)
            r1, t2 = 0, time.perf_counter()
            while time.perf_counter() - t2 < 3.0:
                m.simulate_spherical(model, T.identity(), Tb[:1], bvh=CPU_WALK, nthreads=1, out=one)
                r1 += 1
            dt1 = time.perf_counter() - t2
            cpu = {"value": round(reps * n_rays / dt, 1), "unit": "rays/s", "cores": usable, "kind": "port",
                   "one_thread_value": round(r1 * n_rays / dt1, 1),
                   "scalar_bvh2_walk_value": round(rs_ * n_rays / dts, 1),
                   "all_visible_threads_value": (round(ra_ * n_rays / dta, 1) if ra_ else None),
                   "host": {"visible_threads": visible, "cgroup_cpu_quota": quota, "threads_used": usable},
                   "sample": "%d x the same 128x1024 / 100k-triangle scan (8 per call) in %.1f s on %d threads = the CPUs this "
                             "process may use (affinity %d, cgroup cpu.max quota %s; persistent pool, static chunks of 512 rays) + "
                             "%d scans in %.1f s on ONE thread; CPU oracle, BVH4 walk with SSE slab tests + scalar Moeller-Trumbore "
                             "(same intersector and tie-break as the checker's scalar BVH2 walk, whose rate on the same threads is "
                             "scalar_bvh2_walk_value), five output attributes" % (reps, dt, usable, visible, ("%.1f CPUs" % quota) if quota else "none", r1, dt1)}
            # ---- round 5: stated CPU baselines for the other two metrics -- never targets.  The oracle's loops, timed on the
            # same host CPUs: the find (BVH4 + SSE walk) and the p2l reduction (one pass of raw double sums) on `usable` threads, the 3x3
            # solve on one; the reference's CorrespondencesCPU::computeCrossStatistics is one call per iteration as well.
            cpu.update(_cpu_baselines_c3_c4_v1(orc, m, np, syn, T, model, ra, usable))
            # traversal-traffic view of the roofline: B_trav = sum over rays of nodes_visited * 32 + triangles
            # tested * 36, counted by the instrumented oracle on the identical rays and a BVH2 / one-triangle-per-leaf
            # reference tree (deterministic), against the aggregate L2 rate of 34.5 TB/s
            m1 = orc.Mesh(v, f, max_leaf=1)
            cnt = m1.simulate_spherical(model, T.identity(), Tb[:1], bvh=True, nthreads=usable, counters=True,
                                        want=("hits",))["counters"]
            b_trav = cnt["nodes_visited"] * 32 + cnt["tris_tested"] * 36
            extras["traversal_view"] = {
                "B_trav_bytes_per_scan": int(b_trav), "B_trav_bytes_per_ray": round(b_trav / n_rays, 1),
                "nodes_visited_per_ray": round(cnt["nodes_visited"] / n_rays, 2), "tris_tested_per_ray": round(cnt["tris_tested"] / n_rays, 2),
                "achieved_TBps": round(b_trav / (kernel_ms * 1e-3) / 1e12, 3), "l2_aggregate_peak_TBps": 34.5,
                "frac_of_l2": round(b_trav / (kernel_ms * 1e-3) / 1e12 / 34.5, 4),
                "frac_of_hbm_IF_it_came_from_hbm": round(b_trav / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "tree": "NOMINAL figure of SURVEY.md 8(d): oracle BVH2, 1 triangle per leaf, 32-B nodes / 36-B triangles, same rays as "
                        "the timed scan -- NOT what the product's BVH4 moves (see product_tree); none of it is HBM traffic "
                        "(roofline.traffic: the map is served from L2 / MALL)"}
            pt = product_tree_bytes("c2_sphere100k")
            if pt is not None:
                # what the PRODUCT's tree moves for these rays: node visits x 128 B + triangle records x 64 B, counted by the CPU model
                # of the product's traversal (tools/product_tree_bytes.py -> profiles/traversal_product_tree.json)
                extras["traversal_view"]["product_tree"] = {
                    "bytes_per_scan": pt["bytes_per_scan"], "bytes_per_ray": pt["bytes_per_ray"],
                    "node_visits_per_ray": pt["node_visits_per_ray"], "records_per_ray": pt["records_per_ray"],
                    "achieved_TBps": round(pt["bytes_per_scan"] / (kernel_ms * 1e-3) / 1e12, 3),
                    "frac_of_l2": round(pt["bytes_per_scan"] / (kernel_ms * 1e-3) / 1e12 / 34.5, 4),
                    "source": "profiles/traversal_product_tree.json (tools/product_tree_bytes.py: BVH4 of the product's builder, "
                              "frontier start, 128-B nodes, 64-B records)"}
        metric = "ray-mesh intersections/s (128x1024 scan, 100k-tri mesh)"
        unit = "rays/s"
        workload = ("C2: 1 pose x 128x1024 spherical LiDAR, UV-sphere 100k triangles, find() only, 5 output "
                    "attributes; N>1 = independent replicas")
        parallelism = "replicas%d" % world
    else:
        # ---- particle filter, weak scaling: 100k particles per GPU, one all-gather of the weights per step
        from rmcl_amd import distributed as D
        n_local, n_beams = 100000, 256
        n_total = n_local * world
        poses_all, attrs_all = syn.uniform_particles(n_total, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
        lo, hi = D.shard_bounds(n_total, rank, world)
        dirs = syn.model_directions(syn.model_pf16())
        beams = ra.beams_from_points(dirs * np.float32(6.0))
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.setInput(beams, T.identity())
        d_poses = ra.DeviceArray.from_host(ctx, poses_all[lo:hi])
        d_attrs = ra.DeviceArray.from_host(ctx, attrs_all[lo:hi])
        w_local = torch.empty(hi - lo, dtype=torch.float32, device="cuda")
        sharded = D.ShardedSensorUpdate(upd, n_total, rank, world)

        def step():
            if dist is not None:
                return sharded.update(d_poses, d_attrs, w_local)
            upd.update(d_poses, d_attrs, n_particles=hi - lo)
            upd.extract_weights(d_attrs, hi - lo, w_local.data_ptr())
            return w_local

        for _ in range(max(args.warmup, 1)):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        torch.cuda.synchronize()
        elapsed = max_over_ranks(t1 - t0)
        units_per_step = n_local * n_beams
        kernel_ms = upd.time_update(d_poses, d_attrs, hi - lo, iters=5)
        b_alg = algorithmic_bytes_pf(hi - lo, n_beams)
        # (the updater runs on the library's default variant -- this workload sets none --, which launches the accumulation form:
        # capi_pf.cpp / kernels.hip launch_pf_update; a run under rmclhip_pf_set_variant would need the label re-derived)
        kname = "k_pf_update_v3<19 rows, leaves <= 2, accumulate> (library default variant)"
        traffic = measured_traffic("k_pf_update_v3")
        extras["particle_updates_per_s"] = round(world * args.steps * n_local / elapsed, 1)
        extras["allgather_bytes"] = 4 * n_total
        metric = "particle-beam evaluations/s (100k particles x 256 beams per GPU, 100k-tri mesh)"
        unit = "rays/s"
        workload = "C4 per GPU: 100k particles x 16x16 beams, UV-sphere 100k triangles, fused update + weight all-gather"
        parallelism = "particles-sharded dp%d + all_gather(4B x N)" % world
        n_rays = units_per_step

    achieved = b_alg / (kernel_ms * 1e-3) / 1e9
    # `frac` is the contract's figure (algorithmic bytes / kernel time against the HBM peak).  It is NOT what bounds these kernels:
    # measured HBM traffic is 0.44x the algorithmic bytes for the find (the map is served from L2 / MALL; the launch ends with its
    # slowest wave's chain of dependent fetches) and 1.8x for the particle filter (round 5: the order-independent accumulation keeps no
    # per-beam storage; round 4's error scratch made it ~20x) -- that kernel is bound by VALU issue and occupancy
    roofline = {"bound": "latency / issue (frac = the contract's HBM fraction)", "contract_bound": "hbm",
                "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(kernel_ms, 5),
                "kernel_units_per_s": round(units_per_step / (kernel_ms * 1e-3), 1)}
    if rank == 0:
        out = {
            "metric": metric,
            "value": round(world * args.steps * units_per_step / elapsed, 1),
            "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "units_per_step_per_gpu": units_per_step, "triangles": int(len(f)),
                       "parallelism": parallelism, "kernel_variant": args.variant,
                       "traversal": (("autotuned (rmclhip_rcc_autotune): kind %d" % extras["headline_traversal"]["timed_kind"])
                                     if "headline_traversal" in extras else "automatic rule / --variant")},
            # whole-job pose-corrections/s (aggregate over the ranks, weak: a batch per GPU) for the reference's own batch shape and
            # for full-size scans; the single-scan 10-iteration correction (C3 schedule R) does not shard: rank 0's figure, per rank
            "pose_corrections_per_s": {
                "batch_1000x16x900": (pose_batches or {}).get("batch_1000x16x900", {}).get("weak", {}).get("pose_corrections_per_s"),
                "batch_64x128x1024": (pose_batches or {}).get("batch_64x128x1024", {}).get("weak", {}).get("pose_corrections_per_s"),
                "n_gpus": world, "scaling": "weak (one batch per GPU, block-partitioned pose lists, no collective)",
                "single_scan_10_iterations_per_rank": extras.get("c3_schedule_R_pose_corrections_per_s"),
                "single_scan_unchanged_caller_loop_per_rank": extras.get("c3_schedule_R_unchanged_caller_pose_corrections_per_s")},
            # the one BASELINE configuration that shards (C5: particles block-partitioned, one weight all-gather per step); at N = 1 the
            # per-GPU term.  At N > 1 `value` above is N independent replicas of C2 -- THIS block and pose_corrections_per_s are the
            # informative multi-GPU figures
            "particle_filter_sharded": ({k: extras["pf_sharded"].get(k) for k in ("shape", "c5_step_ms", "c5_shard_update_ms", "c5_allgather_ms",
                                                                               "particle_beam_evals_per_s", "collective", "collective_ranks",
                                                                               "collective_backend", "gathered_vector_length",
                                                                               "gathered_vector_complete_on_every_rank")}
                                        if isinstance(extras.get("pf_sharded"), dict) else None),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extras": extras,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same environment
    contract torch.distributed.run provides), pass rank 0's stdout through (the ONE JSON line), wait for all of them.  If a
    rank dies the others are terminated (they would wait in a collective forever)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    rc = 0
    alive = list(procs)
    while alive:
        for pr in list(alive):
            code = pr.poll()
            if code is None:
                continue
            alive.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in alive:
                    other.terminate()
        time.sleep(0.05)
    return rc


def _cpu_quota():
    """host threads this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU boxes of this pool show
    256 hardware threads but run the job in a cgroup with cpu.max = 1600000 100000, i.e. 16 CPUs: more threads than that
    only add scheduling overhead (round 2 measured 7.5x on "256 threads")."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = float(parts[0]) / float(parts[1])
            else:
                q = float(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        quota = q / float(fh2.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    eff = n if quota is None else max(1, min(n, int(math.ceil(quota))))
    return eff, n, quota


def _pose_batch_block(ra, syn, T, np, torch, dist, hm, rank, world):
    """pose-corrections/s over ranks (north_star: "at 1/2/4/8 GPUs"; SURVEY.md 8(e): pose batches shard by pose, no exchange): the
    v1 corrector shape of the reference's own benchmark -- 1000 poses x one 16x900 scan per correct()
    (lidar_corrector_optix_benchmark.cpp:86-133) -- and 64 poses x 128x1024, on the 100k-triangle sphere.
    weak: every rank corrects a whole batch of its own (what bench.py's `scaling` says); strong: ONE batch block-partitioned over the
    ranks with distributed.ShardedBatchCorrector.  Timed like the headline: barrier + synchronize on both sides, max over ranks."""
    from rmcl_amd import distributed as D

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    out = {"partition": "block partition of the pose list (rmcl_amd.distributed.shard_bounds), map replicated, NO collective on the data path",
           "reference_shape": "lidar_corrector_optix_benchmark.cpp:86-133 (1000 poses per correct(), vlp16_900)"}
    truth = syn.pose_c2_truth()
    for key, model, nposes, reps in (("batch_1000x16x900", syn.model_vlp16_900(0.0), 1000, 5), ("batch_64x128x1024", syn.model_c2(), 64, 5)):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        rcc.find(truth)
        rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
        rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
        rng = np.random.RandomState(1000 + rank)          # weak: a batch of its own per rank
        mine = np.array([T.mult(truth, T.transform_from_rpy(tuple(rng.uniform(-0.2, 0.2, 3)), (0, 0, rng.uniform(-0.05, 0.05))))
                         for _ in range(nposes)], dtype=T.TRANSFORM)
        rng0 = np.random.RandomState(999)                  # strong: the SAME list on every rank
        shared = np.array([T.mult(truth, T.transform_from_rpy(tuple(rng0.uniform(-0.2, 0.2, 3)), (0, 0, rng0.uniform(-0.05, 0.05))))
                           for _ in range(nposes)], dtype=T.TRANSFORM)
        sh = D.ShardedBatchCorrector(lambda blk: rcc.correct_batch(blk), rank, world)
        res = {}
        for mode in ("weak", "strong"):
            fn = (lambda: rcc.correct_batch(mine)) if mode == "weak" else (lambda: sh.correct(shared))
            fn()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = reduce_max(time.perf_counter() - t0) / reps
            total = nposes * (world if mode == "weak" else 1)
            res[mode] = {"ms_per_batch": round(dt * 1e3, 4), "poses_per_batch_all_ranks": total,
                         "pose_corrections_per_s": round(total / dt, 1)}
            sync_all()
        out[key] = res
        rcc.close()
    return out


def _pf_sharded_block(ra, syn, T, np, torch, dist, ctx, rank, world, n_local=125000, n_beams=256, n_tri=1000000):
    """config C5's per-GPU term on every rank: n_local particles x n_beams beams on the 1M-triangle sphere; the cloud of
    world x n_local particles is block-partitioned, the map replicated; per step: fused sensor update, extraction of the
    weights, ONE all-gather of 4 B x N (RCCL for world > 1), and the {sum, max} all-reduce the resampler needs."""
    from rmcl_amd import distributed as D
    v, f = syn.uv_sphere(n_tri)
    t0 = time.perf_counter()
    hm = ra.import_hip_map(ctx, v, f)
    build_s = time.perf_counter() - t0
    n_total = n_local * world
    lo, hi = D.shard_bounds(n_total, rank, world)
    # the same cloud on every world size: particle i is drawn from seed (42, i // n_local)
    poses, attrs = syn.uniform_particles(n_local, seed=42 + rank, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, T.identity())
    d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    w_local = torch.empty(hi - lo, dtype=torch.float32, device="cuda")
    sharded = D.ShardedSensorUpdate(upd, n_total, rank, world)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    rsm = ra.GladiatorResamplerHip(ctx)
    gathered_len = {"n": None}

    def step():
        # sensor update -> ONE all-gather of the weights -> {sum, max} on this rank's own copy of the gathered vector (round 6: no
        # second collective; rmclhip_resampler_compute_stats_weights = the single-GPU kernel and order)
        if dist is not None:
            w = sharded.update(d_poses, d_attrs, w_local)
        else:
            upd.update(d_poses, d_attrs, n_particles=hi - lo)
            upd.extract_weights(d_attrs, hi - lo, w_local.data_ptr())
            w = w_local
        gathered_len["n"] = int(w.shape[0])
        D.gathered_sum_max(w[:n_total], rsm.compute_stats_weights)
        return w

    for _ in range(2):
        step()
    sync_all()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    step_ms = reduce_max((time.perf_counter() - t0) / reps * 1e3)
    sync_all()
    update_ms = reduce_max(upd.time_update(d_poses, d_attrs, hi - lo, iters=3))
    ag_ms = 0.0
    if dist is not None:
        for _ in range(3):
            D.allgather_weights(w_local, n_total)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(20):
            D.allgather_weights(w_local, n_total)
        torch.cuda.synchronize()
        ag_ms = reduce_max((time.perf_counter() - t0) / 20 * 1e3)
    # what the collective itself saw: the process group's size and backend, and the gathered vector's length CHECKED ON EVERY RANK
    # (a scaling run then shows that RCCL moved N ranks' shards, not that N processes ran side by side)
    len_ok = 1.0 if (gathered_len["n"] is not None and gathered_len["n"] >= n_total) else 0.0
    if dist is not None:
        tt = torch.tensor([len_ok], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        len_ok = float(tt.item())
    rsm.close()
    upd.close()
    hm.release()
    return {"shape": "C5 per GPU: %d particles x %d beams, UV-sphere %d triangles, %d GPU(s), %d particles in total" %
                     (n_local, n_beams, n_tri, world, n_total),
            "c5_shard_update_ms": round(update_ms, 4), "c5_step_ms": round(step_ms, 4), "c5_allgather_ms": round(ag_ms, 4),
            "allgather_bytes": 4 * n_total, "collective": ((("RCCL" if str(dist.get_backend()) == "nccl" else str(dist.get_backend()) + " (plumbing test, not RCCL)") +
                                                            " all_gather_into_tensor; {sum, max} from the gathered vector on every rank (no all-reduce)") if dist is not None else "none (1 GPU)"),
            "collective_ranks": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (str(dist.get_backend()) if dist is not None else "none"),
            "gathered_vector_length": gathered_len["n"], "gathered_vector_complete_on_every_rank": bool(len_ok == 1.0),
            "particle_beam_evals_per_s": round(n_total * n_beams / (step_ms * 1e-3), 1),
            "particle_updates_per_s": round(n_total / (step_ms * 1e-3), 1), "map_build_upload_s": round(build_s, 2)}


def _large_maps(ra, syn, T, np, ctx):
    """C2 find, the v1 batch (1000 x 16x900) and C3 schedule (R) on the 1 M- and 10 M-face spheres.  `find_*_rotating_poses`: 16
    different poses in turn -- the same scan repeated would live on what the previous launch left in the 256 MB MALL."""
    out = {}
    ref = {1000000: {"optix_rays_per_s": 852e6, "optix_pose_corrections_per_s": 59.2e3, "embree_rays_per_s": 71.6e6},
           10000000: {"optix_rays_per_s": 462e6, "optix_pose_corrections_per_s": 32.1e3, "embree_rays_per_s": 31.6e6}}
    model = syn.model_c2()
    n_rays = model.phi.size * model.theta.size
    rng = np.random.RandomState(7)
    poses = [syn.pose_c2_truth()] + [T.transform_from_rpy(tuple(rng.uniform(-3.0, 3.0, 3)), tuple(rng.uniform(-0.4, 0.4, 2)) + (rng.uniform(-3.1, 3.1),))
                                     for _ in range(15)]
    for tag, nf in (("1m", 1000000), ("10m", 10000000)):
        v, f = syn.uv_sphere(nf)
        t0 = time.perf_counter()
        hm = ra.import_hip_map(ctx, v, f)
        r = {"map_build_s": round(time.perf_counter() - t0, 3)}
        info = hm.info()
        r["map"] = {k: info[k] for k in ("n_faces", "n_nodes", "stack_need", "device_bytes")}
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        Tbm = poses[0]
        ms = median_kernel_ms(lambda: rcc.time_find(Tbm, iters=20), 5)
        r["find_sphere%s_ms" % tag] = round(ms, 5)
        r["find_sphere%s_rays_per_s" % tag] = round(n_rays / (ms * 1e-3), 1)
        for P in poses:
            rcc.find(P)

        def rot():
            rcc.sync()
            t1 = time.perf_counter()
            for _ in range(4):
                for P in poses:
                    rcc.find_async(P)
            rcc.sync()
            return (time.perf_counter() - t1) / (4 * len(poses)) * 1e3
        msr = median_kernel_ms(rot, 5)
        r["find_sphere%s_rotating_poses_ms" % tag] = round(msr, 5)
        r["find_sphere%s_rotating_poses_rays_per_s" % tag] = round(n_rays / (msr * 1e-3), 1)
        r["find_kind"] = rcc.find_variant(1)
        tr = measured_traffic("k_find_kind23_c2_scan_16_poses_in_turn_sphere%s" % tag)
        if tr is not None:
            r["find_sphere%s_rotating_poses_measured_traffic_bytes" % tag] = tr
            r["find_sphere%s_rotating_poses_measured_TBps" % tag] = round(tr / (msr * 1e-3) / 1e12, 3)
        rcc.find(Tbm)
        rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
        rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
        est = T.mult(Tbm, syn.pose_c2_perturbation())
        rcc.correct_once(est, T.identity(), 10, 0.0, False)
        cms = rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=30)
        r["c3_schedule_R_%s_ms" % tag] = round(cms, 4)
        r["c3_schedule_R_%s_pose_corrections_per_s" % tag] = round(1e3 / cms, 1)
        rcc.close()
        small = ra.RCCHipSpherical(hm)
        small.setTsb(T.identity())
        small.setModel(syn.model_vlp16_900())
        rng1 = np.random.RandomState(1)
        v1poses = np.array([T.transform_from_rpy(tuple(rng1.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng1.uniform(-3, 3)))
                            for _ in range(1000)], dtype=T.TRANSFORM)
        small.find(T.identity())
        small.set_dataset_from_ranges(small.modelView()["ranges"])
        small.params.max_dist = 1.0
        small.correct_batch(v1poses)
        dt = _median_call_ms(lambda: small.correct_batch(v1poses), reps=5, warm=1)
        r["v1_bench_1000x16x900_%s_ms" % tag] = round(dt, 4)
        r["v1_bench_1000x16x900_%s_rays_per_s" % tag] = round(1000 * 16 * 900 / (dt * 1e-3), 1)
        r["v1_bench_1000x16x900_%s_pose_corrections_per_s" % tag] = round(1e6 / dt, 1)
        r["v1_bench_1000x16x900_%s_find_ms" % tag] = round(small.time_find_batch(v1poses, iters=3), 4)
        tr = measured_traffic("k_find_kind24_v1_batch_1000x16x900_sphere%s" % tag)
        if tr is not None:
            r["v1_bench_1000x16x900_%s_find_measured_traffic_bytes" % tag] = tr
            r["v1_bench_1000x16x900_%s_find_measured_TBps" % tag] = round(tr / (r["v1_bench_1000x16x900_%s_find_ms" % tag] * 1e-3) / 1e12, 3)
        small.close()
        r["reference_source_comments_other_hardware"] = ref[nf]
        hm.release()
        out["sphere" + tag] = r
        del v, f
    out["note"] = ("the reference records these rows for 1000 copies of ONE pose on its authors' machines (BASELINE.md); here 1000 DIFFERENT poses. "
                   "measured_traffic = L2-side FETCH_SIZE + WRITE_SIZE of profiles/traffic.json (tools/pmc_large_maps.sh); FETCH_SIZE tallies 64 B per "
                   "128-B line request (profiles/r05_fetch_size_calibration.txt): exact for 64-B nodes / records, a lower bound otherwise")
    return out


def _stack_bounded_maps(ra, syn, T, np, ctx):
    out = {}
    f32 = np.float32
    H, W = 64, 512
    for name, (v, f), model, Tbm in (
            ("exp_chain_2000", syn.exp_chain(2000, 1.05), T.spherical_model(f32(-0.2), f32(0.4 / (H - 1)), H, f32(-0.3), f32(0.6 / W), W, f32(0.0), f32(1e30)),
             T.transform_from_rpy((-1.0, 0.04, 0.03), (0.0, 0.0, 0.0))),
            ("nested_triangles_200", syn.nested_triangles(200, 1.2, 1e-3), T.spherical_model(f32(-0.4), f32(1.85 / (H - 1)), H, f32(-math.pi), f32(2 * math.pi / W), W, f32(0.0), f32(1e12)),
             T.transform_from_rpy((0.001, -0.002, -1.0), (0.0, 0.0, 0.3))),
            ("sliver_fan_200k", syn.sliver_fan(200000), T.spherical_model(f32(-1.5), f32(3.0 / (H - 1)), H, f32(-math.pi), f32(2 * math.pi / W), W, f32(0.01), f32(1e6)),
             T.transform_from_rpy((1.0, 2.0, 3.0), (0.1, 0.2, 0.3)))):
        hm = ra.import_hip_map(ctx, v, f)
        info = hm.info()
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        first = rcc.time_find(Tbm, iters=1)
        ms = median_kernel_ms(lambda: rcc.time_find(Tbm, iters=20), 5) if first < 1.0 else min(first, rcc.time_find(Tbm, iters=2))
        rcc.find(Tbm)
        nh = int(rcc.modelView()["hits"].sum())
        out[name] = {"n_faces": info["n_faces"], "stack_need": info["stack_need"], "height_fallbacks": info["height_fallbacks"],
                     "guarded_nodes": info["guarded_nodes"], "find_%dx%d_us" % (H, W): round(ms * 1e3, 2), "rays_hit": nh}
        rcc.close()
        hm.release()
    out["note"] = ("adversarial meshes, reported because they load at all now; the sliver fan (every triangle's box reaches the shared apex) is the classic "
                   "worst case of an object-split BVH -- rays near the disc's plane visit thousands of overlapping boxes (tens of ms per 32 k rays); "
                   "round 6's spatial splits help CAD-like mixes (profiles/r06_sbvh_ab.txt), not this fan: every sliver crosses every plane through the "
                   "apex, so a split duplicates all of them (DESIGN.md section 3)")
    return out


def _c5_full_loopback(ra, syn, T, np):
    v, f = syn.uv_sphere(1000000)
    n = 1000000
    poses, attrs = syn.uniform_particles(n, seed=5, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
    sh = ra.ShardedParticleFilterHip(v, f, devices=(0,) * 8, loopback=True)
    sh.set_particles(poses, attrs)
    ms = _median_call_ms(lambda: sh.step(beams, T.identity()), reps=3, warm=1)
    msr = _median_call_ms(lambda: sh.step(beams, T.identity(), T_bnew_bold=T.transform_from_rpy((0.01, 0.0, 0.0), (0, 0, 0.001)), forget_rate=0.05,
                                          resample="gladiator", seed=3), reps=3, warm=1)
    sh.close()
    return {"label": "ONE GPU, eight LOOPBACK ranks (in-process copies, no RCCL, no xGMI): the ndev = 8 code path at C5's full size, not a multi-GPU measurement",
            "shape": "1 000 000 particles x 256 beams, UV-sphere 1 000 000 triangles",
            "c5_full_loopback8_step_ms": round(ms, 3), "c5_full_loopback8_step_with_motion_and_tournament_ms": round(msr, 3),
            "particle_beam_evals_per_s": round(n * 256 / (ms * 1e-3), 1)}


def _cpu_baselines_c3_c4_v1(orc, m, np, syn, T, model, ra, usable):
    """the same host CPUs, the other two metrics.  The oracle's find (BVH4 + SSE walk) and its one-pass threaded reduction
    (orc_statistics_p2l_fast: what a tuned CPU reduction does; the checker's element-by-element forms are 10x slower and are not timed):
    oracle-side C3 (find + 10 x statistics_p2l + umeyama: micp_localization.cpp:900-964), C4 sensor update (PCDSensorUpdaterEmbree.cpp:
    290-342, grain 128 like :330-331) and the v1 batch (1000 x 16x900: lidar_corrector_embree_benchmark.cpp:127-135) on the host CPUs"""
    out = {}
    ident = T.identity()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, ident, truth, bvh=2, nthreads=usable, want=("ranges",))
    dirs = orc.spherical_directions(model)
    ds = (dirs * meas["ranges"][:, None]).astype(np.float32)
    mask = np.where((meas["ranges"] < np.float32(model.range.min)) | (meas["ranges"] > np.float32(model.range.max)), 0, 1).astype(np.uint8)

    def c3(nthreads, buf):
        sim = m.simulate_spherical(model, ident, est, bvh=2, nthreads=nthreads, want=("hits", "points", "normals"), out=buf)
        T_delta = ident
        for _ in range(10):
            s_ = orc.statistics_p2l_fast(T_delta, ds, mask, sim["points"], sim["normals"], sim["hits"], 1.0, nthreads)
            T_delta = orc.tmult(T_delta, orc.umeyama(orc.cs_merge(orc.cs_identity(), s_)))
        return T_delta

    for key, nt, budget in (("c3_pose_corrections_per_s", usable, 2.5), ("c3_pose_corrections_per_s_one_thread", 1, 2.5)):
        buf = m.simulate_spherical(model, ident, est, bvh=2, nthreads=nt, want=("hits", "points", "normals"))
        c3(nt, buf)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            c3(nt, buf)
            reps += 1
        out[key] = round(reps / (time.perf_counter() - t0), 2)
    out["c3_sample"] = ("C3 = 1 find (128x1024, sphere-100k, BVH4 + SSE walk) + 10 x (statistics_p2l as one pass of raw double sums + Umeyama) per "
                        "correction; find and reduction on the usable threads (resp. one), the 3x3 solve on one; ~2.5 s each")
    # C4: sensor update of a bounded sample of the 100 000 x 256 workload (20 000 particles), static chunks of <= 128 particles per fetch
    n_s = 20000
    poses, attrs = syn.uniform_particles(n_s, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
    a = attrs.copy()
    m.pf_update(poses[:2000], a[:2000], beams, ident, orc.pf_params(), bvh=2, nthreads=usable)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.5:
        a = attrs.copy()
        m.pf_update(poses, a, beams, ident, orc.pf_params(), bvh=2, nthreads=usable)
        reps += 1
    dt = time.perf_counter() - t0
    out["c4_particle_beam_evals_per_s"] = round(reps * n_s * len(beams) / dt, 1)
    out["c4_particle_updates_per_s"] = round(reps * n_s / dt, 1)
    out["c4_sample"] = "%d x (20 000 particles x 256 beams, sphere-100k) in %.1f s on %d threads: the oracle's sensorUpdate (BVH4 + SSE walk, double exp, in-order Gaussian1D merge)" % (reps, dt, usable)
    # v1 batch: 1000 poses x 16x900, per pose raycast + reduce (Tpre = I) + Umeyama
    small = syn.model_vlp16_900()
    rng = np.random.RandomState(1)
    v1poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng.uniform(-3, 3))) for _ in range(1000)], dtype=T.TRANSFORM)
    meas = m.simulate_spherical(small, ident, ident, bvh=2, nthreads=usable, want=("ranges",))
    sdirs = orc.spherical_directions(small)
    sds = (sdirs * meas["ranges"][:, None]).astype(np.float32)
    smask = np.ones(len(sds), np.uint8)
    nray = len(sds)
    buf = m.simulate_spherical(small, ident, v1poses, bvh=2, nthreads=usable, want=("hits", "points", "normals"))
    t0 = time.perf_counter()
    sim = m.simulate_spherical(small, ident, v1poses, bvh=2, nthreads=usable, want=("hits", "points", "normals"), out=buf)
    t_sim = time.perf_counter() - t0
    for i in range(len(v1poses)):
        sl = slice(i * nray, (i + 1) * nray)
        orc.umeyama(orc.statistics_p2l_fast(ident, sds, smask, sim["points"][sl], sim["normals"][sl], sim["hits"][sl], 1.0, 1))
    dt = time.perf_counter() - t0
    out["v1_batch_pose_corrections_per_s"] = round(len(v1poses) / dt, 1)
    out["v1_batch_rays_per_s"] = round(len(v1poses) * nray / dt, 1)
    out["v1_sample"] = ("one batch of 1000 poses x 16x900 rays, sphere-100k: raycast of all poses on %d threads (%.2f s) + per pose statistics_p2l (one pass) + Umeyama on "
                        "one thread (%.2f s); the reference's source comments record 5 464 corrections/s (Embree, its authors' desktop)" % (usable, t_sim, dt - t_sim))
    return out


def _pf_c4(ra, syn, T, np, ctx, hm, n_particles, n_beams, iters, bb=((-5, -5, -1), (5, 5, 1)), converged_at=None, particle_minor=False, variant=None):
    """config C4's sensor update; converged_at: a converged cloud ~ N(that pose, 0.25 m, 5 deg yaw) instead of the uniform one;
    particle_minor: the particle-coherent dealing (rmclhip_pf_set_mapping 1, 16 slots per workgroup, Morton order of x / y / yaw)"""
    if converged_at is None:
        poses, attrs = syn.uniform_particles(n_particles, seed=42, bb_min=bb[0] + (0, 0, -math.pi), bb_max=bb[1] + (0, 0, math.pi))
    else:
        poses, attrs = syn.converged_particles(n_particles, converged_at, 0.25, 5.0, seed=42)
    dirs = syn.model_directions(syn.model_pf16())
    sel = np.linspace(0, len(dirs) - 1, n_beams).astype(int)
    beams = ra.beams_from_points(dirs[sel] * np.float32(6.0))
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, T.identity())
    d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    if particle_minor:
        upd.set_mapping(1, 16, ra.DeviceArray.from_host(ctx, syn.morton_order_xy_yaw(poses)))
    if variant is not None:
        upd.set_variant(variant)
    upd.time_update(d_poses, d_attrs, n_particles, iters=1)
    ms = sorted(upd.time_update(d_poses, d_attrs, n_particles, iters=iters) for _ in range(5))[2]
    upd.close()
    return ms, n_particles * n_beams


def _pf_c4_unfused(ra, syn, T, np, ctx, hm, n_particles, n_beams):
    """config C4 in the reference's GPU schedule (one single-beam update per beam): (ms with a host wait after every beam, ms with one wait
    at the end) -- rmclhip_pf_time_update_unfused, host clock inside the library"""
    poses, attrs = syn.uniform_particles(n_particles, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    dirs = syn.model_directions(syn.model_pf16())
    sel = np.linspace(0, len(dirs) - 1, n_beams).astype(int)
    beams = ra.beams_from_points(dirs[sel] * np.float32(6.0))
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.init()
    upd.setInput(beams, T.identity())
    d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    a = upd.time_update_unfused(d_poses, d_attrs, n_particles, sync_each_beam=True, iters=2)
    b = upd.time_update_unfused(d_poses, d_attrs, n_particles, sync_each_beam=False, iters=2)
    upd.close()
    return a, b


def _median_call_ms(fn, reps=25, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t1 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t1)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def _pf_cycle(ra, syn, T, np, ctx, hm, n_particles):
    """one particle-filter cycle besides the sensor update: motion update (with the wall-collision ray),
    likelihood statistics and the gladiator tournament; median wall clock of the synchronous ABI calls."""
    poses, attrs = syn.uniform_particles(n_particles, seed=43, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    attrs["likelihood"]["mean"] = np.random.RandomState(1).uniform(0, 1, n_particles)
    attrs["likelihood"]["n_meas"] = 5000
    d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    d_pn, d_an = ra.DeviceArray(ctx, T.TRANSFORM, n_particles), ra.DeviceArray(ctx, T.PARTICLE_ATTRIBUTES, n_particles)
    out = {}
    step = T.transform_from_rpy((0.05, 0.0, 0.0), (0.0, 0.0, 0.01))
    for coll in (False, True):
        mo = ra.TFMotionUpdaterHip(hm, check_collision=coll)
        out["pf_motion_update%s_ms" % ("_collision" if coll else "")] = round(
            _median_call_ms(lambda: mo.update(d_p, d_a, n_particles, step, 0.001)), 4)
        mo.close()
    rs = ra.GladiatorResamplerHip(ctx)
    out["pf_resample_gladiator_ms"] = round(_median_call_ms(lambda: rs.update(d_p, d_a, d_pn, d_an, n_particles)), 4)
    out["pf_likelihood_stats_ms"] = round(_median_call_ms(lambda: rs.compute_stats(d_a, n_particles)), 4)
    rs.close()
    # the node's other resampler plugin (ResidualResamplerCPU.cpp:55-203): the reference's sequential loop as three parallel passes
    rr = ra.ResidualResamplerHip(ctx)
    out["pf_resample_residual_ms"] = round(_median_call_ms(lambda: rr.update(d_p, d_a, d_pn, d_an, n_particles), reps=9), 4)
    out["pf_resample_residual_loop_iterations"] = rr.last_draws
    rr.close()
    return out


if __name__ == "__main__":
    main()
