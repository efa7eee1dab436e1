#!/usr/bin/env python
"""bench.py -- headline benchmark of the RMCL / MICP-L hot path on MI355X.

Metric (BASELINE.json): ray-mesh intersections/s (+ derived pose-corrections/s), 100k-triangle mesh,
128x1024 spherical scan.  Workload at every N: config C2 = one pose x 131 072 rays against the
sphere-100k mesh per step, i.e. one RCC find() launch (rmcl/src/rmcl/registration/RCCEmbree.cpp:26-36)
with all five output attributes written.  A single MICP pose does not shard:
--gpus N runs N independent replicas, one process per GPU, different pose per rank, no data-path
collective ("replicas only", weak scaling).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the stream the kernel
runs on (rmclhip_rcc_time_find); `cpu_baseline` times the CPU oracle (a port -- the reference's Embree
path cannot be built here) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_raycast(n_rays, n_tri, n_poses):
    """SURVEY.md 8(d) B_rc: outputs 33 B/ray (hits 1 + ranges 4 + points 12 + normals 12 + face ids 4)
    + 36 B/triangle + 32 B per BVH2 node of the reference tree + 32 B/pose."""
    return n_rays * 33 + n_tri * 36 + (2 * n_tri - 1) * 32 + n_poses * 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--variant", type=int, default=1, help="find traversal: 1 per-lane while-while (default), 0 wave-packet")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time C3 (MICP loop), batch and C4 (particle filter)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rmcl_amd as ra
    from rmcl_amd import synthetic as syn, types as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    ctx = ra.Context(local_rank)
    v, f = syn.uv_sphere(100000)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c2()
    n_rays = model.phi.size * model.theta.size
    rcc = ra.RCCHipSpherical(hm)
    rcc.set_variant(args.variant)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    # replicas: rank r localises a scan from a different pose inside the same map
    Tbm = T.mult(syn.pose_c2_truth(), T.transform_from_rpy((0.05 * rank, -0.03 * rank, 0.0), (0.0, 0.0, 0.11 * rank)))

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(max(args.warmup, 1)):
        rcc.find_async(Tbm)
    rcc.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rcc.find_async(Tbm)
    rcc.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant kernel, measured live: HIP events on the rcc's own stream around back-to-back launches
    kernel_ms = rcc.time_find(Tbm, iters=max(50, min(args.steps, 500)))
    b_rc = algorithmic_bytes_raycast(n_rays, len(f), 1)
    achieved = b_rc / (kernel_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "k_find<spherical,%s>" % ("packet" if args.variant == 0 else "lane"),
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": b_rc, "kernel_ms": round(kernel_ms, 5),
                "kernel_rays_per_s": round(n_rays / (kernel_ms * 1e-3), 1)}

    extras = {}
    if rank == 0:
        # C3: MICP-L inner loop, schedule (R) 1 find + 10 x (reduce + solve) and (B) 10 x (find + reduce + solve)
        # measured scan = the product's own simulation at the ground-truth pose (no oracle involved)
        rcc.find(syn.pose_c2_truth())
        rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
        rcc.params.max_dist = 1.0
        rcc.adaptive_max_dist_min = 0.15
        est = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation())
        for name, refind in (("R", False), ("B", True)):
            rcc.correct_once(est, T.identity(), 10, 0.0, refind)
            reps = 20
            t1 = time.perf_counter()
            for _ in range(reps):
                rcc.correct_once(est, T.identity(), 10, 0.0, refind)
            dt = (time.perf_counter() - t1) / reps
            extras["c3_schedule_%s_ms" % name] = round(dt * 1e3, 4)
            extras["c3_schedule_%s_pose_corrections_per_s" % name] = round(1.0 / dt, 1)
            extras["c3_schedule_%s_icp_iterations_per_s" % name] = round(10.0 / dt, 1)
        red_ms = rcc.time_reduce(T.identity(), iters=100)
        extras["reduce_kernel_ms"] = round(red_ms, 5)
        extras["reduce_GBps"] = round((n_rays * 38 + 64) / (red_ms * 1e-3) / 1e9, 1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc  # cpu_baseline leg only: the oracle is the thing timed, never the product path
        m = orc.Mesh(v, f)
        cores = os.cpu_count() or 1
        m.simulate_spherical(model, T.identity(), Tbm, bvh=True, nthreads=cores, want=("hits", "ranges", "points", "normals", "face_ids"))
        reps, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 10.0 and reps < 200:
            m.simulate_spherical(model, T.identity(), Tbm, bvh=True, nthreads=cores)
            reps += 1
        dt = time.perf_counter() - t1
        cpu = {"value": round(reps * n_rays / dt, 1), "unit": "rays/s", "cores": cores, "kind": "port",
               "sample": "%d x the same 128x1024 / 100k-triangle scan, CPU oracle (BVH2 + same intersector), %d threads"
                         % (reps, cores)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "ray-mesh intersections/s (128x1024 scan, 100k-tri mesh)",
            "value": round(world * args.steps * n_rays / elapsed, 1),
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: 1 pose x 128x1024 spherical LiDAR, UV-sphere 100k triangles, find() only, "
                                   "5 output attributes; N>1 = independent replicas",
                       "rays_per_step": n_rays, "triangles": int(len(f)), "parallelism": "replicas%d" % world,
                       "kernel_variant": args.variant},
            "pose_corrections_per_s": extras.get("c3_schedule_R_pose_corrections_per_s"),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extras": extras,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
