#!/usr/bin/env python
"""The reference's own 1 M- and 10 M-face benchmark rows (lidar_corrector_{optix,embree}_benchmark.cpp:144-152, :161-169) on the HIP path:
C2 find (128x1024), the v1 batch 1000 x 16x900, C3 schedule (R), on UV spheres of 100 k / 1 M / 10 M faces -- the first maps that leave
the 256 MB MALL.  Parity of every map against a brute-force sample of the scan's own rays (no BVH), then timings, then the automatic
rule against a measurement (rmclhip_rcc_autotune).

  python tools/large_maps.py [--faces 100000,1000000,10000000] [--sample 256] [--out gpurun_out/large_maps.json]
  python tools/large_maps.py --faces 10000000 --only find        (one kernel loop: the PMC passes of tools/pmc_large_maps.sh)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def scan_poses(T, syn, n):
    """n sensor poses well inside the radius-10 sphere: the same scan from different places touches different triangles, so a loop
    over them cannot live on what the previous launch left in the MALL"""
    import numpy as np
    rng = np.random.RandomState(7)
    out = [syn.pose_c2_truth()]
    for _ in range(n - 1):
        out.append(T.transform_from_rpy(tuple(rng.uniform(-3.0, 3.0, 3)), tuple(rng.uniform(-0.4, 0.4, 2)) + (rng.uniform(-3.1, 3.1),)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", default="100000,1000000,10000000")
    ap.add_argument("--sample", type=int, default=256, help="rays of the scan checked against EVERY triangle")
    ap.add_argument("--only", choices=("all", "find", "find_rot", "v1", "c3"), default="all")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import rmcl_amd as ra
    from rmcl_amd import synthetic as syn, types as T

    ctx = ra.Context(0)
    res = {"device": ctx.device_name(), "maps": {}}
    for nf in [int(x) for x in args.faces.split(",")]:
        r = {}
        t0 = time.perf_counter()
        v, f = syn.uv_sphere(nf)
        r["mesh_gen_s"] = round(time.perf_counter() - t0, 3)
        t0 = time.perf_counter()
        hm = ra.import_hip_map(ctx, v, f)
        r["map_build_s"] = round(time.perf_counter() - t0, 3)
        info = hm.info()
        r["map"] = {k: info[k] for k in ("n_faces", "n_nodes", "max_depth", "stack_need", "device_bytes", "height_fallbacks", "guarded_nodes")}
        model = syn.model_c2()
        n_rays = model.phi.size * model.theta.size
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        Tbm = syn.pose_c2_truth()
        poses = scan_poses(T, syn, 16)
        b_alg = n_rays * 33 + nf * 36 + (2 * nf - 1) * 32 + 32
        r["algorithmic_bytes_per_scan"] = b_alg

        if not args.no_parity and args.only == "all":
            import oracle as orc
            t0 = time.perf_counter()
            m = orc.Mesh(v, f)
            r["oracle_mesh_s"] = round(time.perf_counter() - t0, 2)
            rcc.find(Tbm)
            gpu = rcc.modelView()
            dirs = orc.spherical_directions(model)
            idx = np.sort(np.random.RandomState(99).choice(len(dirs), size=args.sample, replace=False))
            t0 = time.perf_counter()
            sub = m.simulate_o1dn(len(idx), 1, model.range.min, model.range.max, (0.0, 0.0, 0.0), dirs[idx], T.identity(), Tbm, bvh=False,
                                  nthreads=16, want=("hits", "ranges", "face_ids"))
            r["brute_force_s"] = round(time.perf_counter() - t0, 2)
            r["parity_brute_force_sample"] = {"rays": len(idx), "hits_equal": bool(np.array_equal(gpu["hits"][idx], sub["hits"])),
                                              "face_ids_equal": bool(np.array_equal(gpu["face_ids"][idx], sub["face_ids"])),
                                              "ranges_max_rel": float(np.max(np.abs(gpu["ranges"][idx] - sub["ranges"]) / sub["ranges"]))}
            if nf <= 1000000:
                t0 = time.perf_counter()
                ref = m.simulate_spherical(model, T.identity(), Tbm, bvh=2, nthreads=16, want=("hits", "ranges", "face_ids"))
                r["oracle_bvh_all_rays_s"] = round(time.perf_counter() - t0, 2)
                r["parity_oracle_bvh_all_rays"] = {"hits_equal": bool(np.array_equal(gpu["hits"], ref["hits"])),
                                                   "face_ids_equal": bool(np.array_equal(gpu["face_ids"], ref["face_ids"])),
                                                   "ranges_bit_equal": bool(np.array_equal(gpu["ranges"], ref["ranges"]))}
            rr = np.linalg.norm(gpu["points"].astype(np.float64), axis=1)   # sensor frame; |R p + t| needs the pose:
            from scipy.spatial.transform import Rotation
            R = Rotation.from_quat([Tbm["R"]["x"], Tbm["R"]["y"], Tbm["R"]["z"], Tbm["R"]["w"]]).as_matrix()
            t = np.array([Tbm["t"]["x"], Tbm["t"]["y"], Tbm["t"]["z"]], np.float64)
            rad = np.linalg.norm(gpu["points"].astype(np.float64) @ R.T + t, axis=1)
            r["hit_on_radius"] = {"all_hit": bool(gpu["hits"].all()), "r_min": float(rad.min()), "r_max": float(rad.max())}
            del m, rr

        def med(fn, k=5):
            ts = sorted(fn() for _ in range(k))
            return ts[len(ts) // 2]

        if args.only in ("all", "find"):
            r["rule_kind"] = rcc.find_variant(1)
            r["find_same_pose_ms"] = round(med(lambda: rcc.time_find(Tbm, iters=20)), 5)
            r["find_same_pose_rays_per_s"] = round(n_rays / (r["find_same_pose_ms"] * 1e-3), 1)
        if args.only in ("all", "find_rot"):
            # 16 different poses in turn (wall clock around 4 x 16 asynchronous finds: the launches queue behind each other)
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
            r["find_rotating_poses_ms"] = round(med(rot), 5)
            r["find_rotating_poses_rays_per_s"] = round(n_rays / (r["find_rotating_poses_ms"] * 1e-3), 1)
            r["find_rotating_poses_algorithmic_GBps"] = round(b_alg / (r["find_rotating_poses_ms"] * 1e-3) / 1e9, 1)
        if args.only == "all":
            kinds = {}
            for kind in (23, 24, 2, 0):
                rcc.set_traversal(kind)
                kinds[str(kind)] = round(med(lambda: rcc.time_find(Tbm, iters=10), 3), 5)
            rcc.set_traversal(15)
            r["find_same_pose_ms_by_kind"] = kinds
            k_t, ms_t = rcc.autotune(Tbm)
            r["autotune"] = {"chosen_kind": k_t, "ms": round(ms_t, 5)}
            rcc.setModel(model)
        if args.only in ("all", "c3"):
            rcc.find(Tbm)
            rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
            rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
            est = T.mult(Tbm, syn.pose_c2_perturbation())
            rcc.correct_once(est, T.identity(), 10, 0.0, False)
            ms = rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=30)
            r["c3_schedule_R_ms"] = round(ms, 4)
            r["c3_schedule_R_pose_corrections_per_s"] = round(1e3 / ms, 1)
            msb = rcc.time_correct_once(est, T.identity(), 10, 0.0, True, iters=10)
            r["c3_schedule_B_ms"] = round(msb, 4)
        rcc.close()
        if args.only in ("all", "v1"):
            small = ra.RCCHipSpherical(hm)
            small.setTsb(T.identity())
            small.setModel(syn.model_vlp16_900())
            rng = np.random.RandomState(1)
            v1poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng.uniform(-3, 3)))
                                for _ in range(1000)], dtype=T.TRANSFORM)
            small.find(T.identity())
            small.set_dataset_from_ranges(small.modelView()["ranges"])
            small.params.max_dist = 1.0
            small.correct_batch(v1poses)

            def v1():
                t1 = time.perf_counter()
                small.correct_batch(v1poses)
                return (time.perf_counter() - t1) * 1e3
            dt = med(v1)
            r["v1_bench_1000x16x900_ms"] = round(dt, 4)
            r["v1_bench_rays_per_s"] = round(1000 * 16 * 900 / (dt * 1e-3), 1)
            r["v1_bench_pose_corrections_per_s"] = round(1e6 / dt, 1)
            r["v1_bench_find_batch_ms"] = round(small.time_find_batch(v1poses, iters=3), 4)
            r["v1_batch_rule_kind"] = small.find_variant(1000)
            if args.only == "all":
                k_b, ms_b = small.autotune_batch(v1poses)
                r["v1_batch_autotune"] = {"chosen_kind": k_b, "ms": round(ms_b, 4)}
                same = np.array([T.transform_from_rpy((0.0, 0.0, 0.2), (0, 0, 0))] * 1000, dtype=T.TRANSFORM)
                small.setModel(syn.model_vlp16_900())
                small.correct_batch(same)
                t1 = time.perf_counter()
                small.correct_batch(same)
                r["v1_bench_identical_poses_ms"] = round((time.perf_counter() - t1) * 1e3, 4)
            small.close()
        r["reference_source_comments"] = {
            100000: {"optix_rays_per_s": 1.06e9, "optix_corrections_per_s": 73.7e3, "embree_rays_per_s": 78.7e6},
            1000000: {"optix_rays_per_s": 852e6, "optix_corrections_per_s": 59.2e3, "embree_rays_per_s": 71.6e6},
            10000000: {"optix_rays_per_s": 462e6, "optix_corrections_per_s": 32.1e3, "embree_rays_per_s": 31.6e6}}.get(nf)
        res["maps"][str(nf)] = r
        print(json.dumps({str(nf): r}), flush=True)
        hm.release()
        del v, f
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
