#!/usr/bin/env python
"""profiles/r05_pmc_large_maps.txt + the large-map entries of profiles/traffic.json from the passes of tools/pmc_large_maps.sh
(kernel durations from the --stats pass, FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS from the PMC passes, one counter set per run).

FETCH_SIZE is reported as measured (requests x 64 B) AND doubled: tools/ubench/gather_calib.hip (profiles/r05_fetch_size_calibration.txt)
shows one request per touched 128-B line, tallied at 64 B -- exact for 64-B records (quantised nodes, one triangle record), half the
bytes for 128-B records (full-precision nodes, two adjacent records).  A BVH walk mixes both, so the truth lies between the two.

  python tools/large_maps_traffic.py gpurun_out/pmc_large_r05 profiles
"""
import json
import os
import re
import sys


def stat(path, kernel):
    """(calls, avg_ns) of the first kernel line containing `kernel`"""
    for line in open(path):
        if line.startswith(kernel):
            m = re.match(r".*?\)\s+(\d+)\s+([\d.]+)\s", line)
            if m:
                return int(m.group(1)), float(m.group(2))
    return None


def pmc(path, kernel, counter):
    on = False
    for line in open(path):
        if line.startswith("# PMC"):
            on = True
            continue
        if on and line.startswith(kernel) and (" " + counter + " ") in line:
            return float(line.split(counter)[1].split()[0])
    return None


def main(d, out_dir):
    rows, traffic = [], {}
    for what, kernel, rays, nposes in (("find_rot", "k_find<1u, 23", 131072, 1), ("v1", "k_find<1u, 24", 14400000, 1000)):
        for nf in (100000, 1000000, 10000000):
            p = lambda tag: os.path.join(d, "%s_%d_%s.txt" % (what, nf, tag))
            if not os.path.exists(p("stats")):
                continue
            calls, avg_ns = stat(p("stats"), kernel)
            f = pmc(p("fetch"), kernel, "FETCH_SIZE") * 1024.0
            w = pmc(p("write"), kernel, "WRITE_SIZE") * 1024.0
            hit, miss = pmc(p("l2"), kernel, "TCC_HIT_sum"), pmc(p("l2"), kernel, "TCC_MISS_sum")
            alg = rays * 33 + nf * 36 + (2 * nf - 1) * 32 + 32 * nposes
            key = "%s_sphere%s" % ("k_find_kind23_c2_scan_16_poses_in_turn" if what == "find_rot" else "k_find_kind24_v1_batch_1000x16x900",
                                   {100000: "100k", 1000000: "1m", 10000000: "10m"}[nf])
            traffic[key] = {"measured_in": "round 5 (tools/pmc_large_maps.sh)", "kernel_avg_us": round(avg_ns / 1e3, 2), "dispatches": calls,
                            "fetch_bytes_measured": round(f), "write_bytes_measured": round(w),
                            "hbm_bytes_per_launch": round(f + w), "hbm_bytes_per_launch_upper_bound_x2_fetch": round(2 * f + w),
                            "algorithmic_bytes": alg, "l2_hit_rate": round(hit / (hit + miss), 3)}
            rows.append((what, nf, avg_ns / 1e3, rays / (avg_ns * 1e-9), f / 1e6, w / 1e6, (f + w) / (avg_ns * 1e-9) / 1e12,
                         (2 * f + w) / (avg_ns * 1e-9) / 1e12, alg / 1e6, (f + w) / alg, hit / (hit + miss), (f + w) / rays))
    with open(os.path.join(out_dir, "r05_pmc_large_maps.txt"), "w") as fh:
        fh.write(__doc__.split("\n\n")[1] + "\n\n")
        fh.write("%-9s %9s %10s %10s %10s %9s %12s %12s %10s %9s %7s %9s\n" % (
            "workload", "faces", "kernel_us", "Grays/s", "fetch_MB", "write_MB", "TB/s(x1)", "TB/s(x2 f)", "algo_MB", "traf/algo", "L2hit", "B/ray(x1)"))
        for r in rows:
            fh.write("%-9s %9d %10.2f %10.3f %10.1f %9.1f %12.2f %12.2f %10.1f %9.2f %7.2f %9.0f\n" % (
                r[0], r[1], r[2], r[3] / 1e9, r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]))
        fh.write("\nfind_rot = one C2 scan (128 x 1024 rays, kind 23: 128-B nodes) per launch, 16 poses in turn (336 launches); v1 = the find of the reference's\n"
                 "benchmark batch, 1000 poses x 16 x 900 rays in one launch (kind 24: 64-B quantised nodes).  algo_MB = SURVEY 8(d)'s contract figure\n"
                 "(outputs + the WHOLE map once per launch); a BVH walk touches a vanishing part of a large map, hence traf/algo << 1 for one scan.\n"
                 "HBM peak 8 TB/s (spec), ~6.3 TB/s achievable (MI355X_MICROARCH.md).\n")
    tj = os.path.join(out_dir, "traffic.json")
    old = json.load(open(tj)) if os.path.exists(tj) else {}
    old.update(traffic)
    json.dump(old, open(tj, "w"), indent=1, sort_keys=True)
    print(open(os.path.join(out_dir, "r05_pmc_large_maps.txt")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
