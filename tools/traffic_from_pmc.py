#!/usr/bin/env python
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh.

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch (rocprofv3, derived from TCC_EA0_RDREQ / _WRREQ).  On gfx950
FETCH_SIZE reads exactly 1/2 of the bytes of a WIDE coalesced streaming read (16 B/lane) -- that correction
(x2) is applied to the streaming reduction kernel only; BVH traversal reads are 16-B gathers / scalar 64-B
requests for which the counter is uncalibrated, so they are reported as measured, with the x2 bound alongside.
"""
import json
import sqlite3
import sys


def avg(db, kernel_like, counter):
    con = sqlite3.connect(db)
    row = con.execute(
        "select avg(v), count(*) from (select sum(p.counter_value) as v from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
        "where k.name like ? and p.counter_name = ? group by p.dispatch_id)", (kernel_like, counter)).fetchone()
    return (row[0] or 0.0), row[1]


def main(d, out):
    res = {}
    # key = the kernel as bench.py names it (traversal kind of k_find), like = its demangled template arguments
    # (round 6: bench.py's default run autotunes -- the autotune's own trial launches of other kinds land in the same pass and are told
    # apart by the kernel name; the rule's kind 23 is profiled with --no-autotune)
    specs = (("k_find_kind32", "%k_find<1u, 32%", "find_v15", False), ("k_find_kind23", "%k_find<1u, 23%", "find_v15rule", False),
             ("k_find_kind2", "%k_find<1u, 2,%", "find_v2", False), ("k_pf_update_v3_c5_shard_sphere1m", "%k_pf_update_v3%", "pfc5", False),
             ("k_pf_update_v3", "%k_pf_update_v3%", "pf", False),
             ("k_micp_moments", "%k_micp_moments%", "red", True), ("k_reduce_partials", "%k_reduce_partials%", "red", True))
    for key, like, pre, wide in specs:
        try:
            f, nf = avg("%s/%s_fetch_results.db" % (d, pre), like, "FETCH_SIZE")
            w, nw = avg("%s/%s_write_results.db" % (d, pre), like, "WRITE_SIZE")
        except Exception as e:  # pass missing
            print("skip", key, e)
            continue
        if nf == 0 and nw == 0:
            continue
        fb, wb = f * 1024.0, w * 1024.0
        res[key] = {"fetch_bytes_measured": round(fb), "write_bytes_measured": round(wb),
                    "fetch_correction": 2.0 if wide else 1.0,
                    "hbm_bytes_per_launch": round(fb * (2.0 if wide else 1.0) + wb),
                    "hbm_bytes_per_launch_upper_bound_x2_fetch": round(2 * fb + wb), "dispatches": [nf, nw]}
    try:   # keys that were not re-measured in this run keep their previous entry
        old = json.load(open(out))
    except Exception:
        old = {}
    old.update(res)
    res = old
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
