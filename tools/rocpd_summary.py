#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output format of ROCm 7.2's
`rocprofv3 --kernel-trace --stats` / `--pmc`) as text: per kernel AND launch geometry
(probe and re-rank are the same template with different grids), calls / total / avg / min /
max duration, and the per-dispatch average of any PMC counters collected.

    python tools/rocpd_summary.py <results.db> [--match pqv] > profiles/xxx.txt
"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="", help="only kernels whose name contains this")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
                      "accum_vgpr_count, sgpr_count, duration from kernels").fetchall()
    groups = defaultdict(list)
    meta = {}
    for name, gx, gy, gz, wx, lds, vg, ag, sg, dur in rows:
        if args.match and args.match not in name:
            continue
        key = (short(name), gx, gy, gz)
        groups[key].append(dur)
        meta[key] = (wx, lds, vg, ag, sg)
    total = sum(sum(v) for v in groups.values()) or 1
    print(f"# rocprofv3 kernel trace summary of {args.db}")
    print(f"# {'kernel':<92} {'grid(threads)':>24} {'calls':>6} {'total_us':>12} {'avg_us':>11} "
          f"{'min_us':>11} {'max_us':>11} {'pct':>6}  wg lds vgpr agpr sgpr")
    for key, durs in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        name, gx, gy, gz = key
        wx, lds, vg, ag, sg = meta[key]
        t = sum(durs)
        print(f"{name:<94} {f'{gx}x{gy}x{gz}':>24} {len(durs):>6} {t/1e3:>12.3f} {t/len(durs)/1e3:>11.3f} "
              f"{min(durs)/1e3:>11.3f} {max(durs)/1e3:>11.3f} {100*t/total:>6.2f}  {wx} {lds} {vg} {ag} {sg}")
    try:
        pmc = db.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, value, "
                         "duration from counters_collection").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        acc = defaultdict(lambda: [0, 0.0, 0.0])
        for name, gx, gy, gz, cname, val, dur in pmc:
            if args.match and args.match not in name:
                continue
            a = acc[(short(name), gx, gy, gz, cname)]
            a[0] += 1; a[1] += val; a[2] += dur
        print("\n# PMC counters: per-dispatch averages (FETCH_SIZE / WRITE_SIZE are in KiB as reported; "
              "see MI355X_MICROARCH.md #HBM for the gfx950 calibration caveat)")
        print(f"# {'kernel':<92} {'grid(threads)':>24} {'counter':>12} {'dispatches':>10} {'avg_value':>16} {'avg_us':>11}")
        for (name, gx, gy, gz, cname), (cnt, val, dur) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
            print(f"{name:<94} {f'{gx}x{gy}x{gz}':>24} {cname:>12} {cnt:>10} {val/cnt:>16.2f} {dur/cnt/1e3:>11.3f}")


if __name__ == "__main__":
    main()
