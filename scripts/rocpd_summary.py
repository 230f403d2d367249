#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel and per (kernel, grid) totals.

    python scripts/rocpd_summary.py gpurun_out/prof/xxx_results.db [--by-grid]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, workgroup_x, duration, vgpr_count, accum_vgpr_count, lds_size "
                     "from kernels").fetchall()
    total = sum(r[4] for r in rows)
    agg = {}
    for name, gx, gy, wx, dur, vg, ag, lds in rows:
        short = name.split("(")[0].replace("void ", "")
        key = (short, gx // max(wx, 1), gy) if by_grid else (short,)
        a = agg.setdefault(key, [0, 0.0, vg, ag, lds])
        a[0] += 1
        a[1] += dur
    print(f"total kernel time {total/1e6:.3f} ms over {len(rows)} dispatches")
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}  vgpr agpr lds")
    for key, (n, dur, vg, ag, lds) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        label = key[0] + (f" grid=({key[1]},{key[2]})" if by_grid else "")
        print(f"{label:60s} {n:6d} {dur/1e6:10.3f} {dur/n/1e3:10.1f} {100*dur/total:6.2f}  {vg} {ag} {lds}")


if __name__ == "__main__":
    main()
