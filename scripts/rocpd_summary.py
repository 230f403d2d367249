#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel and per (kernel, grid) totals.

    python scripts/rocpd_summary.py gpurun_out/prof/xxx_results.db [--by-grid | --sequence]

--sequence: the dispatches of the LAST complete denoiser step (between two sample_step_k launches) in launch order, one line each.
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    c = sqlite3.connect(db)
    if "--sequence" in sys.argv:
        rows = c.execute("select name, grid_x, workgroup_x, start, end from kernels order by start").fetchall()
        cuts = [i for i, r in enumerate(rows) if "sample_step_k" in r[0]]
        lo, hi = (cuts[-2] + 1, cuts[-1] + 1) if len(cuts) >= 2 else (0, len(rows))
        t0 = rows[lo][3]
        print(f"dispatches {lo}..{hi - 1}: one denoiser step, {(rows[hi - 1][4] - t0) / 1e3:.1f} us wall, {sum(r[4] - r[3] for r in rows[lo:hi]) / 1e3:.1f} us of kernels")
        for name, gx, wx, st, en in rows[lo:hi]:
            short = name.split("(")[0].replace("void ", "")
            print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} us  grid {gx // max(wx, 1):6d}  {short}")
        return
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
