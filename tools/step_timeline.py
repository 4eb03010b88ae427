#!/usr/bin/env python
"""One step of the bench inside a rocprofv3 kernel trace (rocpd SQLite), kernel by kernel: start offset, duration, idle time in front of it.
The step = the last run of kernels that starts with the PFN scatter of the batch (the largest-grid pfn_scatter launch) -- `which` steps
from the end.  Sums: kernel time, idle time between kernels, span.

    python tools/step_timeline.py /tmp/prof1/.../bench_results.db [steps_from_end=2] [big|small]
"""
import sqlite3
import sys


def main(path, back=2, which="big"):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    pick = max if which == "big" else min                                  # "small": the one-frame graphs of the latency measurement
    big = pick(r[3] for r in rows if "pfn_scatter" in r[0] and "unscatter" not in r[0])
    heads = [i for i, r in enumerate(rows) if "pfn_scatter" in r[0] and "unscatter" not in r[0] and r[3] == big]
    whole = [k for k in range(len(heads) - 1) if any("table_heads" in r[0] or "rows_heads" in r[0] for r in rows[heads[k]:heads[k + 1]])]   # (the stage timings after the steps repeat single stages)
    i0, i1 = heads[whole[-back]], heads[whole[-back] + 1]
    step = rows[i0:i1]
    t0, prev_end, ksum, idle = step[0][1], step[0][1], 0, 0
    print(f"# source: {path}; step = kernels {i0}..{i1 - 1} ({len(step)} launches)")
    print(f"# {'start_us':>9} {'dur_us':>9} {'idle_before_us':>14}  kernel")
    for name, s, e, g in step:
        gap = max(0, s - prev_end)
        idle += gap
        ksum += e - s
        prev_end = max(prev_end, e)
        print(f"  {(s - t0) / 1e3:>9.1f} {(e - s) / 1e3:>9.1f} {gap / 1e3:>14.1f}  {name[:100]}")
    print(f"# span {(rows[i1][1] - t0) / 1e3:.1f} us to the next step's first kernel; kernels {ksum / 1e3:.1f} us; idle between kernels {idle / 1e3:.1f} us "
          f"(+ {(rows[i1][1] - prev_end) / 1e3:.1f} us before the next step)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2, sys.argv[3] if len(sys.argv) > 3 else "big")
