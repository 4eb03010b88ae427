#!/usr/bin/env python
"""Per-kernel statistics (the `rocprofv3 --stats` kernel table) from a rocprofv3 rocpd SQLite result file.

    python tools/rocpd_stats.py gpurun_out/prof_r1/bench_results.db > profiles/r01_bench_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
               max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size)
        from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# {'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7}")
    for (name, calls, tot, avg, mn, mx, vg, ag, sg, lds) in rows[:top]:
        short = name if len(name) <= 90 else name[:87] + "..."
        print(f"  {short:<90} {calls:>7} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100.0 * tot / total:>6.2f} {vg or 0:>5} {ag or 0:>5} {sg or 0:>5} {lds or 0:>7}")


if __name__ == "__main__":
    main(sys.argv[1])
