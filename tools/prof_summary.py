"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite database (``*_results.db``): calls, average / total duration per (kernel, grid).
Round 6: the PERSISTENT kernels of the two-stage encode have one grid whatever their work (the listed cells are a device-side count), so a
(kernel, grid) row mixes the batch-of-32 launches with the one-frame ones of the latency loop: `long_calls / long_avg_us` are the launches
longer than half the row's longest -- the bench's own batch -- and that is the figure bench.py's `roofline.avg_launch_us` must agree with.
``python tools/prof_summary.py <db> [top]``"""
import sqlite3
import sys


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e3, grid_x, grid_y from kernels group by name, grid_x, grid_y order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'share':>6s}  {'grid':18s} {'long_calls':>10s} {'long_avg_us':>11s}")
    for r in rows[:top]:
        name = r[0].replace("qv2x::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        d = [x[0] / 1e3 for x in c.execute("select end-start from kernels where name = ? and grid_x = ? and grid_y = ?", (r[0], r[4], r[5]))]
        long_ = [x for x in d if x > max(d) / 2]
        print(f"{name[:72]:72s} {r[1]:6d} {r[2]:9.1f} {r[3] / tot * 100:5.1f}%  {'(%d,%d)' % (r[4], r[5]):18s} {len(long_):10d} {sum(long_) / len(long_):11.1f}")


if __name__ == "__main__":
    main()
