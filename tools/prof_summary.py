"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite database (``*_results.db``): calls, average / total duration per (kernel, grid).
``python tools/prof_summary.py <db> [top]``"""
import sqlite3
import sys


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e3, grid_x, grid_y from kernels group by name, grid_x, grid_y order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'share':>6s}  grid")
    for r in rows[:top]:
        name = r[0].replace("qv2x::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{name[:72]:72s} {r[1]:6d} {r[2]:9.1f} {r[3] / tot * 100:5.1f}%  ({r[4]},{r[5]})")


if __name__ == "__main__":
    main()
