"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / avg / min / max duration (+ grid)."""
import sqlite3
import sys


def main(path, top=25):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, grid_x, workgroup_x, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
        "from kernels group by name, grid_x order by sum(end-start) desc").fetchall()
    tot = sum(r[7] for r in rows)
    print(f"{'kernel':88s} {'grid':>8s} {'wg':>5s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
    for r in rows[:top]:
        print(f"{r[0][:88]:88s} {r[1]:8d} {r[2]:5d} {r[3]:6d} {r[4] / 1e3:8.2f} {r[5] / 1e3:8.2f} {r[6] / 1e3:8.2f} "
              f"{100 * r[7] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
