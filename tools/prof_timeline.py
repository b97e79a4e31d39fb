"""Timeline of one decode token from a rocprofv3 rocpd database: every dispatch between two consecutive embed_kernel
launches of the steady state, with its duration and the gap to its predecessor; then per-kernel averages of both."""
import sqlite3
import sys
from collections import defaultdict


def main(path, which=-2):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    emb = [i for i, r in enumerate(rows) if "embed_kernel" in r[0]]
    a, b = emb[which - 1], emb[which]
    tok = rows[a:b]
    print("token span %.1f us, %d dispatches" % ((tok[-1][2] - tok[0][1]) / 1e3, len(tok)))
    dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
    for i, r in enumerate(tok):
        key = (r[0].split("(")[0][-40:], r[3])
        dur[key] += (r[2] - r[1]) / 1e3
        if i:
            gap[key] += (r[1] - tok[i - 1][2]) / 1e3
        n[key] += 1
    print("%-44s %9s %5s %9s %9s" % ("kernel", "grid", "n", "dur us", "gap before"))
    td = tg = 0.0
    for k in dur:
        print("%-44s %9d %5d %9.2f %9.2f" % (k[0], k[1], n[k], dur[k] / n[k], gap[k] / n[k]))
        td += dur[k]
        tg += gap[k]
    print("sum of durations %.1f us, sum of gaps %.1f us" % (td, tg))
    for r in tok[:12]:
        print("  %-40s start +%8.2f dur %6.2f" % (r[0].split("(")[0][-40:], (r[1] - tok[0][1]) / 1e3, (r[2] - r[1]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
