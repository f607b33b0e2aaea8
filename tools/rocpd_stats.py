"""Kernel statistics out of a rocprofv3 rocpd database (ROCm 7.2's default output: <name>_results.db; older calls got CSV files).
usage: rocpd_stats.py results.db [first_marker_substring [pass_index]]
  Prints calls / total / average / share per kernel name (template arguments kept, parameter lists cut).  With a marker (a substring
  of a kernel that starts every pass, e.g. logmel_pass1) only dispatches from the pass_index-th occurrence (default: the last) on are
  counted, so warm-up passes are left out."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n: str) -> str:
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((anonymous namespace)\)::", "", n)
    depth, out = 0, []
    for ch in n:                       # cut the parameter list: the first '(' at template depth 0
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    lo, hi = 0, len(rows)
    if len(sys.argv) > 2:
        marks = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
        k = int(sys.argv[3]) if len(sys.argv) > 3 else -1
        lo = marks[k]
        hi = marks[k + 1] if (k + 1 < len(marks) and k != -1) else len(rows)
    rows = rows[lo:hi]
    agg = defaultdict(lambda: [0, 0.0])
    for n, s, e in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s) / 1e3
    total = sum(a[1] for a in agg.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"{len(rows)} dispatches, kernel time {total / 1e3:.3f} ms, span {span / 1e3:.3f} ms (gaps {100 * (1 - total / span):.1f} %)")
    print(f"{'calls':>6} {'total us':>10} {'avg us':>9} {'%':>6}  kernel")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:6d} {t:10.1f} {t / c:9.2f} {100 * t / total:6.2f}  {n}")


if __name__ == "__main__":
    main()
