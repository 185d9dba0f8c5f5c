#!/usr/bin/env python3
"""For every dispatch of a replayed step whose kernel name contains NAME (and whose grid_x equals GRID, if given): when it ran, for
how long, on which hardware queue, what ran just before it on that queue, and which kernels of other queues overlapped it.
usage: kernel_neighbours.py trace.db NAME [GRID] [step]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
name = sys.argv[2].lower()
grid = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
rows = list(db.execute("select name, start, end, queue_id, grid_x from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "seed_advance" in r[0]]
k = int(sys.argv[4]) if len(sys.argv) > 4 else len(starts) - 6
step = rows[starts[k]:starts[k + 1]]
t0 = step[0][1]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "")))[:48]
for i, (n, s, e, q, g) in enumerate(step):
    if name not in n.lower() or (grid is not None and g != grid):
        continue
    prev = [r for r in step[:i] if r[3] == q]
    p = prev[-1] if prev else None
    others = [short(r[0]) + " (%.0f us)" % ((r[2] - r[1]) / 1e3) for r in step if r[3] != q and r[1] < e and r[2] > s]
    print("%8.3f ms +%6.1f us q%d grid=%d %s | previous on queue: %s ended %.1f us earlier | beside: %s" % (
        (s - t0) / 1e6, (e - s) / 1e3, q, g, short(n), short(p[0]) if p else "-", (s - p[2]) / 1e3 if p else 0.0, ", ".join(others[:4]) or "-"))
