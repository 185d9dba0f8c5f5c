#!/usr/bin/env python3
"""Per-hardware-queue timeline of ONE captured step from a rocprofv3 kernel trace (.db): which kernels ran where and when, and how
much the queues overlapped (the step's branches run on separate HIP streams; hipGraph maps them to queues). Usage: stream_timeline.py db [step]
[--grep substring] [--dump t0 t1]: also list every kernel of the step whose name contains the substring (start, end, queue, kernels of OTHER queues
that ran during it)."""
import collections, re, sqlite3, sys
argv = [a for a in sys.argv[1:]]
grep = None
if "--grep" in argv:
    i = argv.index("--grep"); grep = argv[i + 1].lower(); del argv[i:i + 2]
sys.argv = [sys.argv[0]] + argv
dump = None
if "--dump" in argv:   # --dump t0_ms t1_ms: every kernel of the step starting in that window (start, duration, gap to the previous kernel of its queue)
    i = argv.index("--dump"); dump = (float(argv[i + 1]), float(argv[i + 2])); del argv[i:i + 3]
    sys.argv = [sys.argv[0]] + argv
names = "--names" in argv
if names:
    argv.remove("--names"); sys.argv = [sys.argv[0]] + argv
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "seed_advance" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 6
step = rows[starts[k]:starts[k + 1]]
t0 = step[0][1]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "")))[:44]
print("step %d: %.2f ms, %d kernels" % (k, (step[-1][2] - t0) / 1e6, len(step)))
segs, prevq = [], None
for n, s, e, q in step:
    if q != prevq:
        segs.append([s, e, q, 0, collections.Counter()])
        prevq = q
    segs[-1][1] = e; segs[-1][3] += 1; segs[-1][4][short(n)] += 1
for s, e, q, c, names in segs:
    if c >= 6 or e - s > 80e3:
        print("  %6.2f-%6.2f ms q%d %4d kernels  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, c, ", ".join("%s x%d" % kv for kv in names.most_common(3))))
byq = collections.defaultdict(list)
for n, s, e, q in step:
    byq[q].append((s, e))
ev = sorted([(s, 1) for l in byq.values() for s, e in l] + [(e, -1) for l in byq.values() for s, e in l])
both = act = 0; last = ev[0][0]; busy = collections.Counter()
for t, d in ev:
    busy[min(act, 2)] += t - last; last = t; act += d
print("  time with 0 / 1 / >=2 kernels in flight: %.2f / %.2f / %.2f ms" % (busy[0] / 1e6, busy[1] / 1e6, busy[2] / 1e6))
if grep:
    hits = [(n, s, e, q) for n, s, e, q in step if grep in n.lower()]
    print("  %d kernels matching %r:" % (len(hits), grep))
    for n, s, e, q in hits:
        others = [short(n2) for n2, s2, e2, q2 in step if q2 != q and s2 < e and e2 > s]
        print("    %7.3f-%7.3f ms q%d %-40s beside %d kernels of other queues%s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, short(n), len(others), (": " + ", ".join(sorted(set(others))[:3])) if others else ""))
if names or dump:   # launches and summed duration per kernel name in this one step
    cnt = collections.defaultdict(lambda: [0, 0.0])
    for n, s_, e, q in step:
        c = cnt[short(n)]; c[0] += 1; c[1] += (e - s_) / 1e3
    print("  per kernel name in this step (launches, summed us):")
    for n, (c, us) in sorted(cnt.items(), key=lambda kv: -kv[1][1]):
        print("    %4d  %8.1f us  %s" % (c, us, n))
if dump:
    last_end = {}
    print("  kernels starting in %.2f .. %.2f ms:" % dump)
    for n, s, e, q in step:
        t = (s - t0) / 1e6
        if dump[0] <= t < dump[1]:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            print("    %7.3f ms q%d dur %7.1f us gap %6.1f us  %s" % (t, q, (e - s) / 1e3, gap, short(n)))
        last_end[q] = e
