import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id, grid_x from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "seed_advance" in r[0]]
k = len(starts) - 6
step = rows[starts[k]:starts[k + 1]]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:40]
prev = collections.Counter(); nxt = collections.Counter()
for i, r in enumerate(step):
    if "copyBuffer" in r[0]:
        prev[short(step[i-1][0])] += 1
        if i + 1 < len(step): nxt[short(step[i+1][0])] += 1
print("copyBuffer in step:", sum(prev.values()))
print("preceded by:", prev.most_common(12))
print("followed by:", nxt.most_common(12))
