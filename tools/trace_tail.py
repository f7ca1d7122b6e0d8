#!/usr/bin/env python3
"""Per-kernel averages and the last dispatches of a rocprofv3 --kernel-trace run kept as a rocpd database (t_results.db)."""
import collections
import sqlite3
import sys

db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
show = int(sys.argv[3]) if len(sys.argv) > 3 else 30
c = sqlite3.connect(db)
rows = list(c.execute("select name,start,end,stream_id,queue_id from kernels order by start"))
d = collections.defaultdict(list)
for n, s, e, st, qid in rows:
    d[n[:60]].append(e - s)
for n, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    if n.startswith(("zvk", "void zvk")):
        print(f"{n:60s} calls {len(v):5d} avg {sum(v)/len(v)/1e3:9.2f} us  min {min(v)/1e3:9.2f}")
t0 = rows[-skip][1]
for n, s, e, st, qid in rows[-skip:-skip + show]:
    print(f"{n[:44]:44s} start {(s-t0)/1e3:8.1f}  dur {(e-s)/1e3:7.1f}  stream {st} queue {qid}")
