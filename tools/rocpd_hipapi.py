# -*- coding: utf-8 -*-
"""HIP API call statistics out of a rocprofv3 rocpd database (--hip-trace): calls, total and mean duration per API."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
reg = [t for t in tabs if t.startswith("rocpd_region") and "ext" not in t and "args" not in t]
strs = [t for t in tabs if t.startswith("rocpd_string")]
if not reg or not strs:
    print("tables:", tabs); sys.exit(0)
reg, strs = reg[0], strs[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % reg)]
q = "select s.string, count(*), sum(r.end - r.start), avg(r.end - r.start), min(r.end - r.start) from %s r join %s s on r.name_id = s.id group by s.string order by 3 desc" % (reg, strs)
print("%-44s %8s %12s %10s %10s" % ("api", "calls", "total_us", "avg_us", "min_us"))
for name, n, tot, avg, mn in cur.execute(q):
    print("%-44s %8d %12.1f %10.2f %10.2f" % (name[:44], n, tot / 1e3, avg / 1e3, mn / 1e3))
