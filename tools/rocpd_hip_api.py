#!/usr/bin/env python3
"""Summarise the HIP API calls of a rocprofv3 --hip-trace run (rocpd sqlite): per call count / total / average duration.   usage: tools/rocpd_hip_api.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
reg = [t for t in tables if t.startswith("rocpd_region")]
if not reg:
    print("no region table:", tables); sys.exit(1)
t = reg[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
strs = [x for x in tables if x.startswith("rocpd_string")][0]
q = "select s.string, count(*), sum(r.end - r.start), avg(r.end - r.start) from %s r join %s s on r.name_id = s.id group by s.string order by 3 desc" % (t, strs)
print("%-36s %8s %12s %10s" % ("call", "count", "total us", "avg us"))
for name, n, tot, avg in cur.execute(q):
    print("%-36s %8d %12.1f %10.2f" % (name, n, tot / 1e3, avg / 1e3))
